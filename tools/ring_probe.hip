// ring_probe - does the DMA engine read a pinned chunk more slowly when CPU threads have just written it?
//   hipcc -O2 -std=c++17 tools/ring_probe.hip -o ring_probe -lpthread && ./ring_probe [threads] [chunks]
// A ring of pinned 16 MB chunks, filled by `threads` workers from a 1 GB source buffer (standing in for the
// page cache), each filled chunk copied to the GPU by one stream, a chunk refilled once its copy is done -
// the shape of wd_load_tile_files_batch's chunk loop.  Fill variants: ordinary stores (memcpy, what pread
// does), non-temporal stores (the data goes to DRAM past the caches), no fill at all (copies alone).
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <fcntl.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

__global__ void k_hammer(uint4 *buf, size_t n, int rounds)     // keeps HBM busy: read-modify-write of a big buffer
{
    for (int r = 0; r < rounds; r++)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            uint4 v = buf[i];
            v.x += 1;
            buf[i] = v;
        }
}

__global__ void k_spin(unsigned long long *out, long long clocks)   // keeps every CU busy without touching memory
{
    const long long t0 = clock64();
    unsigned long long x = threadIdx.x;
    while (clock64() - t0 < clocks)
        x = x * 6364136223846793005ull + 1442695040888963407ull;
    if (x == 42)
        out[0] = x;
}

static void nt_copy(uint8_t *dst, const uint8_t *src, size_t n)
{
    for (size_t i = 0; i < n; i += 64) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32));
        _mm256_stream_si256((__m256i *)(dst + i), a);
        _mm256_stream_si256((__m256i *)(dst + i + 32), b);
    }
    _mm_sfence();
}

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 16, K = argc > 2 ? atoi(argv[2]) : 4;
    const size_t chunk = 16u << 20, src_bytes = 1ull << 30, total_chunks = 256;
    uint8_t *src = (uint8_t *)malloc(src_bytes);
    memset(src, 3, src_bytes);
    std::vector<uint8_t *> ring((size_t)K);
    std::vector<hipEvent_t> copied((size_t)K);
    for (int k = 0; k < K; k++) {
        if (hipHostMalloc((void **)&ring[k], chunk, hipHostMallocDefault) != hipSuccess)
            return 1;
        memset(ring[k], 1, chunk);
        hipEventCreateWithFlags(&copied[k], hipEventDisableTiming | hipEventBlockingSync);
    }
    uint8_t *dev;
    hipMalloc((void **)&dev, chunk * 2);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int gpu_load = argc > 3 ? atoi(argv[3]) : 0;     // 1: a kernel streaming HBM beside the copies, 2: a kernel occupying the CUs
    uint4 *big = nullptr;
    const size_t big_n = (4ull << 30) / sizeof(uint4);
    hipStream_t st2;
    hipStreamCreateWithFlags(&st2, hipStreamNonBlocking);
    unsigned long long *d_out;
    hipMalloc((void **)&d_out, 8);
    if (gpu_load == 1)
        hipMalloc((void **)&big, big_n * sizeof(uint4));
    // a 2 GB file in the page cache, for the pread fill
    char fname[] = "/tmp/wd_ring_probe_XXXXXX";
    const int wfd = mkstemp(fname);
    for (int r = 0; r < 2; r++)
        if (write(wfd, src, src_bytes) != (ssize_t)src_bytes)
            return 2;
    close(wfd);
    for (int mode = 1; mode < 5; mode++) {                  // 1 memcpy, 2 non-temporal, 3 pread, 4 pread into a bounce buffer + non-temporal
        if (gpu_load == 1)
            hipLaunchKernelGGL(k_hammer, dim3(4096), dim3(256), 0, st2, big, big_n, 40);
        if (gpu_load == 2)
            hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, st2, d_out, 400000000ll);
        std::mutex mu;
        std::condition_variable cv;
        size_t free_upto = (size_t)K;                        // chunks < free_upto may be filled
        std::vector<std::atomic<int>> parts(total_chunks);
        for (auto &p : parts)
            p = 0;
        const int pieces = 8;                                // a chunk is filled in 8 pieces of 2 MB by whoever comes
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (;;) {
                const size_t job = next.fetch_add(1);
                const size_t g = job / pieces, piece = job % pieces;
                if (g >= total_chunks)
                    return;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return g < free_upto; });
                }
                uint8_t *d = ring[g % K] + piece * (chunk / pieces);
                const uint8_t *s = src + ((job * (chunk / pieces)) % src_bytes);
                if (mode == 1)
                    memcpy(d, s, chunk / pieces);
                else if (mode == 2)
                    nt_copy(d, s, chunk / pieces);
                else {
                    thread_local int fd = open(fname, O_RDONLY);
                    thread_local uint8_t *bounce = (uint8_t *)aligned_alloc(64, 256 << 10);
                    const off_t at = (off_t)((job * (chunk / pieces)) % (2 * src_bytes - chunk));
                    if (mode == 3) {
                        size_t got = 0;
                        while (got < chunk / pieces) {
                            const ssize_t k = pread(fd, d + got, chunk / pieces - got, at + (off_t)got);
                            if (k <= 0)
                                break;
                            got += (size_t)k;
                        }
                    } else {
                        for (size_t got = 0; got < chunk / pieces; got += 256 << 10) {
                            if (pread(fd, bounce, 256 << 10, at + (off_t)got) != (256 << 10))
                                break;
                            nt_copy(d + got, bounce, 256 << 10);
                        }
                    }
                }
                if (parts[g].fetch_add(1) == pieces - 1) {
                    std::lock_guard<std::mutex> lk(mu);
                    cv.notify_all();
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++)
            pool.emplace_back(worker);
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t g = 0; g < total_chunks; g++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return parts[g].load() == pieces; });
            }
            hipMemcpyAsync(dev + (g & 1) * chunk, ring[g % K], chunk, hipMemcpyHostToDevice, st);
            hipEventRecord(copied[g % K], st);
            if (g >= 1) {
                hipEventSynchronize(copied[(g - 1) % K]);
                std::lock_guard<std::mutex> lk(mu);
                free_upto = g + K;
                cv.notify_all();
            }
        }
        hipStreamSynchronize(st);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        {
            std::lock_guard<std::mutex> lk(mu);
            free_upto = total_chunks + K;
            cv.notify_all();
        }
        for (auto &t : pool)
            t.join();
        const bool still = gpu_load && hipStreamQuery(st2) == hipErrorNotReady;
        hipStreamSynchronize(st2);
        printf("%-34s %d threads, ring of %d, GPU %s: %.1f GB/s through the ring%s\n",
               mode == 1 ? "filled with ordinary stores" : mode == 2 ? "filled with non-temporal stores"
               : mode == 3 ? "filled by pread (page cache)" : "pread into 256 KB bounce + non-temporal", threads, K,
               gpu_load == 1 ? "streaming HBM" : gpu_load == 2 ? "CUs occupied" : "idle", total_chunks * chunk / dt / 1e9,
               gpu_load && !still ? " (the kernel ended early)" : "");
        fflush(stdout);
    }
    unlink(fname);
    return 0;
}
