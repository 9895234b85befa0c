// microbench_fetch.hip - calibrates rocprofv3's FETCH_SIZE / TCC_MISS for the scan kernel's
// access pattern (one byte out of a cache line, lines far apart), as MI355X_MICROARCH.md
// section HBM asks before trusting an absolute.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mb tools/microbench_fetch.hip && /tmp/mb
// and under `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- /tmp/mb`
// (TCC_MISS_sum / TCC_HIT_sum in a second pass).  Each kernel prints its known byte counts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one byte from each of n distinct 128-byte lines (odd multiplier mod 2^b is a bijection)
__global__ void k_one_byte_per_line(const uint8_t *buf, uint32_t line_mask, uint32_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t line = (i * 2654435761u) & line_mask;
        acc += buf[(size_t)line * 128u + (line & 63u)];            // somewhere in the first half
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
// two bytes per line, one in each 64-byte half
__global__ void k_both_halves(const uint8_t *buf, uint32_t line_mask, uint32_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t line = (i * 2654435761u) & line_mask;
        acc += buf[(size_t)line * 128u + (line & 63u)];
        acc += buf[(size_t)line * 128u + 64u + (line & 63u)];
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
// 11 consecutive bytes (a honeycomb row segment) from each line
__global__ void k_row_segment(const uint8_t *buf, uint32_t line_mask, uint32_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n * 11u; i += gridDim.x * blockDim.x) {
        uint32_t seg = i / 11u, j = i - seg * 11u;
        uint32_t line = (seg * 2654435761u) & line_mask;
        acc += buf[(size_t)line * 128u + 20u + j];
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
// plain streaming read, 16 B per lane
__global__ void k_stream(const uint4 *buf, size_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = buf[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main()
{
    const size_t span = 4ull << 30;                 // 4 GiB, far beyond L2 + Infinity Cache
    const uint32_t n_lines = (uint32_t)(span / 128);
    const uint32_t n = n_lines / 2;                 // visit half of the lines, each once
    uint8_t *buf; uint32_t *sink;
    CK(hipMalloc(&buf, span)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, span)); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch, double lines, double useful) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; r++) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-22s %.3f ms  lines %.0f  -> %.1f Glines/s; if 64 B/line %.2f TB/s, if 128 B/line %.2f TB/s; useful %.0f B\n",
               name, best, lines, lines / best / 1e6, lines * 64 / best / 1e9, lines * 128 / best / 1e9, useful);
    };
    dim3 g(256 * 16), b(256);
    timeit("one_byte_per_line", [&] { hipLaunchKernelGGL(k_one_byte_per_line, g, b, 0, 0, buf, n_lines - 1, n, sink); }, n, n);
    timeit("both_halves", [&] { hipLaunchKernelGGL(k_both_halves, g, b, 0, 0, buf, n_lines - 1, n, sink); }, n, 2.0 * n);
    timeit("row_segment_11B", [&] { hipLaunchKernelGGL(k_row_segment, g, b, 0, 0, buf, n_lines - 1, n / 4, sink); }, n / 4, 11.0 * (n / 4));
    timeit("stream_16B_per_lane", [&] { hipLaunchKernelGGL(k_stream, g, b, 0, 0, (const uint4 *)buf, span / 16, sink); }, span / 128.0, (double)span);
    return 0;
}
