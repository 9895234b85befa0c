// welldup_ingest.hip - files of a run directory into HBM: DEFLATE on the GPU (gpu_inflate.inc), the host
// loaders behind it, CBCL tile blocks, the interleaved layout, and the gather of a few wells' bytes for
// the duplicate log.  Replaces the file side of Tile.get_seqs (bcl_direct_reader.py:200-216, :222-253,
// :255-325, :333-345).
#include "wd_ctx.h"

#ifndef WD_UNIT_ID
#define WD_UNIT_ID "unknown"
#endif
namespace wd { const char *unit_id_ingest() { return WD_UNIT_ID; } }      // hash of this unit's sources (wd_build_id)

using namespace wd;

// (at global scope: completes InfJob / InfResult, which the context points to)
#include "gpu_inflate.inc"

namespace {

#include "device_common.inc"
#include "ingest_kernels.inc"
#include "gpu_inflate_kernels.inc"

}  // namespace

extern "C" {

// ---- ingest ---------------------------------------------------------------------------
namespace {

struct SlotLease {
    wd_ctx *ctx;
    wd_ctx::IngestSlot *slot = nullptr;
    explicit SlotLease(wd_ctx *c) : ctx(c)
    {
        std::lock_guard<std::mutex> g(ctx->ingest_mu);
        for (auto *s : ctx->ingest_slots)
            if (!s->busy) {
                slot = s;
                break;
            }
        if (!slot) {
            slot = new wd_ctx::IngestSlot();
            ctx->ingest_slots.push_back(slot);
        }
        slot->busy = true;
    }
    ~SlotLease()
    {
        std::lock_guard<std::mutex> g(ctx->ingest_mu);
        slot->busy = false;
    }
};

// whole file -> memory; false if it cannot be opened / read
bool slurp(const char *path, std::vector<uint8_t> &buf)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return false;
    bool ok = fseek(f, 0, SEEK_END) == 0;
    long n = ok ? ftell(f) : -1;
    ok = ok && n >= 0 && fseek(f, 0, SEEK_SET) == 0;
    if (ok) {
        buf.resize((size_t)n);
        ok = n == 0 || fread(buf.data(), 1, (size_t)n, f) == (size_t)n;
    }
    fclose(f);
    return ok;
}

// whole file -> the slot's own buffer (grow-only; 16 zero bytes follow the data, as fast_gunzip
// wants).  A fresh multi-megabyte vector per call means an mmap, its page faults and a munmap
// per file, and many loader threads then queue up on the process's memory-map lock.
bool slurp_into(const char *path, uint8_t *&buf, size_t &cap, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return false;
    bool ok = fseek(f, 0, SEEK_END) == 0;
    long n = ok ? ftell(f) : -1;
    ok = ok && n >= 0 && fseek(f, 0, SEEK_SET) == 0;
    if (ok && (size_t)n + 16 > cap) {
        free(buf);
        cap = (size_t)n + 16 + ((size_t)n >> 3);
        buf = (uint8_t *)malloc(cap);
        if (!buf) {
            cap = 0;
            ok = false;
        }
    }
    if (ok) {
        ok = n == 0 || fread(buf, 1, (size_t)n, f) == (size_t)n;
        memset(buf + n, 0, 16);
        *len = (size_t)n;
    }
    fclose(f);
    return ok;
}

constexpr size_t kInflateSlack = 274 + 320;   // room fast_gunzip may ask for beyond the data
#include "fast_inflate.inc"

int slot_reserve(wd_ctx *ctx, wd_ctx::IngestSlot *s, size_t need)
{
    if (!s->stream) {
        std::lock_guard<std::mutex> g(ctx->ingest_mu);
        size_t idx = 0;
        while (idx < ctx->ingest_slots.size() && ctx->ingest_slots[idx] != s)
            idx++;
        hipStream_t &st = ctx->slot_streams[idx % wd_ctx::kSlotStreams];
        if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
            return WD_ERR_HIP;
        s->stream = st;
    }
    if (need > s->cap) {
        (void)hipHostFree(s->pinned);
        s->pinned = nullptr;
        s->cap = 0;
        if (hipHostMalloc((void **)&s->pinned, need, hipHostMallocDefault) != hipSuccess)
            return WD_ERR_NOMEM;
        s->cap = need;
    }
    return WD_OK;
}

}  // namespace

int wd_interleave4(wd_ctx *ctx, const uint8_t *const src[4], int64_t n_clusters, uint8_t *dst_dev)
try {
    if (!ctx || !src || !dst_dev || n_clusters < 0)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    if (n_clusters > 0)
        hipLaunchKernelGGL(k_interleave4, dim3((unsigned)((n_clusters + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           ctx->stream, src[0], src[1], src[2], src[3], (long long)n_clusters, (uint32_t *)dst_dev);
    WD_HIP(ctx, hipGetLastError());
    return WD_OK;
} WD_CATCH

int wd_gunzip(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap, size_t *produced, int mode)
try {
    if (!src || !dst || !produced || (mode != 0 && mode != 1))
        return WD_ERR_ARG;
    *produced = 0;
    if (mode == 1) {
        std::vector<uint8_t> in(src_len + 16, 0), out(dst_cap + kInflateSlack);
        memcpy(in.data(), src, src_len);
        size_t n = 0;
        if (!fast_gunzip(in.data(), src_len, out.data(), dst_cap + 274, &n) || n > dst_cap)
            return WD_ERR_UNSUPPORTED;                     // the loaders would turn to zlib here
        memcpy(dst, out.data(), n);
        *produced = n;
        return WD_OK;
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (src_len > 0xFFFFFFFFu || dst_cap > 0x7FFFFFFFu)
        return WD_ERR_ARG;
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK)
        return WD_ERR_NOMEM;
    zs.next_in = const_cast<Bytef *>(src);
    zs.avail_in = (uInt)src_len;
    zs.next_out = dst;
    zs.avail_out = (uInt)dst_cap;
    int rc = WD_OK;
    for (;;) {
        const int zr = inflate(&zs, Z_NO_FLUSH);
        if (zr == Z_STREAM_END) {
            if (zs.avail_in == 0)
                break;
            if (inflateReset(&zs) != Z_OK) {
                rc = WD_ERR_CORRUPT;
                break;
            }
            continue;
        }
        if (zr != Z_OK || zs.avail_out == 0 || zs.avail_in == 0) {
            // corrupt; too long for dst; truncated
            rc = zr != Z_OK ? WD_ERR_CORRUPT : zs.avail_out == 0 ? WD_ERR_IO : WD_ERR_TRUNCATED;
            break;
        }
    }
    *produced = (size_t)(zs.next_out - dst);
    inflateEnd(&zs);
    return rc;
} WD_CATCH

// These two may be called from several host threads at once on one context (each call leases
// its own pinned buffer and copy stream); they do not touch the context's error string.
int wd_load_bcl_gz(wd_ctx *ctx, const char *path, uint8_t *dst_dev, int64_t n_clusters)
{
    return wd_load_bcl_gz_strided(ctx, path, dst_dev, n_clusters, 1);
}

int wd_load_bcl_gz_strided(wd_ctx *ctx, const char *path, uint8_t *dst_dev, int64_t n_clusters, int well_stride)
try {
    if (!ctx || !path || !dst_dev || n_clusters < 0 || (well_stride != 1 && well_stride != 4))
        return WD_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess)
        return WD_ERR_HIP;
    SlotLease lease(ctx);
    size_t raw_len = 0;
    if (!slurp_into(path, lease.slot->file, lease.slot->file_cap, &raw_len))
        return WD_ERR_IO;                                  // FileNotFoundError in the reference
    const uint8_t *raw = lease.slot->file;
    const size_t want = (size_t)n_clusters + 4;
    int rc = slot_reserve(ctx, lease.slot, want + 64 + kInflateSlack);
    if (rc)
        return rc;
    size_t produced = 0;
    bool bad = false, truncated = false;
    if (!ctx->fast_inflate ||
        !fast_gunzip(raw, raw_len, lease.slot->pinned, want + 64 + 274, &produced) || produced > want + 64) {
        // zlib: gunzip (possibly several concatenated members) straight into the pinned buffer
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK)
            return WD_ERR_NOMEM;
        zs.next_in = const_cast<Bytef *>(raw);
        zs.avail_in = (uInt)std::min<size_t>(raw_len, 0xFFFFFFFFu);
        produced = 0;
        bad = raw_len > 0xFFFFFFFFu;
        // (the reference reads the whole file, so a stream that decodes to more than a plane is still
        // decoded to its end - into a scratch buffer - for its CRC and end marker to be checked)
        std::vector<uint8_t> spill;
        while (!bad) {
            const bool full = produced >= want + 64;
            if (full && spill.empty())
                spill.resize(1u << 16);
            zs.next_out = full ? spill.data() : lease.slot->pinned + produced;
            zs.avail_out = full ? (uInt)spill.size() : (uInt)std::min<size_t>(want + 64 - produced, 0x7FFFFFFFu);
            const uInt in_before = zs.avail_in, out_before = zs.avail_out;
            const int zr = inflate(&zs, Z_NO_FLUSH);
            if (!full)
                produced = (size_t)(zs.next_out - lease.slot->pinned);
            if (zr == Z_STREAM_END) {
                if (zs.avail_in == 0)
                    break;
                if (inflateReset(&zs) != Z_OK)
                    bad = true;
                continue;
            }
            if (zr == Z_BUF_ERROR && zs.avail_in == 0) {
                truncated = true;                          // nothing left to read and no end marker seen
                break;
            }
            if (zr != Z_OK && !(zr == Z_BUF_ERROR && zs.avail_out == 0)) {
                bad = true;
                break;
            }
            if (zs.avail_in == 0 && zs.avail_out != 0) {
                truncated = true;                          // the stream ends before its end marker
                break;
            }
            if (zs.avail_in == in_before && zs.avail_out == out_before) {
                bad = true;                                // no progress: cannot happen with room and input at hand
                break;
            }
        }
        inflateEnd(&zs);
    }
    // what gzip.open(..).read() raises in the reference (bcl_direct_reader.py:208-209): BadGzipFile /
    // zlib.error for corrupt data, EOFError for a truncated file - not "file not found"
    if (bad)
        return WD_ERR_CORRUPT;
    if (truncated)
        return WD_ERR_TRUNCATED;
    if (produced < 4)
        return WD_ERR_FORMAT;
    uint32_t header;
    memcpy(&header, lease.slot->pinned, 4);
    if ((int64_t)header != n_clusters)                     // bcl_direct_reader.py:338
        return WD_ERR_FORMAT;
    if (produced < want)
        return WD_ERR_INDEX;                               // the reference fails at slurped_file[idx]
    if (n_clusters > 0 && well_stride == 1) {
        if (hipMemcpyAsync(dst_dev, lease.slot->pinned + 4, (size_t)n_clusters, hipMemcpyHostToDevice,
                           lease.slot->stream) != hipSuccess ||
            hipStreamSynchronize(lease.slot->stream) != hipSuccess)
            return WD_ERR_HIP;
    } else if (n_clusters > 0) {
        // interleaved layout: the plane lands in the slot's device scratch and is scattered into its
        // byte lane of the group of four cycles (same stream, so the order is given)
        const size_t need = ((size_t)n_clusters + 255) & ~(size_t)255;
        if (need > lease.slot->dev_cap) {
            (void)hipFree(lease.slot->dev);
            lease.slot->dev = nullptr;
            lease.slot->dev_cap = 0;
            if (hipMalloc((void **)&lease.slot->dev, need) != hipSuccess)
                return WD_ERR_NOMEM;
            lease.slot->dev_cap = need;
        }
        if (hipMemcpyAsync(lease.slot->dev, lease.slot->pinned + 4, (size_t)n_clusters, hipMemcpyHostToDevice,
                           lease.slot->stream) != hipSuccess)
            return WD_ERR_HIP;
        hipLaunchKernelGGL(k_scatter_plane4, dim3((unsigned)((n_clusters + 4ll * kBlock - 1) / (4ll * kBlock))), dim3(kBlock),
                           0, lease.slot->stream, lease.slot->dev, (long long)n_clusters, dst_dev);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(lease.slot->stream) != hipSuccess)
            return WD_ERR_HIP;
    }
    return WD_OK;
} WD_CATCH

// ---- a batch of .bcl.gz files through the GPU decoder ------------------------------------------
namespace {

// A batch call's turn at the shared ring, the copy stream and the launches: calls are served in the
// order they arrived (so that batches a caller queued up are read in that order), one at a time.
// The slot is chosen and locked while the turn is held: a slot's previous holder has had its turn
// and only waits for the GPU, and no later call can take the slot first.
extern "C++" {
// Threads of one call: joined whichever way the call ends (an exception on the way out of an
// extern "C" entry point must not meet a joinable std::thread: that is std::terminate).  A thread
// that cannot be started (EAGAIN under a process limit, no memory) is not an error while one runs.
struct Crew {
    std::vector<std::thread> v;
    std::function<void()> wake;                  // lets waiting threads go before the join of an unwind
    template <class F>
    int start(int want, F &fn, long long fail_after = -1)
    {
        int started = 0;
        try {
            v.reserve(v.size() + (size_t)std::max(want, 0));
            for (int t = 0; t < want; t++) {
                if (fail_after >= 0 && started >= fail_after)
                    throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
                v.emplace_back(std::ref(fn));
                started++;
            }
        } catch (const std::system_error &) {
        } catch (const std::bad_alloc &) {
        }
        return started;
    }
    void join()
    {
        for (auto &t : v)
            if (t.joinable())
                t.join();
        v.clear();
    }
    ~Crew()
    {
        if (!v.empty() && wake)
            wake();
        join();
    }
};

// The body of an extern "C" entry point that allocates: C++ exceptions end here, as error codes.
template <class F>
int guarded(wd_ctx *ctx, F &&body)
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        if (ctx)
            (void)hipDeviceSynchronize();        // nothing of a failed call stays in flight
        return WD_ERR_NOMEM;
    } catch (...) {
        if (ctx)
            (void)hipDeviceSynchronize();
        return WD_ERR_STATE;
    }
}

}  // extern "C++"

struct InflateTurn {
    wd_ctx *ctx;
    unsigned ticket;
    bool held = true;
    explicit InflateTurn(wd_ctx *c) : ctx(c), ticket(c->inflate_calls.fetch_add(1))
    {
        std::unique_lock<std::mutex> lk(ctx->inflate_mu);
        ctx->inflate_cv.wait(lk, [&] { return ctx->inflate_serving == ticket; });
    }
    void unlock()
    {
        if (!held)
            return;
        held = false;
        {
            std::lock_guard<std::mutex> lk(ctx->inflate_mu);
            ctx->inflate_serving = ticket + 1;
        }
        ctx->inflate_cv.notify_all();
    }
    ~InflateTurn() { unlock(); }
    InflateTurn(const InflateTurn &) = delete;
    InflateTurn &operator=(const InflateTurn &) = delete;
};

// Reader threads of a batch: what the caller asks for, but no more than the CPUs this process may use
// (its affinity mask and its cgroup's quota).  The readers copy at memory speed; more of them than
// CPUs only makes the quota run out in the middle of a period, and every thread of the process stops
// until the next one (tools/ring_probe.hip: 54 GB/s with 16 threads on 16 CPUs, 21 with 32).
inline int reader_threads(int asked)
{
    static const int cpus = [] {
        int n = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0)
            n = std::min(n > 0 ? n : 1 << 20, CPU_COUNT(&set));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {         // "max 100000" or "<quota> <period>"
            long long quota = 0, period = 0;
            if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
                n = std::min<long long>(n, std::max<long long>(1, quota / period));
            fclose(f);
        }
        return std::max(1, n);
    }();
    if (const char *e = getenv("WD_READER_THREADS"))
        return std::max(1, std::min(atoi(e), 256));
    return std::max(1, std::min({asked, 256, cpus}));
}

// A file (or a stretch of one) from the page cache into a chunk of the pinned ring.
//
// pread() straight into the ring - what rounds 1 and 2 did - is a kernel copy with ordinary stores: the
// chunk's lines sit dirty in the caches of whichever cores ran the readers, and the DMA engine that
// reads the chunk a moment later has to pull them out of there: 34-40 GB/s through the ring instead of
// the 54 GB/s the engine does on memory nobody has just written (tools/ring_probe.hip: the same ring
// filled by memcpy 49, by non-temporal stores 54, by pread 40.5 with 16 threads and 34 with 32).  So the
// readers pread into a small buffer of their own (it stays in the core's L2) and move it on with
// NON-TEMPORAL stores, which go to memory past the caches: the engine finds the chunk in DRAM.
// dst is 16-byte aligned (the files of a chunk start at multiples of 16).
constexpr size_t kBounceBytes = 256u << 10;

inline void nt_copy(uint8_t *dst, const uint8_t *src, size_t n)
{
    size_t i = 0;
    if (((uintptr_t)dst & 15) == 0) {
        for (; i + 64 <= n; i += 64) {
            const __m128i a = _mm_loadu_si128((const __m128i *)(src + i)), b = _mm_loadu_si128((const __m128i *)(src + i + 16));
            const __m128i c = _mm_loadu_si128((const __m128i *)(src + i + 32)), d = _mm_loadu_si128((const __m128i *)(src + i + 48));
            _mm_stream_si128((__m128i *)(dst + i), a);
            _mm_stream_si128((__m128i *)(dst + i + 16), b);
            _mm_stream_si128((__m128i *)(dst + i + 32), c);
            _mm_stream_si128((__m128i *)(dst + i + 48), d);
        }
    }
    if (i < n)
        memcpy(dst + i, src + i, n - i);
}

// -> bytes read (== n on success).  `direct`: the old way (WD_RING_DIRECT=1, for comparisons).
inline size_t read_into_ring(int fd, uint8_t *dst, size_t n, off_t at, std::vector<uint8_t> &bounce, bool direct)
{
    size_t got = 0;
    if (direct) {
        while (got < n) {
            const ssize_t k = pread(fd, dst + got, n - got, at + (off_t)got);
            if (k <= 0)
                break;
            got += (size_t)k;
        }
        return got;
    }
    if (bounce.size() < kBounceBytes)
        bounce.resize(kBounceBytes);
    while (got < n) {
        const ssize_t k = pread(fd, bounce.data(), std::min(kBounceBytes, n - got), at + (off_t)got);
        if (k <= 0)
            break;
        nt_copy(dst + got, bounce.data(), (size_t)k);      // (got stays a multiple of 16 until the last piece)
        got += (size_t)k;
    }
    _mm_sfence();                                           // the stores are on their way before the chunk is reported read
    return got;
}

// Wait for a chunk's copy.  The events are blocking ones (a thread woken by an interrupt); WD_CHUNK_SPIN=1
// polls instead, for experiments with what the wake-up costs.
hipError_t wait_copied(hipEvent_t ev)
{
    static const bool spin = getenv("WD_CHUNK_SPIN") && atoi(getenv("WD_CHUNK_SPIN")) != 0;
    if (!spin)
        return hipEventSynchronize(ev);
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady)
            return e;
        for (int i = 0; i < 64; i++)
            __builtin_ia32_pause();
    }
}

// how the ring's chunks are pinned (WD_RING_FLAGS: experiments with what the readers' writes cost the DMA)
unsigned ring_flags()
{
    const char *e = getenv("WD_RING_FLAGS");
    if (!e)
        return hipHostMallocDefault;
    unsigned f = 0;
    if (strstr(e, "wc")) f |= hipHostMallocWriteCombined;
    if (strstr(e, "noncoherent")) f |= hipHostMallocNonCoherent;
    if (strstr(e, "coherent") && !strstr(e, "noncoherent")) f |= hipHostMallocCoherent;
    if (strstr(e, "portable")) f |= hipHostMallocPortable;
    return f;
}

// buffers of a batch: pinned ring, streams, arena for `arena_bytes` of compressed files, n job slots
// ... the part every batch shares: the pinned ring (pinning memory is what takes time: 25 ms for four chunks
// of 16 MB), the streams and their events.  Also reached through option "inflate_warm", which lets a caller
// have it done beside its own start-up work instead of inside the first batch.
}  // namespace
extern "C++" int wd::inflate_prepare_shared(wd_ctx *ctx, int n_chunks)
{
    std::lock_guard<std::mutex> only_one(ctx->inflate_shared_mu);         // (a warm-up call beside a batch's)
    if (ctx->inflate_chunk_cap != ctx->inflate_chunk_bytes) {            // the option changed: new buffers
        // (the batch before may still be copying out of the old ones)
        if (ctx->inflate_streams[wd_ctx::kInflateStreams] &&
            hipStreamSynchronize(ctx->inflate_streams[wd_ctx::kInflateStreams]) != hipSuccess)
            return WD_ERR_HIP;
        for (auto &ch : ctx->inflate_chunks) {
            (void)hipHostFree(ch.pinned);
            ch.pinned = nullptr;
        }
        ctx->inflate_chunk_cap = ctx->inflate_chunk_bytes;
    }
    for (int c = 0; c < n_chunks; c++) {
        wd_ctx::InflateChunk &ch = ctx->inflate_chunks[c];
        if (!ch.copied && hipEventCreateWithFlags(&ch.copied, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess)
            return WD_ERR_HIP;
        if (!ch.pinned && hipHostMalloc((void **)&ch.pinned, ctx->inflate_chunk_cap + 64, ring_flags()) != hipSuccess)
            return WD_ERR_NOMEM;
    }
    // the copy stream and as many decode streams as are used (creating and destroying a stream costs 1 - 3 ms)
    const int n_dec = std::max(1, std::min(wd_ctx::kInflateStreams, getenv("WD_DECODE_STREAMS") ? atoi(getenv("WD_DECODE_STREAMS"))
                                                                                                 : ctx->inflate_decode_streams));
    for (int u = 0; u <= wd_ctx::kInflateStreams; u++) {
        hipStream_t &st = ctx->inflate_streams[u];
        if ((u < n_dec || u == wd_ctx::kInflateStreams) && !st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
            return WD_ERR_HIP;
    }
    for (auto &ev : ctx->inflate_ready)
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
            return WD_ERR_HIP;
    for (auto &ev : ctx->inflate_joined)
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
            return WD_ERR_HIP;
    return WD_OK;
}
namespace {

int inflate_prepare(wd_ctx *ctx, wd_ctx::InflateSlot &sl, int n_chunks, size_t arena_bytes, size_t n_jobs)
{
    if (!sl.done && hipEventCreateWithFlags(&sl.done, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess)
        return WD_ERR_HIP;
    if (const int rc = inflate_prepare_shared(ctx, n_chunks))
        return rc;
    if (arena_bytes > sl.arena_cap) {
        (void)hipFree(sl.arena);
        sl.arena = nullptr;
        sl.arena_cap = 0;
        const size_t want = arena_bytes + (arena_bytes >> 2) + 64;
        if (hipMalloc((void **)&sl.arena, want) != hipSuccess)
            return WD_ERR_NOMEM;
        sl.arena_cap = want;
    }
    if (n_jobs > sl.jobs_cap) {
        (void)hipHostFree(sl.h_jobs);
        (void)hipFree(sl.d_jobs);
        (void)hipHostFree(sl.h_res);
        (void)hipFree(sl.d_res);
        sl.h_jobs = sl.d_jobs = nullptr;
        sl.h_res = sl.d_res = nullptr;
        sl.jobs_cap = 0;
        const size_t want = n_jobs + (n_jobs >> 1) + 64;
        if (hipHostMalloc((void **)&sl.h_jobs, sizeof(InfJob) * want, hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&sl.d_jobs, sizeof(InfJob) * want) != hipSuccess ||
            hipHostMalloc((void **)&sl.h_res, sizeof(InfResult) * want, hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&sl.d_res, sizeof(InfResult) * want) != hipSuccess)
            return WD_ERR_NOMEM;
        sl.jobs_cap = want;
    }
    return WD_OK;
}

}  // namespace

int wd_load_bcl_gz_batch(wd_ctx *ctx, int n_files, const char *const *paths, uint8_t *const *dst_dev,
                         int64_t n_clusters, int threads, int *rc_out)
{
    return wd_load_tile_files_batch(ctx, n_files, paths, dst_dev, nullptr, n_clusters, 1, threads, rc_out);
}

static int load_tile_files_batch_impl(wd_ctx *ctx, int n_files, const char *const *paths, uint8_t *const *dst_dev,
                                      const uint8_t *is_filter, int64_t n_clusters, int well_stride, int threads, int *rc_out);

int wd_load_tile_files_batch(wd_ctx *ctx, int n_files, const char *const *paths, uint8_t *const *dst_dev,
                             const uint8_t *is_filter, int64_t n_clusters, int well_stride, int threads, int *rc_out)
{
    return guarded(ctx, [&] {
        return load_tile_files_batch_impl(ctx, n_files, paths, dst_dev, is_filter, n_clusters, well_stride, threads, rc_out);
    });
}

static int load_tile_files_batch_impl(wd_ctx *ctx, int n_files, const char *const *paths, uint8_t *const *dst_dev,
                                      const uint8_t *is_filter, int64_t n_clusters, int well_stride, int threads, int *rc_out)
{
    if (well_stride != 1 && well_stride != 4)
        return WD_ERR_ARG;
    if (!ctx || n_files < 0 || (n_files && (!paths || !dst_dev)) || n_clusters < 0 || n_clusters > 0x7FFFFFF0ll)
        return WD_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess)
        return WD_ERR_HIP;
    const auto call_t0 = std::chrono::steady_clock::now();               // (WD_INFLATE_STATS)
    // a slot for the whole call, the shared ring / streams only while this batch is read and launched
    InflateTurn batch_lock(ctx);
    wd_ctx::InflateSlot &slot = ctx->inflate_slots[batch_lock.ticket % wd_ctx::kInflateSlots];
    std::lock_guard<std::mutex> slot_lock(slot.mu);
    threads = reader_threads(threads);
    const bool ring_direct = getenv("WD_RING_DIRECT") && atoi(getenv("WD_RING_DIRECT")) != 0;
    constexpr int kChunks = wd_ctx::kInflateChunks, kStreams = wd_ctx::kInflateStreams;
    const size_t chunk_bytes = ctx->inflate_chunk_bytes;

    enum : int { PENDING = 1, HOST = 2, EARLY = 3 };                     // beside the WD_* codes (<= 0)
    std::vector<int> rc((size_t)n_files, PENDING), early_rc((size_t)n_files, WD_OK);
    std::atomic<long long> n_early{0};
    std::vector<size_t> size((size_t)n_files, 0), offset((size_t)n_files, 0);   // offset: in the group's chunk
    std::vector<uint32_t> stream_off((size_t)n_files, 0);
    std::vector<uint64_t> trailer((size_t)n_files, 0);                   // CRC-32 | length << 32, as the file ends
    std::vector<int> group_of((size_t)n_files, -1);
    struct Group { int first, last; size_t bytes, arena_at; std::atomic<int> remaining{0}; };
    std::vector<std::unique_ptr<Group>> groups;
    size_t arena_bytes = 0, n_jobs = 0;
    const auto turn_t0 = std::chrono::steady_clock::now();               // (this call's turn has come)
    // sizes, then groups of consecutive files that fit a chunk
    for (int i = 0; i < n_files; i++) {
        struct stat st;
        if (!paths[i] || !dst_dev[i] || (((uintptr_t)dst_dev[i] & 3) && (well_stride == 1 || (is_filter && is_filter[i])))) {
            rc[(size_t)i] = WD_ERR_ARG;
        } else if (stat(paths[i], &st) != 0 || !S_ISREG(st.st_mode)) {
            rc[(size_t)i] = WD_ERR_IO;                                   // FileNotFoundError in the reference
        } else if (is_filter && is_filter[i] && (st.st_size < 12 || (int64_t)st.st_size != 12 + n_clusters)) {
            rc[(size_t)i] = WD_ERR_FORMAT;                               // bcl_direct_reader.py:240
        } else if ((size_t)st.st_size + 16 > chunk_bytes || st.st_size < 12 || (uint64_t)st.st_size > 0x1FFFFFF0ull ||
                   (st.st_size < 18 && !(is_filter && is_filter[i]))) {
            rc[(size_t)i] = HOST;
        } else {
            size[(size_t)i] = (size_t)st.st_size;
            const size_t padded = (size[(size_t)i] + 15) & ~(size_t)15;
            if (groups.empty() || groups.back()->bytes + padded > chunk_bytes)
                groups.emplace_back(new Group{i, i, 0, arena_bytes});
            Group &g = *groups.back();
            offset[(size_t)i] = g.bytes;
            g.bytes += padded;
            arena_bytes += padded;
            g.last = i;
            g.remaining.fetch_add(1);
            group_of[(size_t)i] = (int)groups.size() - 1;
            n_jobs++;
        }
    }
    const int n_groups = (int)groups.size();
    const auto stat_t1 = std::chrono::steady_clock::now();
    if (n_groups) {
        // (interleaved layout: the planes are decoded into the arena and scattered into their byte lanes)
        const size_t plane_room = well_stride == 4 ? (((size_t)n_clusters + 8 + 255) & ~(size_t)255) : 0;
        const int prc = inflate_prepare(ctx, slot, std::min(n_groups, kChunks),
                                        ((arena_bytes + 255) & ~(size_t)255) + plane_room * n_jobs, n_jobs);
        if (prc)
            return prc;
        // (the batch before may still be decoding; its chunk copies are behind us after this)
        if (hipStreamSynchronize(ctx->inflate_streams[kStreams]) != hipSuccess)
            return WD_ERR_HIP;
    }

    std::mutex mu;
    std::condition_variable cv;
    int free_upto = kChunks;                 // groups < free_upto may be filled
    bool abort_all = false;
    std::atomic<int> next_file{0};

    auto reader = [&]() {
        for (;;) {
            const int i = next_file.fetch_add(1);
            if (i >= n_files)
                return;
            const int g = group_of[(size_t)i];
            if (g < 0)
                continue;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return abort_all || g < free_upto; });
                if (abort_all)
                    return;
            }
            uint8_t *dst = ctx->inflate_chunks[g % kChunks].pinned + offset[(size_t)i];
            const size_t sz = size[(size_t)i];
            bool ok = false, early = false;
            const int fd = open(paths[i], O_RDONLY);
            if (fd >= 0) {
                thread_local std::vector<uint8_t> bounce;
                ok = read_into_ring(fd, dst, sz, 0, bounce, ring_direct) == sz;
                close(fd);
            }
            if (ok && is_filter && is_filter[i]) {                       // .filter: header 0, 3, n (:148-152, :236), then the bytes
                uint32_t head[3];
                memcpy(head, dst, 12);
                if (head[0] != 0 || head[1] != 3 || (int64_t)head[2] != n_clusters)
                    rc[(size_t)i] = WD_ERR_FORMAT;
            } else if (!ok || !inf_gzip_header(dst, sz, &stream_off[(size_t)i])) {
                rc[(size_t)i] = HOST;                                    // let the host path say what is wrong with it
            } else {
                memcpy(&trailer[(size_t)i], dst + sz - 8, 8);
                // A file that expands four hundredfold and more (a failed cycle: a plane of no-calls) holds
                // stretches the GPU decoder declines (one piece of the stream, 256-fold).  Sending it
                // through the launch only to decode it on the host afterwards would hold this batch back
                // by a serial tail: this thread decodes it NOW, beside the reads, the copies and the launch.
                if ((trailer[(size_t)i] >> 32) >= (uint64_t)sz * 400) {
                    early = true;
                    rc[(size_t)i] = EARLY;                               // (before the group is reported read: the chunk loop must not queue it)
                }
            }
            if (groups[(size_t)g]->remaining.fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> lk(mu);
                cv.notify_all();
            }
            if (early) {
                const int hrc = wd_load_bcl_gz_strided(ctx, paths[i], dst_dev[i], n_clusters, well_stride);
                early_rc[(size_t)i] = hrc;
                n_early.fetch_add(1);
            }
        }
    };
    Crew pool;
    pool.wake = [&] {
        std::lock_guard<std::mutex> lk(mu);
        abort_all = true;
        cv.notify_all();
    };
    if (pool.start(std::min(threads, std::max(1, n_files)), reader, ctx->test_thread_limit) == 0)
        return WD_ERR_NOMEM;                     // not one reader thread could be started

    // The chunks go to the arena one by one on the copy stream; a launch on the decode stream waits
    // for the copy of its last chunk (see kInflateLaunchFiles).
    std::vector<int> job_file;                                           // file index of every job, in launch order
    job_file.reserve(n_jobs);
    int hip_rc = WD_OK;
    hipStream_t copy_stream = ctx->inflate_streams[kStreams];
    const int copy_depth = std::max(1, std::min(kChunks - 1, getenv("WD_COPY_DEPTH") ? atoi(getenv("WD_COPY_DEPTH")) : 1));
    const int n_dec = std::max(1, std::min(kStreams, getenv("WD_DECODE_STREAMS") ? atoi(getenv("WD_DECODE_STREAMS")) : ctx->inflate_decode_streams));
    const size_t launch_files = (size_t)std::max(64, getenv("WD_LAUNCH_FILES") ? atoi(getenv("WD_LAUNCH_FILES")) : ctx->inflate_launch_files);
    int si = (int)(ctx->inflate_launch_seq % (unsigned)n_dec);           // consecutive launches decode on the streams in turn
    unsigned used_streams = 0;                                           // bit u: this batch launched on stream u
    size_t j0 = 0;                                                       // first job of the launch being gathered
    double wait_read_s = 0, wait_copy_s = 0;                             // (WD_INFLATE_STATS) what the chunk loop waits for
    const bool dma_probe = getenv("WD_INFLATE_STATS") && atoi(getenv("WD_INFLATE_STATS")) >= 2;
    std::vector<hipEvent_t> dma_ev;
    size_t dma_bytes = 0;
    const auto loop_t0 = std::chrono::steady_clock::now();
    for (int g = 0; g < n_groups && hip_rc == WD_OK; g++) {
        Group &grp = *groups[(size_t)g];
        {
            const auto w0 = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return grp.remaining.load() == 0; });
            wait_read_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        }
        wd_ctx::InflateChunk &ch = ctx->inflate_chunks[g % kChunks];
        uint8_t *dev = slot.arena + grp.arena_at;
        std::vector<int> plain;                                          // .filter files of the chunk: copied, not decoded
        for (int i = grp.first; i <= grp.last; i++) {
            if (group_of[(size_t)i] != g || rc[(size_t)i] != PENDING)
                continue;
            if (is_filter && is_filter[i]) {
                plain.push_back(i);
                continue;
            }
            InfJob &j = slot.h_jobs[job_file.size()];
            j.file = reinterpret_cast<const uint32_t *>(dev + offset[(size_t)i]);
            j.obase = well_stride == 4 ? slot.arena + ((arena_bytes + 255) & ~(size_t)255) +
                                             ((((size_t)n_clusters + 8 + 255) & ~(size_t)255) * job_file.size())
                                       : dst_dev[i] - 4;
            j.file_bytes = (uint32_t)size[(size_t)i];
            j.stream_off = stream_off[(size_t)i];
            j.out_cap = (uint32_t)(n_clusters + 4);
            j.pad_ = 0;
            job_file.push_back(i);
        }
        if (dma_probe && (int)dma_ev.size() < 2 * n_groups) {           // (WD_INFLATE_STATS=2: how long the engine itself takes)
            hipEvent_t e0 = nullptr, e1 = nullptr;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            dma_ev.push_back(e0);
            dma_ev.push_back(e1);
            (void)hipEventRecord(e0, copy_stream);
        }
        if (hipMemcpyAsync(dev, ch.pinned, grp.bytes, hipMemcpyHostToDevice, copy_stream) != hipSuccess ||
            (dma_probe && hipEventRecord(dma_ev.back(), copy_stream) != hipSuccess) ||
            hipEventRecord(ch.copied, copy_stream) != hipSuccess) {
            hip_rc = WD_ERR_HIP;
            break;
        }
        dma_bytes += grp.bytes;
        for (int i : plain) {
            if (n_clusters > 0 && hipMemcpyAsync(dst_dev[i], dev + offset[(size_t)i] + 12, (size_t)n_clusters,
                                                  hipMemcpyDeviceToDevice, copy_stream) != hipSuccess) {
                hip_rc = WD_ERR_HIP;
                break;
            }
            rc[(size_t)i] = WD_OK;                                       // (the call returns after the copy stream has drained)
        }
        if (hip_rc != WD_OK)
            break;
        // enough files for a launch, or the last chunk: decode them
        if (g + 1 == n_groups || job_file.size() - j0 >= launch_files) {
            const unsigned nj = (unsigned)(job_file.size() - j0);
            si = (int)(ctx->inflate_launch_seq % (unsigned)n_dec);
            hipStream_t stream = ctx->inflate_streams[si];
            if (nj) {
                ctx->inflate_launch_seq++;
                used_streams |= 1u << si;
                if (hipEventRecord(ctx->inflate_ready[si], copy_stream) != hipSuccess ||
                    hipStreamWaitEvent(stream, ctx->inflate_ready[si], 0) != hipSuccess ||
                    hipMemcpyAsync(slot.d_jobs + j0, slot.h_jobs + j0, sizeof(InfJob) * nj,
                                   hipMemcpyHostToDevice, stream) != hipSuccess) {
                    hip_rc = WD_ERR_HIP;
                    break;
                }
                // waves per file: eight while every file of the launch gets a CU of its own (24 ms per
                // file), else four (34 ms, two files per CU - three, 36 ms, when no file of the launch
                // expands much: the small-window form); one wave per file (89 ms, three per CU) on request
                // (a launch that is one of several of its batch shares the chip with the others: four waves)
                const int waves = ctx->inflate_waves ? ctx->inflate_waves : (nj <= 256 && n_jobs <= 256) ? 8 : 4;
                bool slim = true;
                for (size_t q = j0; q < job_file.size() && slim; q++) {
                    const int i = job_file[q];
                    slim = (trailer[(size_t)i] >> 32) * 4 <= (uint64_t)size[(size_t)i] * 7;
                }
                InfJob *dj = slot.d_jobs + j0;
                InfResult *dr = slot.d_res + j0;
                if (waves == 8)
                    hipLaunchKernelGGL((k_inflate<8, 256>), dim3(nj), dim3(512), 0, stream, dj, dr);
                else if (waves == 4 && slim)
                    hipLaunchKernelGGL((k_inflate<4, 256, 4>), dim3(nj), dim3(256), 0, stream, dj, dr);
                else if (waves == 4)
                    hipLaunchKernelGGL((k_inflate<4, 256>), dim3(nj), dim3(256), 0, stream, dj, dr);
                else
                    hipLaunchKernelGGL((k_inflate<1, 512>), dim3(nj), dim3(64), 0, stream, dj, dr);
                hipLaunchKernelGGL(k_inflate_crc, dim3(nj), dim3(256), 0, stream, dj, dr);
                if (well_stride == 4 && n_clusters > 0)
                    for (unsigned q = 0; q < nj; q++)
                        hipLaunchKernelGGL(k_scatter_plane4,
                                           dim3((unsigned)((n_clusters + 4ll * kBlock - 1) / (4ll * kBlock))), dim3(kBlock), 0,
                                           stream, slot.h_jobs[j0 + q].obase + 4, (long long)n_clusters,
                                           dst_dev[job_file[j0 + q]]);
                if (hipGetLastError() != hipSuccess ||
                    hipMemcpyAsync(slot.h_res + j0, slot.d_res + j0, sizeof(InfResult) * nj,
                                   hipMemcpyDeviceToHost, stream) != hipSuccess) {
                    hip_rc = WD_ERR_HIP;
                    break;
                }
            }
            j0 = job_file.size();
        }
        if (g >= copy_depth) {
            // group g - copy_depth + kChunks wants the chunk of group g - copy_depth: once that copy is
            // done the readers may fill it again.  copy_depth copies are queued at any time, so that the
            // engine has the next one at hand when this thread is late in noticing that one has ended
            // (it shares its CPUs with the readers)
            const auto w0 = std::chrono::steady_clock::now();
            if (wait_copied(ctx->inflate_chunks[(g - copy_depth) % kChunks].copied) != hipSuccess) {
                hip_rc = WD_ERR_HIP;
                break;
            }
            wait_copy_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            std::lock_guard<std::mutex> lk(mu);
            free_upto = g - copy_depth + 1 + kChunks;
            cv.notify_all();
        }
    }
    if (hip_rc != WD_OK) {
        std::lock_guard<std::mutex> lk(mu);
        abort_all = true;
        cv.notify_all();
    }
    pool.join();
    std::vector<uint8_t> was_early((size_t)n_files, 0);
    for (int i = 0; i < n_files; i++)
        if (rc[(size_t)i] == EARLY) {
            rc[(size_t)i] = early_rc[(size_t)i];                         // the host loader's verdict, as for every file it takes
            was_early[(size_t)i] = 1;
        }
    ctx->inflate_files_host += n_early.load();
    ctx->inflate_files_early += n_early.load();
    if (getenv("WD_INFLATE_STATS")) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return 1e3 * std::chrono::duration<double>(b - a).count();
        };
        fprintf(stderr, "[wd inflate] waited %.1f ms for its turn, %.1f ms of stat, %.1f ms of buffers and the copy stream; "
                        "chunk loop %.1f ms for %d chunks: waited %.1f ms for the readers, %.1f ms for chunk copies\n",
                ms(call_t0, turn_t0), ms(turn_t0, stat_t1), ms(stat_t1, loop_t0),
                ms(loop_t0, std::chrono::steady_clock::now()), n_groups, 1e3 * wait_read_s, 1e3 * wait_copy_s);
    }
    if (dma_probe && !dma_ev.empty()) {
        (void)hipStreamSynchronize(copy_stream);
        double busy = 0, span = 0, longest = 0;
        for (size_t q = 0; q + 1 < dma_ev.size(); q += 2) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, dma_ev[q], dma_ev[q + 1]);
            busy += ms;
            longest = std::max<double>(longest, ms);
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, dma_ev.front(), dma_ev.back());
        span = ms;
        fprintf(stderr, "[wd inflate] the copies themselves: %.1f MB in %zu copies, engine busy %.1f ms (%.1f GB/s while copying, "
                        "longest copy %.2f ms), first start to last end %.1f ms\n",
                dma_bytes / 1e6, dma_ev.size() / 2, busy, dma_bytes / 1e6 / std::max(busy, 1e-9), longest, span);
        for (hipEvent_t e : dma_ev)
            (void)hipEventDestroy(e);
    }
    // the next batch may start reading; this one waits for its last results
    // (.filter copies ride on the copy stream: the decode stream's event must come after them)
    // (the last launch's stream gathers the others the batch used, then signals the batch done)
    for (int u = 0; u < kStreams && hip_rc == WD_OK && n_groups; u++)
        if (u != si && (used_streams >> u & 1u) &&
            (hipEventRecord(ctx->inflate_joined[u], ctx->inflate_streams[u]) != hipSuccess ||
             hipStreamWaitEvent(ctx->inflate_streams[si], ctx->inflate_joined[u], 0) != hipSuccess))
            hip_rc = WD_ERR_HIP;
    if (hip_rc == WD_OK && n_groups &&
        (hipEventRecord(ctx->inflate_ready[si], copy_stream) != hipSuccess ||
         hipStreamWaitEvent(ctx->inflate_streams[si], ctx->inflate_ready[si], 0) != hipSuccess ||
         hipEventRecord(slot.done, ctx->inflate_streams[si]) != hipSuccess))
        hip_rc = WD_ERR_HIP;
    if (hip_rc != WD_OK)
        (void)hipDeviceSynchronize();                                    // nothing of a failed call stays in flight
    batch_lock.unlock();
    const auto launched_t = std::chrono::steady_clock::now();
    if (hip_rc == WD_OK && n_groups && hipEventSynchronize(slot.done) != hipSuccess)
        hip_rc = WD_ERR_HIP;
    if (hip_rc != WD_OK)
        return hip_rc;

    const bool want_stats = getenv("WD_INFLATE_STATS") != nullptr;
    if (want_stats)
        fprintf(stderr, "[wd inflate] decoded %.1f ms after the last launch was queued, %.1f ms after the call began\n",
                1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - launched_t).count(),
                1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - call_t0).count());
    unsigned long long st[14] = {0}, real_sum = 0;
    for (size_t j = 0; j < job_file.size(); j++) {
        const int i = job_file[j];
        const InfResult &r = slot.h_res[j];
        const uint32_t crc = (uint32_t)trailer[(size_t)i], isize = (uint32_t)(trailer[(size_t)i] >> 32);
        const bool good = r.status == INF_OK && (size_t)r.end_byte + 8 == size[(size_t)i] && crc == r.crc &&
                          isize == r.produced && (int64_t)r.produced == n_clusters + 4 && (int64_t)r.head == n_clusters;
        rc[(size_t)i] = good ? WD_OK : HOST;
        real_sum += r.t_real;
        if (want_stats) {
            const unsigned long long v[12] = {r.t_header, r.t_build, r.t_stage, r.t_pass, r.t_emit, r.t_resolve,
                                              r.t_flush, r.t_total, r.windows, r.passes, r.rounds, r.blocks};
            for (int q = 0; q < 12; q++)
                st[q] += v[q];
            st[12] += r.t_real;
            st[13] += r.t_res1;
        }
    }
    if (!job_file.empty())
        ctx->inflate_us_per_file = (long long)(real_sum / 100 / job_file.size());     // t_real counts 10 ns
    if (want_stats && !job_file.empty()) {
        int occ[4] = {-1, -1, -1, -1};
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[3], (const void *)k_inflate<4, 256, 4>, 256, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[0], (const void *)k_inflate<1, 512>, 64, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[1], (const void *)k_inflate<4, 256>, 256, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[2], (const void *)k_inflate<8, 256>, 512, 0);
        fprintf(stderr, "[wd inflate] workgroups per CU by the runtime's count: 1 wave %d, 4 waves %d (slim: %d), 8 waves %d\n", occ[0],
                occ[1], occ[3], occ[2]);
        const double nf = (double)job_file.size();
        fprintf(stderr, "[wd inflate] files %d in %d chunks | Mclk per file: header %.2f tables %.2f stage %.2f passes %.2f "
                        "emit %.2f resolve %.2f (first sweep %.2f) flush %.2f total %.2f = %.1f ms at %.2f GHz | per file: windows %.0f passes %.0f "
                        "rounds %.0f blocks %.0f\n",
                (int)nf, n_groups, st[0] / 1e6 / nf, st[1] / 1e6 / nf, st[2] / 1e6 / nf, st[3] / 1e6 / nf, st[4] / 1e6 / nf,
                st[5] / 1e6 / nf, st[13] / 1e6 / nf, st[6] / 1e6 / nf, st[7] / 1e6 / nf, st[12] / 1e5 / nf,
                st[12] ? (double)st[7] / (double)st[12] / 10.0 : 0.0, st[8] / nf, st[9] / nf, st[10] / nf, st[11] / nf);
    }

    // whatever the GPU decoder did not take or did not like: the host loader, whose verdict counts
    std::vector<int> todo;
    for (int i = 0; i < n_files; i++)
        if (rc[(size_t)i] == HOST || rc[(size_t)i] == PENDING)
            todo.push_back(i);
    ctx->inflate_files_host += (long long)todo.size();
    long long by_gpu = 0;
    for (int i = 0; i < n_files; i++)
        by_gpu += rc[(size_t)i] == WD_OK && !(is_filter && is_filter[i]) && !was_early[(size_t)i];
    ctx->inflate_files_gpu += by_gpu;
    if (!todo.empty()) {
        std::atomic<size_t> next{0};
        auto host = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= todo.size())
                    return;
                const int i = todo[k];
                rc[(size_t)i] = is_filter && is_filter[i] ? wd_load_filter(ctx, paths[i], dst_dev[i], n_clusters)
                                                          : wd_load_bcl_gz_strided(ctx, paths[i], dst_dev[i], n_clusters, well_stride);
            }
        };
        Crew hp;
        if (hp.start((int)std::min((size_t)threads, todo.size()) - 1, host, ctx->test_thread_limit) >= 0)
            host();                              // this thread works too: the list is done even if none could be started
        hp.join();
    }
    int first = WD_OK;
    for (int i = 0; i < n_files; i++) {
        if (rc_out)
            rc_out[i] = rc[(size_t)i];
        if (first == WD_OK && rc[(size_t)i] != WD_OK)
            first = rc[(size_t)i];
    }
    return first;
}

int wd_load_filter(wd_ctx *ctx, const char *path, uint8_t *dst_dev, int64_t n_clusters)
try {
    if (!ctx || !path || !dst_dev || n_clusters < 0)
        return WD_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess)
        return WD_ERR_HIP;
    std::vector<uint8_t> raw;
    if (!slurp(path, raw))
        return WD_ERR_IO;
    if (raw.size() < 12)
        return WD_ERR_FORMAT;
    uint32_t head[3];
    memcpy(head, raw.data(), 12);
    if (head[0] != 0 || head[1] != 3 || (int64_t)head[2] != n_clusters)   // :148-152, :236
        return WD_ERR_FORMAT;
    if (raw.size() != 12 + (size_t)n_clusters)                            // :240
        return WD_ERR_FORMAT;
    SlotLease lease(ctx);
    int rc = slot_reserve(ctx, lease.slot, (size_t)n_clusters + 64);
    if (rc)
        return rc;
    if (n_clusters > 0) {
        memcpy(lease.slot->pinned, raw.data() + 12, (size_t)n_clusters);
        if (hipMemcpyAsync(dst_dev, lease.slot->pinned, (size_t)n_clusters, hipMemcpyHostToDevice,
                           lease.slot->stream) != hipSuccess ||
            hipStreamSynchronize(lease.slot->stream) != hipSuccess)
            return WD_ERR_HIP;
    }
    return WD_OK;
} WD_CATCH

int wd_load_cbcl_tile(wd_ctx *ctx, const char *path, int tile_number, const uint8_t *filter_dev,
                      int64_t n_clusters, uint8_t *dst_dev)
{
    return wd_load_cbcl_tile_strided(ctx, path, tile_number, filter_dev, n_clusters, dst_dev, 1);
}

int wd_load_cbcl_tile_strided(wd_ctx *ctx, const char *path, int tile_number, const uint8_t *filter_dev,
                              int64_t n_clusters, uint8_t *dst_dev, int well_stride)
try {
    if (!ctx || !path || !dst_dev || !filter_dev || n_clusters < 0 || (well_stride != 1 && well_stride != 4))
        return WD_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess)
        return WD_ERR_HIP;
    FILE *f = fopen(path, "rb");
    if (!f)
        return WD_ERR_IO;
    // header '<HIBBI' + bins + tile table (bcl_direct_reader.py:263-292)
    uint8_t head[12];
    auto bail = [&](int code) { fclose(f); return code; };
    if (fread(head, 1, 12, f) != 12)
        return bail(WD_ERR_FORMAT);
    uint16_t version; uint32_t hsize, bins;
    memcpy(&version, head, 2); memcpy(&hsize, head + 2, 4); memcpy(&bins, head + 8, 4);
    if (version != 1 || hsize <= 32 || head[6] != 2 || head[7] != 2 || bins != 4)   // :266-270
        return bail(WD_ERR_FORMAT);
    std::vector<uint8_t> tab((size_t)bins * 8 + 4);
    if (fread(tab.data(), 1, tab.size(), f) != tab.size())
        return bail(WD_ERR_FORMAT);
    uint32_t tile_count;
    memcpy(&tile_count, tab.data() + tab.size() - 4, 4);
    if (tile_count > (1u << 20))
        return bail(WD_ERR_FORMAT);
    std::vector<uint8_t> offs((size_t)tile_count * 16 + 1);
    if (fread(offs.data(), 1, offs.size(), f) != offs.size())
        return bail(WD_ERR_FORMAT);
    const int excluded = offs.back() ? 1 : 0;
    uint64_t pos = hsize;
    uint32_t usize = 0, csize = 0;
    bool found = false;
    for (uint32_t t = 0; t < tile_count; t++) {
        uint32_t rec[4];
        memcpy(rec, offs.data() + (size_t)t * 16, 16);
        if ((int)rec[0] == tile_number) {
            usize = rec[2];
            csize = rec[3];
            found = true;
            break;
        }
        pos += rec[3];
    }
    if (!found)
        return bail(WD_ERR_FORMAT);                          // assert t_number == tile_as_int (:295)
    std::vector<uint8_t> raw((size_t)csize + 16, 0);         // fast_gunzip reads in 8-byte words
    if (fseek(f, (long)pos, SEEK_SET) != 0 || (csize && fread(raw.data(), 1, csize, f) != csize))
        return bail(WD_ERR_IO);
    fclose(f);

    SlotLease lease(ctx);
    wd_ctx::IngestSlot *sl = lease.slot;
    int rc = slot_reserve(ctx, sl, (size_t)usize + 64 + kInflateSlack);
    if (rc)
        return rc;
    size_t produced = 0;
    if (!ctx->fast_inflate || !fast_gunzip(raw.data(), csize, sl->pinned, (size_t)usize + 274, &produced) ||
        produced > usize) {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK)
            return WD_ERR_NOMEM;
        zs.next_in = raw.data();
        zs.avail_in = csize;
        zs.next_out = sl->pinned;
        zs.avail_out = usize;                                 // GzipFile.read(t_usize) (:301)
        const int zr = inflate(&zs, Z_FINISH);
        produced = (size_t)(zs.next_out - sl->pinned);
        inflateEnd(&zs);
        if (zr != Z_STREAM_END && zr != Z_OK && zr != Z_BUF_ERROR)
            return WD_ERR_CORRUPT;
    }
    const long long n_records = (long long)produced * 2;
    const int chunks = (int)((n_clusters + kCbclChunk - 1) / kCbclChunk);
    const size_t need = ((produced + 255) & ~(size_t)255) + (size_t)std::max(chunks, 1) * 4;
    if (need > sl->dev_cap) {
        (void)hipFree(sl->dev);
        sl->dev = nullptr;
        sl->dev_cap = 0;
        if (hipMalloc((void **)&sl->dev, need) != hipSuccess)
            return WD_ERR_NOMEM;
        sl->dev_cap = need;
    }
    uint32_t *sums = (uint32_t *)(sl->dev + ((produced + 255) & ~(size_t)255));
    if (n_clusters == 0)
        return WD_OK;
    if (produced && hipMemcpyAsync(sl->dev, sl->pinned, produced, hipMemcpyHostToDevice, sl->stream) != hipSuccess)
        return WD_ERR_HIP;
    if (excluded) {
        hipLaunchKernelGGL(k_cbcl_count, dim3(chunks), dim3(kBlock), 0, sl->stream, filter_dev,
                           (long long)n_clusters, sums);
        hipLaunchKernelGGL(k_cbcl_scan, dim3(1), dim3(kBlock), 0, sl->stream, sums, chunks);
    }
    hipLaunchKernelGGL(k_cbcl_expand, dim3(chunks), dim3(kBlock), 0, sl->stream, sl->dev, n_records,
                       filter_dev, sums, (long long)n_clusters, excluded, dst_dev, well_stride);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(sl->stream) != hipSuccess)
        return WD_ERR_HIP;
    // the reference dies with IndexError only if a *requested* well lies beyond the block; a
    // block shorter than the tile is reported the same way here when no filter can excuse it
    if (!excluded && n_records < n_clusters)
        return WD_ERR_INDEX;
    return WD_OK;
} WD_CATCH

// ---- a batch of NovaSeq tile blocks through the GPU decoder -----------------------------------
// Entry i: the block of tile tile_number[i] in the .cbcl file paths[i] (all tiles of a surface share
// one file per cycle) -> an n_clusters-byte plane at dst_dev[i], exactly what wd_load_cbcl_tile does
// (bcl_direct_reader.py:255-325).  The files' headers and tile tables are parsed on the host (once
// per file), reader threads bring the tiles' gzip blocks into the pinned ring, the GPU inflates them
// (one launch for the batch) into packed planes in the arena and expands those (nibble -> byte, the
// excluded-wells indirection through the tile's filter, which must already be in filter_dev[i]).
// An entry the GPU decoder declines or whose checks fail (CRC-32, length) goes through
// wd_load_cbcl_tile, whose return code is reported; so do the table checks' failures.
static int load_cbcl_batch_impl(wd_ctx *ctx, int n, const char *const *paths, const int *tile_number,
                                const uint8_t *const *filter_dev, uint8_t *const *dst_dev, int64_t n_clusters, int well_stride,
                                int threads, int *rc_out);

int wd_load_cbcl_batch(wd_ctx *ctx, int n, const char *const *paths, const int *tile_number,
                       const uint8_t *const *filter_dev, uint8_t *const *dst_dev, int64_t n_clusters, int threads,
                       int *rc_out)
{
    return wd_load_cbcl_batch_strided(ctx, n, paths, tile_number, filter_dev, dst_dev, n_clusters, 1, threads, rc_out);
}

int wd_load_cbcl_batch_strided(wd_ctx *ctx, int n, const char *const *paths, const int *tile_number,
                               const uint8_t *const *filter_dev, uint8_t *const *dst_dev, int64_t n_clusters,
                               int well_stride, int threads, int *rc_out)
{
    return guarded(ctx, [&] {
        return load_cbcl_batch_impl(ctx, n, paths, tile_number, filter_dev, dst_dev, n_clusters, well_stride, threads, rc_out);
    });
}

static int load_cbcl_batch_impl(wd_ctx *ctx, int n, const char *const *paths, const int *tile_number,
                                const uint8_t *const *filter_dev, uint8_t *const *dst_dev, int64_t n_clusters, int well_stride,
                                int threads, int *rc_out)
{
    if (!ctx || n < 0 || (n && (!paths || !tile_number || !filter_dev || !dst_dev)) || n_clusters < 0 ||
        n_clusters > 0x7FFFFFF0ll || (well_stride != 1 && well_stride != 4))
        return WD_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess)
        return WD_ERR_HIP;
    InflateTurn batch_lock(ctx);
    wd_ctx::InflateSlot &slot = ctx->inflate_slots[batch_lock.ticket % wd_ctx::kInflateSlots];
    std::lock_guard<std::mutex> slot_lock(slot.mu);
    threads = reader_threads(threads);
    const bool ring_direct = getenv("WD_RING_DIRECT") && atoi(getenv("WD_RING_DIRECT")) != 0;
    constexpr int kChunks = wd_ctx::kInflateChunks, kStreams = wd_ctx::kInflateStreams;
    const size_t chunk_bytes = ctx->inflate_chunk_bytes;
    enum : int { PENDING = 1, HOST = 2 };
    struct Entry { uint64_t pos = 0; uint32_t usize = 0, csize = 0, stream_off = 0; int excluded = 0; };
    std::vector<Entry> ent((size_t)n);
    std::vector<int> rc((size_t)n, PENDING);
    // the tile tables, once per file (:263-295)
    {
        std::vector<int> order((size_t)n);
        for (int i = 0; i < n; i++)
            order[(size_t)i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) {
            const int c = paths[a] && paths[b] ? strcmp(paths[a], paths[b]) : (paths[a] ? 1 : 0) - (paths[b] ? 1 : 0);
            return c < 0 || (c == 0 && a < b);
        });
        for (size_t k = 0; k < order.size();) {
            size_t k1 = k;
            const char *path = paths[order[k]];
            while (k1 < order.size() && paths[order[k1]] && path && strcmp(paths[order[k1]], path) == 0)
                k1++;
            if (k1 == k)
                k1 = k + 1;
            int file_rc = WD_OK;
            std::vector<uint8_t> offs;
            uint32_t hsize = 0, tile_count = 0;
            FILE *f = path ? fopen(path, "rb") : nullptr;
            if (!path) {
                file_rc = WD_ERR_ARG;
            } else if (!f) {
                file_rc = WD_ERR_IO;
            } else {
                uint8_t head[12];
                uint16_t version = 0;
                uint32_t bins = 0;
                if (fread(head, 1, 12, f) != 12) {
                    file_rc = WD_ERR_FORMAT;
                } else {
                    memcpy(&version, head, 2);
                    memcpy(&hsize, head + 2, 4);
                    memcpy(&bins, head + 8, 4);
                    if (version != 1 || hsize <= 32 || head[6] != 2 || head[7] != 2 || bins != 4)   // :266-270
                        file_rc = WD_ERR_FORMAT;
                }
                if (file_rc == WD_OK) {
                    std::vector<uint8_t> tab((size_t)bins * 8 + 4);
                    if (fread(tab.data(), 1, tab.size(), f) != tab.size()) {
                        file_rc = WD_ERR_FORMAT;
                    } else {
                        memcpy(&tile_count, tab.data() + tab.size() - 4, 4);
                        if (tile_count > (1u << 20)) {                   // (before any allocation sized by it)
                            file_rc = WD_ERR_FORMAT;
                            tile_count = 0;
                        } else {
                            offs.resize((size_t)tile_count * 16 + 1);
                            if (fread(offs.data(), 1, offs.size(), f) != offs.size())
                                file_rc = WD_ERR_FORMAT;
                        }
                    }
                }
                fclose(f);
            }
            for (size_t q = k; q < k1; q++) {
                const int i = order[q];
                if (file_rc != WD_OK || !dst_dev[i] || !filter_dev[i]) {
                    rc[(size_t)i] = file_rc != WD_OK ? file_rc : WD_ERR_ARG;
                    continue;
                }
                uint64_t pos = hsize;
                bool found = false;
                for (uint32_t t = 0; t < tile_count; t++) {
                    uint32_t rec[4];
                    memcpy(rec, offs.data() + (size_t)t * 16, 16);
                    if ((int)rec[0] == tile_number[i]) {
                        ent[(size_t)i].pos = pos;
                        ent[(size_t)i].usize = rec[2];
                        ent[(size_t)i].csize = rec[3];
                        ent[(size_t)i].excluded = offs.back() ? 1 : 0;
                        found = true;
                        break;
                    }
                    pos += rec[3];
                }
                if (!found)
                    rc[(size_t)i] = WD_ERR_FORMAT;                       // assert t_number == tile_as_int (:295)
            }
            k = k1;
        }
    }
    // chunks of the ring, room in the arena: [compressed blocks][packed planes + chunk sums]
    const int exp_chunks = (int)((n_clusters + kCbclChunk - 1) / kCbclChunk);
    std::vector<size_t> offset((size_t)n, 0), out_at((size_t)n, 0);
    std::vector<int> group_of((size_t)n, -1);
    struct Group { int first, last; size_t bytes, arena_at; std::atomic<int> remaining{0}; };
    std::vector<std::unique_ptr<Group>> groups;
    size_t comp_bytes = 0, out_bytes = 0, n_jobs = 0;
    for (int i = 0; i < n; i++) {
        if (rc[(size_t)i] != PENDING)
            continue;
        const Entry &e = ent[(size_t)i];
        if ((size_t)e.csize + 32 > chunk_bytes || e.csize < 18 || e.usize == 0 || e.usize > 0x3FFFFFF0u) {
            rc[(size_t)i] = HOST;
            continue;
        }
        const size_t padded = ((size_t)e.csize + 15) & ~(size_t)15;
        if (groups.empty() || groups.back()->bytes + padded > chunk_bytes)
            groups.emplace_back(new Group{i, i, 0, comp_bytes});
        Group &g = *groups.back();
        offset[(size_t)i] = g.bytes;
        g.bytes += padded;
        comp_bytes += padded;
        g.last = i;
        g.remaining.fetch_add(1);
        group_of[(size_t)i] = (int)groups.size() - 1;
        out_at[(size_t)i] = out_bytes;
        out_bytes += (((size_t)e.usize + 255) & ~(size_t)255) + (size_t)std::max(exp_chunks, 1) * 4 + 252;
        n_jobs++;
    }
    const int n_groups = (int)groups.size();
    if (n_groups) {
        const int prc = inflate_prepare(ctx, slot, std::min(n_groups, kChunks), comp_bytes + out_bytes + 256, n_jobs);
        if (prc)
            return prc;
        if (hipStreamSynchronize(ctx->inflate_streams[kStreams]) != hipSuccess)
            return WD_ERR_HIP;
    }
    uint8_t *out_base = slot.arena ? slot.arena + ((comp_bytes + 255) & ~(size_t)255) : nullptr;
    std::vector<uint64_t> trailer((size_t)n, 0);
    std::mutex mu;
    std::condition_variable cv;
    int free_upto = kChunks;
    bool abort_all = false;
    std::atomic<int> next_file{0};
    auto reader = [&]() {
        for (;;) {
            const int i = next_file.fetch_add(1);
            if (i >= n)
                return;
            const int g = group_of[(size_t)i];
            if (g < 0)
                continue;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return abort_all || g < free_upto; });
                if (abort_all)
                    return;
            }
            Entry &e = ent[(size_t)i];
            uint8_t *dst = ctx->inflate_chunks[g % kChunks].pinned + offset[(size_t)i];
            bool ok = false;
            const int fd = open(paths[i], O_RDONLY);
            if (fd >= 0) {
                thread_local std::vector<uint8_t> bounce;
                ok = read_into_ring(fd, dst, e.csize, (off_t)e.pos, bounce, ring_direct) == e.csize;
                close(fd);
            }
            if (!ok || !inf_gzip_header(dst, e.csize, &e.stream_off))
                rc[(size_t)i] = HOST;
            else
                memcpy(&trailer[(size_t)i], dst + e.csize - 8, 8);
            if (groups[(size_t)g]->remaining.fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> lk(mu);
                cv.notify_all();
            }
        }
    };
    Crew pool;
    pool.wake = [&] {
        std::lock_guard<std::mutex> lk(mu);
        abort_all = true;
        cv.notify_all();
    };
    if (pool.start(std::min(threads, std::max(1, n)), reader, ctx->test_thread_limit) == 0)
        return WD_ERR_NOMEM;
    std::vector<int> job_file;
    job_file.reserve(n_jobs);
    int hip_rc = WD_OK;
    const int si = (int)(ctx->inflate_launch_seq++ % (unsigned)std::max(1, std::min(kStreams, ctx->inflate_decode_streams)));
    hipStream_t copy_stream = ctx->inflate_streams[kStreams], stream = ctx->inflate_streams[si];
    for (int g = 0; g < n_groups && hip_rc == WD_OK; g++) {
        Group &grp = *groups[(size_t)g];
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return grp.remaining.load() == 0; });
        }
        wd_ctx::InflateChunk &ch = ctx->inflate_chunks[g % kChunks];
        uint8_t *dev = slot.arena + grp.arena_at;
        for (int i = grp.first; i <= grp.last; i++) {
            if (group_of[(size_t)i] != g || rc[(size_t)i] != PENDING)
                continue;
            InfJob &j = slot.h_jobs[job_file.size()];
            j.file = reinterpret_cast<const uint32_t *>(dev + offset[(size_t)i]);
            j.obase = out_base + out_at[(size_t)i];
            j.file_bytes = ent[(size_t)i].csize;
            j.stream_off = ent[(size_t)i].stream_off;
            j.out_cap = ent[(size_t)i].usize;
            j.pad_ = 0;
            job_file.push_back(i);
        }
        if (hipMemcpyAsync(dev, ch.pinned, grp.bytes, hipMemcpyHostToDevice, copy_stream) != hipSuccess ||
            hipEventRecord(ch.copied, copy_stream) != hipSuccess) {
            hip_rc = WD_ERR_HIP;
            break;
        }
        if (g >= 1) {                                            // (as in wd_load_tile_files_batch)
            if (hipEventSynchronize(ctx->inflate_chunks[(g - 1) % kChunks].copied) != hipSuccess) {
                hip_rc = WD_ERR_HIP;
                break;
            }
            std::lock_guard<std::mutex> lk(mu);
            free_upto = g + kChunks;
            cv.notify_all();
        }
    }
    const unsigned nj = (unsigned)job_file.size();
    if (hip_rc == WD_OK && nj) {                             // one launch for the batch, then the expansions
        if (hipEventRecord(ctx->inflate_ready[si], copy_stream) != hipSuccess ||
            hipStreamWaitEvent(stream, ctx->inflate_ready[si], 0) != hipSuccess ||
            hipMemcpyAsync(slot.d_jobs, slot.h_jobs, sizeof(InfJob) * nj, hipMemcpyHostToDevice, stream) != hipSuccess)
            hip_rc = WD_ERR_HIP;
    }
    if (hip_rc == WD_OK && nj) {
        const int waves = ctx->inflate_waves ? ctx->inflate_waves : nj <= 256 ? 8 : 4;
        if (waves == 8)
            hipLaunchKernelGGL((k_inflate<8, 256>), dim3(nj), dim3(512), 0, stream, slot.d_jobs, slot.d_res);
        else if (waves == 4)
            hipLaunchKernelGGL((k_inflate<4, 256>), dim3(nj), dim3(256), 0, stream, slot.d_jobs, slot.d_res);
        else
            hipLaunchKernelGGL((k_inflate<1, 512>), dim3(nj), dim3(64), 0, stream, slot.d_jobs, slot.d_res);
        hipLaunchKernelGGL(k_inflate_crc, dim3(nj), dim3(256), 0, stream, slot.d_jobs, slot.d_res);
        hipLaunchKernelGGL(k_inflate_heads, dim3((nj + 255) / 256), dim3(256), 0, stream, slot.d_jobs, slot.d_res, (int)nj);
        // (the expansions take the table's block length on trust; the results below say whether it held)
        for (unsigned q = 0; q < nj && n_clusters > 0; q++) {
            const int i = job_file[q];
            const Entry &e = ent[(size_t)i];
            uint8_t *packed = out_base + out_at[(size_t)i];
            uint32_t *sums = (uint32_t *)(packed + (((size_t)e.usize + 255) & ~(size_t)255));
            if (e.excluded) {
                hipLaunchKernelGGL(k_cbcl_count, dim3(exp_chunks), dim3(kBlock), 0, stream, filter_dev[i], (long long)n_clusters,
                                   sums);
                hipLaunchKernelGGL(k_cbcl_scan, dim3(1), dim3(kBlock), 0, stream, sums, exp_chunks);
            }
            hipLaunchKernelGGL(k_cbcl_expand, dim3(exp_chunks), dim3(kBlock), 0, stream, packed, (long long)e.usize * 2,
                               filter_dev[i], sums, (long long)n_clusters, e.excluded, dst_dev[i], well_stride);
        }
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(slot.h_res, slot.d_res, sizeof(InfResult) * nj, hipMemcpyDeviceToHost, stream) != hipSuccess)
            hip_rc = WD_ERR_HIP;
    }
    if (hip_rc != WD_OK) {
        std::lock_guard<std::mutex> lk(mu);
        abort_all = true;
        cv.notify_all();
    }
    pool.join();
    if (hip_rc == WD_OK && n_groups && hipEventRecord(slot.done, stream) != hipSuccess)
        hip_rc = WD_ERR_HIP;
    if (hip_rc != WD_OK)
        (void)hipDeviceSynchronize();
    batch_lock.unlock();
    if (hip_rc == WD_OK && n_groups && hipEventSynchronize(slot.done) != hipSuccess)
        hip_rc = WD_ERR_HIP;
    if (hip_rc != WD_OK)
        return hip_rc;
    unsigned long long real_sum = 0;
    for (size_t j = 0; j < job_file.size(); j++) {
        const int i = job_file[j];
        const InfResult &r = slot.h_res[j];
        const Entry &e = ent[(size_t)i];
        const uint32_t crc = (uint32_t)trailer[(size_t)i], isize = (uint32_t)(trailer[(size_t)i] >> 32);
        const bool good = r.status == INF_OK && (size_t)r.end_byte + 8 == e.csize && crc == r.crc && isize == r.produced &&
                          r.produced == e.usize;
        // (a block shorter than the tile without excluded wells: the host path's IndexError)
        rc[(size_t)i] = good && (e.excluded || (long long)e.usize * 2 >= n_clusters) ? WD_OK : HOST;
        real_sum += r.t_real;
        if (!good && getenv("WD_INFLATE_STATS"))
            fprintf(stderr, "[wd inflate] block %d declined: status %u produced %u (table %u) end %u of %u crc %08x/%08x isize %u\n",
                    i, r.status, r.produced, e.usize, r.end_byte, e.csize, r.crc, crc, isize);
    }
    if (!job_file.empty())
        ctx->inflate_us_per_file = (long long)(real_sum / 100 / job_file.size());
    std::vector<int> todo;
    long long by_gpu = 0;
    for (int i = 0; i < n; i++) {
        if (rc[(size_t)i] == HOST || rc[(size_t)i] == PENDING)
            todo.push_back(i);
        by_gpu += rc[(size_t)i] == WD_OK;
    }
    ctx->inflate_files_gpu += by_gpu;
    ctx->inflate_files_host += (long long)todo.size();
    if (!todo.empty()) {
        std::atomic<size_t> next{0};
        auto host = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= todo.size())
                    return;
                const int i = todo[k];
                rc[(size_t)i] = wd_load_cbcl_tile_strided(ctx, paths[i], tile_number[i], filter_dev[i], n_clusters, dst_dev[i],
                                                          well_stride);
            }
        };
        Crew hp;
        if (hp.start((int)std::min((size_t)threads, todo.size()) - 1, host, ctx->test_thread_limit) >= 0)
            host();
        hp.join();
    }
    int first = WD_OK;
    for (int i = 0; i < n; i++) {
        if (rc_out)
            rc_out[i] = rc[(size_t)i];
        if (first == WD_OK && rc[(size_t)i] != WD_OK)
            first = rc[(size_t)i];
    }
    return first;
}

int wd_gather_wells(wd_ctx *ctx, const uint8_t *const *planes, int L, const int32_t *idx, int64_t n,
                    int64_t n_clusters, uint8_t *out_host)
try {
    if (!ctx || L < 0 || n < 0 || (n > 0 && L > 0 && (!planes || !idx || !out_host)))
        return fail(ctx, WD_ERR_ARG, "bad gather arguments");
    for (int64_t i = 0; i < n; i++)
        if (idx[i] < 0 || idx[i] >= n_clusters)
            return fail(ctx, WD_ERR_INDEX, "well index outside the tile");
    if (n == 0 || L == 0)
        return WD_OK;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    // a workspace that only grows: a hipFree per call would wait for every kernel in flight on the device
    // (the CLI calls this per tile, with the decoder of the next batches running)
    const size_t need = (((size_t)L * sizeof(void *) + 255) & ~(size_t)255) + (((size_t)n * 4 + 255) & ~(size_t)255) + (size_t)n * L;
    if (need > ctx->gather_cap) {
        WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_gather);
        ctx->d_gather = nullptr;
        ctx->gather_cap = 0;
        const size_t want = need + (need >> 1) + 4096;
        if (hipMalloc((void **)&ctx->d_gather, want) != hipSuccess)
            return fail(ctx, WD_ERR_NOMEM, "gather workspace");
        ctx->gather_cap = want;
    }
    const uint8_t **d_pl = (const uint8_t **)ctx->d_gather;
    int32_t *d_idx = (int32_t *)(ctx->d_gather + (((size_t)L * sizeof(void *) + 255) & ~(size_t)255));
    uint8_t *d_out = (uint8_t *)d_idx + (((size_t)n * 4 + 255) & ~(size_t)255);
    auto done = [&](int code, const char *msg) { return code == WD_OK ? WD_OK : fail(ctx, code, msg); };
    if (hipMemcpyAsync(d_pl, planes, (size_t)L * sizeof(void *), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d_idx, idx, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return done(WD_ERR_HIP, "gather upload");
    const long long total = (long long)n * L;
    hipLaunchKernelGGL(k_gather_wells, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       ctx->stream, d_pl, L, d_idx, (long long)n, d_out, ctx->well_stride);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(out_host, d_out, (size_t)total, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return done(WD_ERR_HIP, "gather kernel");
    return done(WD_OK, "");
} WD_CATCH


}  // extern "C"
