// welldup_core.hip - context, options, device memory helpers, hit log, profile, RCCL binding, synthetic
// data: the part of the C ABI (include/welldup.h) that launches no compare kernel.
//
// libwelldup.so is built from four translation units (one code object each, linked into one library):
//   welldup_core.hip    this file
//   welldup_scan.hip    targets, the sampled scans (k_scan, k_scan_q, k_scan_lines, k_scan_lev_generic),
//                       wd_scan_async / wd_count_tiles, the neighbour-index generator
//   welldup_dense.hip   the dense path (every well a centre): tables and the k_dense_* chain
//   welldup_ingest.hip  files -> HBM: DEFLATE on the GPU, the host loaders, CBCL, wd_gather_wells
// wd_ctx.h holds the context they share; the .inc files hold the kernels, each included by one unit.
#include "wd_ctx.h"

namespace {

#include "synth_kernels.inc"

// -------------------------------------------------------------------------------------
// RCCL, bound at run time
// -------------------------------------------------------------------------------------
struct Id128 { char b[WD_UNIQUE_ID_BYTES]; };   // ncclUniqueId, passed by value
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

bool rccl_load(std::string &err)
{
    if (g_rccl.handle)
        return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h)
            break;
    }
    if (!h) {
        err = std::string("cannot load librccl: ") + dlerror();
        return false;
    }
    auto sym = [&](const char *s) { return dlsym(h, s); };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
        err = "librccl lacks the nccl* entry points";
        return false;
    }
    g_rccl.handle = h;
    return true;
}

thread_local int g_create_status = WD_OK;     // of the calling thread's last wd_create

}  // namespace

namespace wd {

int fail(wd_ctx *ctx, int code, const std::string &msg)
{
    if (ctx)
        ctx->err = msg;
    return code;
}

int bind_device(wd_ctx *ctx)
{
    WD_HIP(ctx, hipSetDevice(ctx->device));
    return WD_OK;
}

void drain_events(wd_ctx *ctx)
{
    for (auto &ev : ctx->events) {
        float ms = 0.f;
        if (hipEventSynchronize(ev.second) == hipSuccess &&
            hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
            ctx->prof_ms += ms;
            ctx->prof_launches += 1;
        }
        ctx->free_events.push_back(ev);
    }
    ctx->events.clear();
}

}  // namespace wd

using namespace wd;

extern "C" {

int wd_version(void) { return 100; }

#ifndef WD_BUILD_ID
#define WD_BUILD_ID "unknown"
#endif
#ifndef WD_UNIT_ID
#define WD_UNIT_ID "unknown"
#endif
const char *wd_build_id(void)
{
    // "<all sources> core=<..> scan=<..> queue=<..> lines=<..> dense=<..> ingest=<..>": the whole tree's hash,
    // then one per translation unit (its .hip and everything it includes), so that evidence about a kernel
    // goes stale when ITS code changes and not when a comment in the ingest does
    static const std::string id = std::string(WD_BUILD_ID) + " core=" + WD_UNIT_ID + " scan=" + unit_id_scan() +
                                  " queue=" + unit_id_queue() + " lines=" + unit_id_lines() +
                                  " dense=" + unit_id_dense() + " ingest=" + unit_id_ingest();
    return id.c_str();
}

const char *wd_strerror(int code)
{
    switch (code) {
    case WD_OK: return "ok";
    case WD_ERR_ARG: return "invalid argument";
    case WD_ERR_INDEX: return "cluster index out of range for this tile";
    case WD_ERR_EMPTY_LEVEL: return "a valid target has an empty level";
    case WD_ERR_HIP: return "HIP runtime error";
    case WD_ERR_NOMEM: return "out of device memory";
    case WD_ERR_STATE: return "call out of order (targets not set?)";
    case WD_ERR_UNSUPPORTED: return "unsupported parameter combination";
    case WD_ERR_COMM: return "RCCL error";
    case WD_ERR_NO_WELLS: return "a cluster has no wells at some level";
    case WD_ERR_IO: return "cannot read file";
    case WD_ERR_FORMAT: return "file header does not match the tile";
    case WD_ERR_CORRUPT: return "compressed data is corrupt";
    case WD_ERR_TRUNCATED: return "compressed file ended before the end-of-stream marker";
    default: return "unknown error";
    }
}

const char *wd_last_error(const wd_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int wd_create_status(void) { return g_create_status; }

wd_ctx *wd_create(int device_id)
{
    g_create_status = WD_OK;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_status = WD_ERR_HIP;
        return nullptr;
    }
    if (device_id < 0) {
        if (hipGetDevice(&device_id) != hipSuccess)
            device_id = 0;
    }
    if (device_id >= ndev) {
        g_create_status = WD_ERR_ARG;
        return nullptr;
    }
    wd_ctx *ctx = new (std::nothrow) wd_ctx();
    if (!ctx) {
        g_create_status = WD_ERR_NOMEM;
        return nullptr;
    }
    ctx->device = device_id;
    if (const char *fi = getenv("WD_FAST_INFLATE"))         // default of the "fast_inflate" option
        ctx->fast_inflate = atoi(fi) ? 1 : 0;
    if (hipSetDevice(device_id) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&ctx->d_status, sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&ctx->d_tblflags, 4 * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&ctx->d_rare, sizeof(ScanRare)) != hipSuccess ||
        hipHostMalloc((void **)&ctx->h_status, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&ctx->d_hit_count, sizeof(unsigned long long)) != hipSuccess) {
        g_create_status = WD_ERR_HIP;
        delete ctx;
        return nullptr;
    }
    ctx->stream = ctx->own_stream;
    (void)hipMemsetAsync(ctx->d_status, 0, sizeof(uint32_t), ctx->stream);
    (void)hipMemsetAsync(ctx->d_hit_count, 0, sizeof(unsigned long long), ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    return ctx;
}

void wd_destroy(wd_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->fast_exit) {
        // The process ends next ("fast_exit"): everything queued anywhere on the device is waited for, and
        // that is all - unpinning the ingest ring, destroying a dozen streams and freeing the scratch one by
        // one cost 25-28 ms of a lane's 0.39 s, and the driver reclaims it all at exit.
        (void)hipDeviceSynchronize();
        if (ctx->comm && g_rccl.CommDestroy)
            g_rccl.CommDestroy(ctx->comm);
        delete ctx;
        return;
    }
    // (WD_INFLATE_STATS: where the time to close a context goes)
    const bool lap_on = getenv("WD_INFLATE_STATS") != nullptr;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!lap_on)
            return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[wd close] %-28s %6.1f ms\n", what, 1e3 * std::chrono::duration<double>(now - lap_t).count());
        lap_t = now;
    };
    drain_events(ctx);
    for (auto &ev : ctx->free_events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    if (ctx->comm && g_rccl.CommDestroy)
        g_rccl.CommDestroy(ctx->comm);
    (void)hipFree(ctx->d_centre);
    (void)hipFree(ctx->d_lvl_off);
    (void)hipFree(ctx->d_nbr);
    (void)hipFree(ctx->d_tbl);
    (void)hipFree(ctx->d_status);
    (void)hipFree(ctx->d_rare);
    (void)hipFree(ctx->d_nbr_t);
    (void)hipFree(ctx->d_rel_t);
    (void)hipFree(ctx->d_udelta);
    (void)hipFree(ctx->d_guni);
    (void)hipFree(ctx->d_ginfo);
    (void)hipFree(ctx->d_uoff);
    (void)hipFree(ctx->d_useg);
    (void)hipFree(ctx->d_wdelta);
    (void)hipFree(ctx->d_wlev);
    (void)hipFree(ctx->d_wmask);
    (void)hipFree(ctx->d_wfull);
    (void)hipFree(ctx->d_tblflags);
    (void)hipFree(ctx->d_mark);
    (void)hipFree(ctx->d_sig);
    (void)hipFree(ctx->d_partial);
    (void)hipFree(ctx->d_mask);
    (void)hipFree(ctx->d_queue);
    (void)hipFree(ctx->d_qcnt);
    (void)hipFree(ctx->d_cand);
    (void)hipFree(ctx->d_pblocks);
    (void)hipFree(ctx->d_centre_q);
    (void)hipFree(ctx->d_lvl_off_q);
    (void)hipFree(ctx->d_perm);
    drop_line_tables(ctx);
    for (hipStream_t q : {ctx->dense_hi, ctx->dense_lo})
        if (q) {
            (void)hipStreamSynchronize(q);
            (void)hipStreamDestroy(q);
        }
    for (hipEvent_t e : {ctx->dense_ev_start, ctx->dense_ev_done, ctx->dense_ev_cmp[0], ctx->dense_ev_cmp[1],
                         ctx->dense_ev_pack[0], ctx->dense_ev_pack[1]})
        if (e)
            (void)hipEventDestroy(e);
    (void)hipFree(ctx->d_rows);
    (void)hipFree(ctx->d_gbase);
    (void)hipHostFree(ctx->h_status);
    (void)hipFree(ctx->d_out_tile);
    (void)hipFree(ctx->d_stage);
    (void)hipFree(ctx->d_out_pt);
    (void)hipFree(ctx->d_hits);
    (void)hipFree(ctx->d_gather);
    (void)hipFree(ctx->d_hit_count);
    lap("scan buffers");
    for (auto &st : ctx->inflate_streams)
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
    for (auto &ev : ctx->inflate_joined)
        if (ev)
            (void)hipEventDestroy(ev);
    for (auto &ev : ctx->inflate_ready)
        if (ev)
            (void)hipEventDestroy(ev);
    lap("ingest streams, events");
    for (auto &ch : ctx->inflate_chunks) {
        (void)hipHostFree(ch.pinned);
        if (ch.copied)
            (void)hipEventDestroy(ch.copied);
    }
    lap("pinned ring");
    for (auto &sl : ctx->inflate_slots) {
        (void)hipFree(sl.arena);
        (void)hipHostFree(sl.h_jobs);
        (void)hipFree(sl.d_jobs);
        (void)hipHostFree(sl.h_res);
        (void)hipFree(sl.d_res);
        if (sl.done)
            (void)hipEventDestroy(sl.done);
    }
    lap("arenas, job tables");
    for (auto *sl : ctx->ingest_slots) {
        (void)hipHostFree(sl->pinned);
        (void)hipFree(sl->dev);
        free(sl->file);
        delete sl;
    }
    for (auto &st : ctx->slot_streams)
        if (st)
            (void)hipStreamDestroy(st);
    if (ctx->own_stream)
        (void)hipStreamDestroy(ctx->own_stream);
    lap("host loader slots, streams");
    delete ctx;
}

int wd_set_stream(wd_ctx *ctx, void *hip_stream)
{
    if (!ctx)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return WD_OK;
}

int wd_synchronize(wd_ctx *ctx)
{
    if (!ctx)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WD_OK;
}

int wd_set_option(wd_ctx *ctx, const char *name, int64_t value)
try {
    if (!ctx || !name)
        return WD_ERR_ARG;
    std::string n(name);
    if (n == "early_exit") {
        ctx->early_exit = value ? 1 : 0;
    } else if (n == "targets_per_block") {
        if (value < 1 || value > kMaxTpb)
            return fail(ctx, WD_ERR_ARG, "targets_per_block must be 1..64");
        ctx->tpb = (int)value;
    } else if (n == "batch_first") {
        ctx->batch_first = (int)value;
    } else if (n == "batch_next") {
        ctx->batch_next = (int)value;
    } else if (n == "profile") {
        ctx->profile = value < 0 ? 0 : (int)std::min<int64_t>(value, 1 << 20);   // n: every n-th scan
        ctx->profile_seq = 0;
    } else if (n == "null_stream") {
        // run on the HIP null (legacy default) stream, e.g. to order with a framework that
        // uses it; 0 returns to the context's own stream
        if (bind_device(ctx))
            return WD_ERR_HIP;
        WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->stream = value ? (hipStream_t) nullptr : ctx->own_stream;
    } else if (n == "fast_exit") {
        ctx->fast_exit = value ? 1 : 0;
    } else if (n == "dense_kernel") {
        ctx->dense_kernel = value < 0 ? -1 : (value ? 1 : 0);
    } else if (n == "dense_tile_chunk") {
        if (value < 1 || value > 1024)
            return WD_ERR_ARG;
        ctx->dense_tile_chunk = (int)value;
    } else if (n == "well_stride") {
        if (value != 1 && value != 4)
            return WD_ERR_ARG;
        ctx->well_stride = (int)value;
    } else if (n == "fast_inflate") {
        ctx->fast_inflate = value ? 1 : 0;
    } else if (n == "dense_overlap") {
        ctx->dense_overlap = value ? 1 : 0;
    } else if (n == "dense_pack_blocks") {
        if (value < 0 || value > (1 << 24))
            return WD_ERR_ARG;
        ctx->dense_pack_blocks = (int)value;
    } else if (n == "dense_part_tiles") {
        if (value < 0 || value > 65535)
            return WD_ERR_ARG;
        ctx->dense_part_tiles = (int)value;
    } else if (n == "line_walk") {
        ctx->line_walk = value < 0 ? -1 : (value ? 1 : 0);
    } else if (n == "line_pairs") {
        if (value < 0 || value > (1 << 20))
            return WD_ERR_ARG;
        if (value != ctx->line_pairs) {
            WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            drop_line_tables(ctx);
        }
        ctx->line_pairs = (int)value;
    } else if (n == "sort_targets") {
        ctx->sort_targets = value ? 1 : 0;
    } else if (n == "sort_strip") {
        if (value < 0 || value > (1 << 20))
            return WD_ERR_ARG;
        ctx->sort_strip = (int)value;            // (takes effect with the next set of targets)
    } else if (n == "lev2_closed") {
        ctx->lev2_closed = value ? 1 : 0;
    } else if (n == "test_thread_limit") {
        ctx->test_thread_limit = value;
    } else if (n == "inflate_waves") {
        if (value != 0 && value != 1 && value != 4 && value != 8)
            return WD_ERR_ARG;
        ctx->inflate_waves = (int)value;
    } else if (n == "inflate_chunk_mb") {
        if (value < 1 || value > 1024)
            return WD_ERR_ARG;
        ctx->inflate_chunk_bytes = (size_t)value << 20;
    } else if (n == "inflate_warm") {
        // the batch loaders' pinned ring, streams and events now, not inside the first batch (callable from
        // a thread of its own while the caller parses its targets file)
        if (value) {
            WD_HIP(ctx, hipSetDevice(ctx->device));
            return inflate_prepare_shared(ctx, wd_ctx::kInflateChunks);
        }
    } else if (n == "dense_pack") {
        ctx->dense_pack = value < 0 ? -1 : (value ? 1 : 0);
    } else if (n == "dense_windows") {
        ctx->dense_windows = value ? 1 : 0;
    } else if (n == "dense_sym") {
        if (ctx->dense_sym != (value ? 1 : 0)) {        // the window tables are built for one or the other
            WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            drop_dense_tables(ctx);
        }
        ctx->dense_sym = value ? 1 : 0;
    } else if (n == "dense_nt") {
        ctx->dense_nt = value ? 1 : 0;
    } else if (n == "dense_queue_cap") {
        if (value < 0)
            return WD_ERR_ARG;
        ctx->dense_queue_cap = value;
    } else if (n == "queue_kernel") {
        ctx->queue_kernel = value ? 1 : 0;
    } else if (n == "queue_first") {
        if (value < 0 || value > 8)
            return fail(ctx, WD_ERR_ARG, "queue_first must be 0 (auto) or 1..8");
        ctx->queue_first = (int)value;
    } else {
        return fail(ctx, WD_ERR_ARG, "unknown option " + n);
    }
    return WD_OK;
} WD_CATCH

int wd_get_option(wd_ctx *ctx, const char *name, int64_t *value)
try {
    if (!ctx || !name || !value)
        return WD_ERR_ARG;
    std::string n(name);
    if (n == "early_exit") *value = ctx->early_exit;
    else if (n == "targets_per_block") *value = ctx->tpb;
    else if (n == "batch_first") *value = ctx->batch_first;
    else if (n == "batch_next") *value = ctx->batch_next;
    else if (n == "profile") *value = ctx->profile;
    else if (n == "queue_kernel") *value = ctx->queue_kernel;
    else if (n == "fast_exit") *value = ctx->fast_exit;
    else if (n == "hitlog_capacity") *value = ctx->hit_cap;         // read-only: records wd_hitlog_fetch can return
    else if (n == "dense_kernel") *value = ctx->dense_kernel;
    else if (n == "dense_tile_chunk") *value = ctx->dense_tile_chunk;
    else if (n == "dense_queue_cap") *value = ctx->dense_queue_cap;
    else if (n == "dense_pack") *value = ctx->dense_pack;
    else if (n == "dense_windows") *value = ctx->dense_windows;
    else if (n == "dense_sym") *value = ctx->dense_sym;
    else if (n == "dense_sym_on") *value = ctx->dense_sym_on ? 1 : 0;      // read-only: the tables built last are one-ended
    else if (n == "dense_nt") *value = ctx->dense_nt;
    else if (n == "fast_inflate") *value = ctx->fast_inflate;
    else if (n == "inflate_chunk_mb") *value = (long long)(ctx->inflate_chunk_bytes >> 20);
    else if (n == "inflate_waves") *value = ctx->inflate_waves;
    else if (n == "test_thread_limit") *value = ctx->test_thread_limit;
    else if (n == "lev2_closed") *value = ctx->lev2_closed;
    else if (n == "sort_targets") *value = ctx->sort_targets;
    else if (n == "line_walk") *value = ctx->line_walk;
    else if (n == "line_pairs") *value = ctx->line_pairs;
    else if (n == "line_walk_blocks") *value = ctx->lw_blocks;      // read-only: blocks of the line walk's tables (-1: not built, 0: does not apply)
    else if (n == "sort_strip") *value = ctx->sort_strip;
    else if (n == "dense_overlap") *value = ctx->dense_overlap;
    else if (n == "dense_part_tiles") *value = ctx->dense_part_tiles;
    else if (n == "dense_pack_blocks") *value = ctx->dense_pack_blocks;
    else if (n == "inflate_files_gpu") *value = ctx->inflate_files_gpu.load();
    else if (n == "inflate_files_host") *value = ctx->inflate_files_host.load();
    else if (n == "inflate_files_early") *value = ctx->inflate_files_early.load();
    else if (n == "inflate_us_per_file") *value = ctx->inflate_us_per_file.load();
    else if (n == "well_stride") *value = ctx->well_stride;
    else if (n == "null_stream") *value = ctx->stream == nullptr ? 1 : 0;
    else if (n == "queue_first") *value = ctx->queue_first;
    // read-only, -1 before the first dense scan of the current targets: 64-target groups of
    // consecutive centres with common neighbour offsets, and those scanned through LDS windows
    else if (n == "dense_uniform_groups") *value = ctx->n_uniform_groups;
    else if (n == "dense_window_groups") *value = ctx->n_window_groups;
    else if (n == "dense_window_dwords") *value = ctx->win_dwords;
    else return fail(ctx, WD_ERR_ARG, "unknown option " + n);
    return WD_OK;
} WD_CATCH

int wd_malloc(wd_ctx *ctx, size_t bytes, void **out_dev)
try {
    if (!ctx || !out_dev)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    *out_dev = nullptr;
    WD_HIP(ctx, hipMalloc(out_dev, bytes ? bytes : 1));
    return WD_OK;
} WD_CATCH

int wd_free(wd_ctx *ctx, void *dev)
{
    if (!ctx)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    WD_HIP(ctx, hipFree(dev));
    return WD_OK;
}

int wd_memcpy_h2d(wd_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    if (!ctx)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    WD_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WD_OK;
}

int wd_memcpy_d2h(wd_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    if (!ctx)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    WD_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return WD_OK;
}

int wd_memset(wd_ctx *ctx, void *dst_dev, int value, size_t bytes)
{
    if (!ctx)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    WD_HIP(ctx, hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
    return WD_OK;
}

int wd_hitlog_enable(wd_ctx *ctx, int64_t capacity)
try {
    if (!ctx || capacity < 0)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    // The buffer only grows: a caller that switches the log on and off around every batch (the CLI does)
    // must not pay a hipFree - which waits for every kernel in flight on the device, the decoder's
    // included - and a hipMalloc each time.  Capacity 0 switches the log off and keeps the memory.
    if ((size_t)capacity > ctx->hit_alloc) {
        WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_hits);
        ctx->d_hits = nullptr;
        ctx->hit_alloc = 0;
        ctx->hit_cap = 0;
        WD_HIP(ctx, hipMalloc((void **)&ctx->d_hits, (size_t)capacity * sizeof(wd_hit)));
        ctx->hit_alloc = (size_t)capacity;
    }
    ctx->hit_cap = capacity;
    WD_HIP(ctx, hipMemsetAsync(ctx->d_hit_count, 0, sizeof(unsigned long long), ctx->stream));
    return WD_OK;
} WD_CATCH

int wd_hitlog_fetch(wd_ctx *ctx, wd_hit *out_host, int64_t max_records, int64_t *total_out)
try {
    if (!ctx || max_records < 0)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    unsigned long long total = 0;
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    WD_HIP(ctx, hipMemcpy(&total, ctx->d_hit_count, sizeof(total), hipMemcpyDeviceToHost));
    if (total_out)
        *total_out = (int64_t)total;
    int64_t n = std::min<int64_t>((int64_t)total, std::min<int64_t>(max_records, ctx->hit_cap));
    if (n > 0 && out_host)
        WD_HIP(ctx, hipMemcpy(out_host, ctx->d_hits, (size_t)n * sizeof(wd_hit), hipMemcpyDeviceToHost));
    return WD_OK;
} WD_CATCH

int wd_profile_get(wd_ctx *ctx, double *total_ms, int64_t *launches)
{
    if (!ctx)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    drain_events(ctx);
    if (total_ms)
        *total_ms = ctx->prof_ms;
    if (launches)
        *launches = ctx->prof_launches;
    return WD_OK;
}

const char *wd_last_kernel(const wd_ctx *ctx) { return ctx ? ctx->last_kernel : ""; }

int wd_profile_reset(wd_ctx *ctx)
{
    if (!ctx)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    drain_events(ctx);
    ctx->prof_ms = 0.0;
    ctx->prof_launches = 0;
    return WD_OK;
}

// ---- what this GPU streams ------------------------------------------------------------
// A read of `bytes` of device memory and nothing else: 16 bytes per lane and load, eight loads in flight per
// thread, non-temporal, one word written per workgroup whose bytes XOR to something (so the loads stay).
// The rate is what the HBM of THIS box gives a kernel that does no work - the boxes of a pool differ by
// several percent, and a roofline fraction is read against it (bench.py: roofline.stream_read).
__global__ __launch_bounds__(kBlock) void k_stream_read(const uint4 *src, size_t n16, uint32_t *sink)
{
    typedef uint32_t U4 __attribute__((ext_vector_type(4)));
    const __attribute__((address_space(1))) U4 *p = (const __attribute__((address_space(1))) U4 *)src;
    U4 acc = {0u, 0u, 0u, 0u};
    const size_t step = (size_t)gridDim.x * kBlock;
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 7 * step < n16; i += 8 * step) {
        U4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
            v[j] = __builtin_nontemporal_load(p + i + j * step);
#pragma unroll
        for (int j = 0; j < 8; j++)
            acc ^= v[j];
    }
    for (; i < n16; i += step)
        acc ^= __builtin_nontemporal_load(p + i);
    const uint32_t x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x9E3779B9u)                            // (never, for all that matters: the sink is there to be written to)
        sink[blockIdx.x & 63] = x;
}

int wd_stream_read_probe(wd_ctx *ctx, const void *src_dev, size_t bytes, int passes, double *ms_per_pass)
try {
    if (!ctx || !src_dev || !ms_per_pass || passes < 1 || bytes < 16 || ((uintptr_t)src_dev & 15u))
        return fail(ctx, WD_ERR_ARG, "wd_stream_read_probe: a 16-byte aligned device buffer, passes >= 1");
    if (bind_device(ctx))
        return WD_ERR_HIP;
    uint32_t *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    WD_HIP(ctx, hipMalloc((void **)&sink, 64 * sizeof(uint32_t)));
    WD_HIP(ctx, hipEventCreate(&e0));
    WD_HIP(ctx, hipEventCreate(&e1));
    const size_t n16 = bytes / 16;
    // (workgroups: enough for every CU's wave slots several times over, few enough that each streams megabytes)
    const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>(1, n16 / (8 * kBlock)), 256 * 64);
    hipLaunchKernelGGL(k_stream_read, dim3(grid), dim3(kBlock), 0, ctx->stream, (const uint4 *)src_dev, n16, sink);   // warm-up
    WD_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (int r = 0; r < passes; r++)
        hipLaunchKernelGGL(k_stream_read, dim3(grid), dim3(kBlock), 0, ctx->stream, (const uint4 *)src_dev, n16, sink);
    WD_HIP(ctx, hipEventRecord(e1, ctx->stream));
    WD_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    WD_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    *ms_per_pass = (double)ms / passes;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    return WD_OK;
} WD_CATCH

// ---- RCCL ----------------------------------------------------------------------------
int wd_comm_unique_id(void *out128)
{
    std::string err;
    if (!out128 || !rccl_load(err))
        return WD_ERR_COMM;
    return g_rccl.GetUniqueId(out128) == 0 ? WD_OK : WD_ERR_COMM;
}

int wd_comm_init(wd_ctx *ctx, int rank, int world, const void *id128)
try {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world)
        return WD_ERR_ARG;
    std::string err;
    if (!rccl_load(err))
        return fail(ctx, WD_ERR_COMM, err);
    if (bind_device(ctx))
        return WD_ERR_HIP;
    if (ctx->comm) {
        g_rccl.CommDestroy(ctx->comm);
        ctx->comm = nullptr;
    }
    Id128 id;
    memcpy(id.b, id128, WD_UNIQUE_ID_BYTES);
    int rc = g_rccl.CommInitRank(&ctx->comm, world, id, rank);
    if (rc != 0)
        return fail(ctx, WD_ERR_COMM, std::string("ncclCommInitRank: ") +
                                          (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
    return WD_OK;
} WD_CATCH

int wd_allreduce_counts(wd_ctx *ctx, int64_t *buf_dev, size_t n)
{
    if (!ctx || (!buf_dev && n))
        return WD_ERR_ARG;
    if (!ctx->comm)
        return fail(ctx, WD_ERR_STATE, "wd_comm_init has not been called");
    if (bind_device(ctx))
        return WD_ERR_HIP;
    // ncclInt64 = 4, ncclSum = 0
    int rc = g_rccl.AllReduce(buf_dev, buf_dev, n, 4, 0, ctx->comm, ctx->stream);
    if (rc != 0)
        return fail(ctx, WD_ERR_COMM, std::string("ncclAllReduce: ") +
                                          (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
    return WD_OK;
}

int wd_comm_destroy(wd_ctx *ctx)
{
    if (!ctx)
        return WD_ERR_ARG;
    if (ctx->comm && g_rccl.CommDestroy) {
        (void)hipStreamSynchronize(ctx->stream);
        g_rccl.CommDestroy(ctx->comm);
    }
    ctx->comm = nullptr;
    return WD_OK;
}

// ---- synthetic data ------------------------------------------------------------------
static uint64_t synth_tile_key(const wd_synth_spec *s, int lane, int tile, uint64_t salt)
{
    return mix64(s->seed * K_SEED + (uint64_t)lane * K_LANE + (uint64_t)tile * K_TILE + salt);
}

int wd_synth_plane(wd_ctx *ctx, uint8_t *dst_dev, const wd_synth_spec *spec, int lane, int tile, int cycle)
{
    if (!ctx || !dst_dev || !spec || spec->n_clusters < 0)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    SynthPlaneArgs a;
    a.dst = dst_dev;
    a.n = spec->n_clusters;
    a.row = spec->row;
    a.key_here = synth_tile_key(spec, lane, tile, (uint64_t)(cycle + 1) * K_CYCLE);
    a.key_next = synth_tile_key(spec, lane, tile, (uint64_t)(cycle + 2) * K_CYCLE);
    a.key_plant = synth_tile_key(spec, lane, tile, SALT_PLANT);
    a.nocall = spec->nocall_per_64k;
    a.plant = spec->plant_per_64k;
    a.far = spec->plant_far;
    a.qlev = spec->qual_levels ? spec->qual_levels : 39u;
    a.cycle = cycle;
    if (a.n == 0)
        return WD_OK;
    unsigned blocks = (unsigned)std::min<int64_t>((a.n + kBlock - 1) / kBlock, 256 * 16);
    hipLaunchKernelGGL(k_synth_plane, dim3(blocks), dim3(kBlock), 0, ctx->stream, a);
    WD_HIP(ctx, hipGetLastError());
    return WD_OK;
}

int wd_synth_filter(wd_ctx *ctx, uint8_t *dst_dev, const wd_synth_spec *spec, int lane, int tile)
{
    if (!ctx || !dst_dev || !spec || spec->n_clusters < 0)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    SynthFilterArgs a;
    a.dst = dst_dev;
    a.n = spec->n_clusters;
    a.key = synth_tile_key(spec, lane, tile, SALT_FILTER);
    a.pass = spec->pass_per_64k;
    a.noise = spec->filter_noise;
    a.dead = spec->tile_dead;
    if (a.n == 0)
        return WD_OK;
    unsigned blocks = (unsigned)std::min<int64_t>((a.n + kBlock - 1) / kBlock, 256 * 16);
    hipLaunchKernelGGL(k_synth_filter, dim3(blocks), dim3(kBlock), 0, ctx->stream, a);
    WD_HIP(ctx, hipGetLastError());
    return WD_OK;
}

}  // extern "C"

