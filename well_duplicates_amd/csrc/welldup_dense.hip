// welldup_dense.hip - the dense path (BASELINE configs[4]: every well a centre): its device tables and
// the chain of k_dense_* kernels (scan_dense.inc).  Called from wd_scan_async (welldup_scan.hip).
#include "wd_ctx.h"

#ifndef WD_UNIT_ID
#define WD_UNIT_ID "unknown"
#endif
namespace wd { const char *unit_id_dense() { return WD_UNIT_ID; } }      // hash of this unit's sources (wd_build_id)

namespace {

#include "device_common.inc"
#include "lev2_stream.inc"
#include "scan_dense.inc"

}  // namespace

namespace wd {

// Group bases of the transposed neighbour table from host-side ring offsets (row = levels+1).
void set_group_bases(wd_ctx *ctx, const int32_t *lvl_off, int T, int levels)
{
    const size_t row = (size_t)levels + 1;
    const int groups = (T + kWave - 1) / kWave;
    ctx->h_gbase.assign((size_t)groups + 1, 0);
    long long pos = 0;
    for (int g = 0; g < groups; g++) {
        int kmax = 1;                                  // at least one row so min(q, gk-1) is valid
        for (int t = g * kWave; t < std::min(T, (g + 1) * kWave); t++)
            kmax = std::max(kmax, lvl_off[(size_t)t * row + levels] - lvl_off[(size_t)t * row]);
        ctx->h_gbase[g] = pos;
        pos += (long long)kmax * kWave;
    }
    ctx->h_gbase[groups] = pos;
    drop_dense_tables(ctx);
}

// The dense path's device tables are built on its first scan after this (ensure_dense_tables).
void drop_dense_tables(wd_ctx *ctx)
{
    drop_line_tables(ctx);                             // (new targets - or an option of the dense tables: rebuilt on demand)
    (void)hipFree(ctx->d_nbr_t);
    (void)hipFree(ctx->d_gbase);
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
    ctx->d_wfull = nullptr;
    ctx->dense_sym_on = false;
    ctx->d_wlev = nullptr;
    ctx->d_udelta = nullptr;
    ctx->d_guni = nullptr;
    ctx->d_ginfo = nullptr;
    ctx->d_uoff = nullptr;
    ctx->d_useg = nullptr;
    ctx->d_wdelta = nullptr;
    ctx->d_wmask = nullptr;
    ctx->win_kpad = ctx->win_dwords = 0;
    ctx->n_uniform_groups = ctx->n_window_groups = -1;
    ctx->d_nbr_t = nullptr;
    ctx->d_gbase = nullptr;
    ctx->d_rel_t = nullptr;
}

// Build the device copy of the transposed table (first dense scan after new targets).
int ensure_dense_tables(wd_ctx *ctx)
{
    if (ctx->d_nbr_t)
        return WD_OK;
    const int groups = (ctx->T + kWave - 1) / kWave;
    const long long total = ctx->h_gbase.empty() ? 0 : ctx->h_gbase.back();
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_gbase, (size_t)(groups + 1) * sizeof(long long)));
    WD_HIP(ctx, hipMemcpyAsync(ctx->d_gbase, ctx->h_gbase.data(), (size_t)(groups + 1) * sizeof(long long),
                               hipMemcpyHostToDevice, ctx->stream));
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_rel_t, std::max<size_t>(1, (size_t)ctx->T * ctx->levels) * sizeof(int32_t)));
    if (ctx->T > 0)
        hipLaunchKernelGGL(k_transpose_off, dim3((ctx->T + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream,
                           ctx->d_lvl_off, ctx->d_rel_t, ctx->T, ctx->levels);
    WD_HIP(ctx, hipMalloc(&ctx->d_udelta, std::max<size_t>(1, (size_t)(total / kWave)) * sizeof(int32_t)));
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_guni, std::max<size_t>(1, (size_t)groups)));
    // int16 offsets from the centre if they all fit (half the index stream), else int32 indices
    const size_t n_el = (size_t)std::max<long long>(1, total);
    WD_HIP(ctx, hipMalloc(&ctx->d_nbr_t, n_el * sizeof(int16_t)));
    ctx->nbr_t16 = true;
    if (groups > 0) {
        uint32_t flags[2] = {0, 0};
        WD_HIP(ctx, hipMemsetAsync(ctx->d_tblflags, 0, sizeof(flags), ctx->stream));
        hipLaunchKernelGGL((k_transpose_nbr<int16_t>), dim3(groups), dim3(kWave), 0, ctx->stream, ctx->d_centre,
                           ctx->d_lvl_off, ctx->d_nbr, ctx->d_gbase, (int16_t *)ctx->d_nbr_t, ctx->T, ctx->levels,
                           ctx->d_tblflags, (int16_t *)ctx->d_udelta, ctx->d_guni);
        WD_HIP(ctx, hipMemcpyAsync(flags, ctx->d_tblflags, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (flags[0]) {
            (void)hipFree(ctx->d_nbr_t);
            ctx->d_nbr_t = nullptr;
            ctx->nbr_t16 = false;
            WD_HIP(ctx, hipMalloc(&ctx->d_nbr_t, n_el * sizeof(int32_t)));
            hipLaunchKernelGGL((k_transpose_nbr<int32_t>), dim3(groups), dim3(kWave), 0, ctx->stream, ctx->d_centre,
                               ctx->d_lvl_off, ctx->d_nbr, ctx->d_gbase, (int32_t *)ctx->d_nbr_t, ctx->T,
                               ctx->levels, ctx->d_tblflags, (int32_t *)ctx->d_udelta, ctx->d_guni);
        }
        // groups of consecutive centres whose neighbours fall into a few runs of offsets: LDS
        // windows (k_dense_windows); the union of a group's offsets may be a little larger than
        // any one target's list
        // (a group that straddles the end of a grid row sees two patterns: up to twice the offsets)
        // Is the neighbour relation symmetric (every well a centre, b in a's rings <=> a in b's)?  Then every
        // pair is compared once, from its lower well, and recorded at both ends (DenseArgs::sym)
        ctx->dense_sym_on = false;
        if (ctx->dense_sym && ctx->T >= 2) {
            WD_HIP(ctx, hipMemsetAsync(ctx->d_tblflags + 2, 0, sizeof(uint32_t), ctx->stream));
            hipLaunchKernelGGL(k_dense_symcheck, dim3((ctx->T + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream,
                               ctx->d_centre, ctx->d_lvl_off, ctx->d_nbr, ctx->T, ctx->levels, ctx->d_tblflags + 2);
            uint32_t bad = 1;
            int32_t c0 = 0;
            WD_HIP(ctx, hipMemcpyAsync(&bad, ctx->d_tblflags + 2, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream));
            WD_HIP(ctx, hipMemcpyAsync(&c0, ctx->d_centre, sizeof(c0), hipMemcpyDeviceToHost, ctx->stream));
            WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->dense_sym_on = bad == 0;
            ctx->centre0 = c0;
        }
        const int64_t kpad = std::min<int64_t>(kWinMaxK, (2 * ctx->k_max + 8 + 31) & ~(int64_t)31);
        if (ctx->k_max >= 1 && ctx->k_max <= kpad) {
            ctx->win_kpad = (int)kpad;
            WD_HIP(ctx, hipMalloc((void **)&ctx->d_ginfo, (size_t)groups * sizeof(int32_t)));
            WD_HIP(ctx, hipMalloc((void **)&ctx->d_uoff, (size_t)groups * kpad * sizeof(uint16_t)));
            WD_HIP(ctx, hipMalloc((void **)&ctx->d_useg, (size_t)groups * kMaxSeg * sizeof(int2)));
            WD_HIP(ctx, hipMalloc((void **)&ctx->d_wdelta, (size_t)groups * kpad * sizeof(int32_t)));
            WD_HIP(ctx, hipMalloc((void **)&ctx->d_wlev, (size_t)groups * kpad));
            WD_HIP(ctx, hipMalloc((void **)&ctx->d_wmask, (size_t)groups * (kpad / 32) * kWave * sizeof(uint32_t)));
            WD_HIP(ctx, hipMalloc((void **)&ctx->d_wfull, (size_t)groups * kpad));
            WD_HIP(ctx, hipMemsetAsync(ctx->d_ginfo, 0, (size_t)groups * sizeof(int32_t), ctx->stream));
            hipLaunchKernelGGL(k_dense_windows, dim3(groups), dim3(kWave), 0, ctx->stream, ctx->d_centre, ctx->d_lvl_off,
                               ctx->d_nbr, ctx->T, ctx->levels, ctx->win_kpad, ctx->d_ginfo, ctx->d_useg, ctx->d_uoff,
                               ctx->d_wdelta, ctx->d_wlev, ctx->d_wmask, ctx->d_tblflags + 1, ctx->dense_sym_on ? 1 : 0,
                               ctx->d_wfull);
            WD_HIP(ctx, hipMemcpyAsync(flags, ctx->d_tblflags, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
            WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->win_dwords = (int)((flags[1] + kWave - 1) / kWave * kWave);     // whole pieces of 64 dwords
        }
    }
    WD_HIP(ctx, hipGetLastError());
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));       // h_gbase may be reused by the caller
    {
        std::vector<uint8_t> guni((size_t)groups);
        std::vector<int32_t> ginfo((size_t)groups, 0);
        if (groups > 0)
            WD_HIP(ctx, hipMemcpy(guni.data(), ctx->d_guni, (size_t)groups, hipMemcpyDeviceToHost));
        if (groups > 0 && ctx->d_ginfo)
            WD_HIP(ctx, hipMemcpy(ginfo.data(), ctx->d_ginfo, (size_t)groups * sizeof(int32_t), hipMemcpyDeviceToHost));
        ctx->n_uniform_groups = ctx->n_window_groups = 0;
        for (int g = 0; g < groups; g++) {
            ctx->n_uniform_groups += guni[g] != 0;
            ctx->n_window_groups += ginfo[g] != 0;
        }
        // the target blocks (kWaves groups each) that hold a group the window kernel leaves to the gather
        // kernel: k_dense_pairs is launched over this list, not over every block of the tile (with every
        // well a centre that is ONE block - the last, partial group - and the launch over all 16 834 cost
        // 29 us per 8 tiles to find it)
        std::vector<int32_t> pb;
        for (int b = 0; b * kWaves < groups; b++) {
            bool any = false;
            for (int w = 0; w < kWaves && b * kWaves + w < groups; w++)
                any = any || ginfo[(size_t)(b * kWaves + w)] == 0;
            if (any)
                pb.push_back(b);
        }
        (void)hipFree(ctx->d_pblocks);
        ctx->d_pblocks = nullptr;
        ctx->n_pblocks = (int)pb.size();
        if (!pb.empty()) {
            WD_HIP(ctx, hipMalloc((void **)&ctx->d_pblocks, pb.size() * sizeof(int32_t)));
            WD_HIP(ctx, hipMemcpy(ctx->d_pblocks, pb.data(), pb.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    return WD_OK;
}

// Packed rows (64 bytes per well) are optional scratch for the Hamming family and mandatory for
// Levenshtein <= 2; false if they cannot be had.
bool dense_rows_reserve(wd_ctx *ctx, int n_tiles, int64_t N)
{
    const size_t rows_need = (size_t)n_tiles * (size_t)N * kRowGroups;
    if (rows_need <= ctx->rows_cap)
        return true;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess)
        return false;
    (void)hipFree(ctx->d_rows);
    ctx->d_rows = nullptr;
    ctx->rows_cap = 0;
    if (hipMalloc((void **)&ctx->d_rows, std::max<size_t>(1, rows_need) * sizeof(uint4)) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    ctx->rows_cap = rows_need;
    return true;
}

// Tiles per part of a dense scan (launch_dense): two halves of a small scan, else as many tiles as
// one compare-stage wave walks; the whole scan when the overlap is off.
int dense_part_size(const wd_ctx *ctx, int n_tiles, int tile_chunk)
{
    if (!ctx->dense_overlap || n_tiles < 2)
        return std::max(1, n_tiles);
    if (ctx->dense_part_tiles > 0)
        return std::min(ctx->dense_part_tiles, n_tiles);
    return n_tiles >= 2 * tile_chunk ? tile_chunk : (n_tiles + 1) / 2;
}

// The dense path of wd_scan_async (scan_dense.inc): signatures, pairs, verify, reduce.
//
// By default the whole scan is one chain on the caller's stream.  Option "dense_overlap" = 1 runs it as
// a pipeline of PARTS of a few tiles, each with one of two scratch sets: the chain of one part is
// serial - the pack kernel needs the marks the compare stage leaves - but its two heavy kernels are
// bound by different things (k_dense_pack streams 140 of the 150 planes: HBM; k_dense_pairs_win
// compares from LDS: instruction issue), so the COMPARE stages (sig, counts, pairs, mark) of all parts
// go one after the other on a high-priority stream of the context's own and the PACK stages (rank,
// pack, verify, reduce) on a normal one, part c's pack stage after its compare stage, part c+2's
// compare stage after part c's pack stage (it takes over the scratch set); "dense_pack_blocks" bounds
// the pack kernel's footprint so that the compare stage finds wave slots beside it.
// MEASURED (round 3, rocprofv3 traces in profiles/r03_b_dense_overlap_*): the kernels do run side by
// side, and each pays for it - beside the pack kernel k_dense_sig takes 3 x, k_dense_counts 6 - 15 x
// and k_dense_pairs_win 1.8 - 3 x as long (latency-bound kernels next to a kernel that keeps every HBM
// queue full), so 16 tiles take 2.74 ms against 2.76 ms in one chain.  Hence the default.
int launch_dense(wd_ctx *ctx, const ScanArgs &a, int n_tiles, int64_t N, bool strided, size_t n_plane_ptrs,
                 int tile_chunk, bool lev2)
{
    int rc = ensure_dense_tables(ctx);
    if (rc)
        return rc;
    // parts: two halves of a small scan, else as many tiles as one compare-stage wave walks
    const int part = dense_part_size(ctx, n_tiles, tile_chunk);
    const int n_parts = (n_tiles + part - 1) / part;
    const int n_sets = n_parts > 1 ? 2 : 1;
    if (n_parts > 1 && !ctx->dense_hi) {
        int least = 0, greatest = 0;
        WD_HIP(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        WD_HIP(ctx, hipStreamCreateWithPriority(&ctx->dense_hi, hipStreamNonBlocking, greatest));
        WD_HIP(ctx, hipStreamCreateWithPriority(&ctx->dense_lo, hipStreamNonBlocking, least));
        for (hipEvent_t *e : {&ctx->dense_ev_start, &ctx->dense_ev_done, &ctx->dense_ev_cmp[0], &ctx->dense_ev_cmp[1],
                              &ctx->dense_ev_pack[0], &ctx->dense_ev_pack[1]})
            WD_HIP(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    DenseArgs d;
    d.stride = a.stride;
    d.centre = a.centre;
    d.lvl_off = a.lvl_off;
    d.nbr_t = ctx->d_nbr_t;
    d.idx16 = ctx->nbr_t16 ? 1 : 0;
    d.rel_t = ctx->d_rel_t;
    d.udelta = ctx->d_udelta;
    d.guni = ctx->d_guni;
    d.ginfo = ctx->dense_windows ? ctx->d_ginfo : nullptr;
    d.uoff = ctx->d_uoff;
    d.useg = ctx->d_useg;
    d.wdelta = ctx->d_wdelta;
    d.wlev = ctx->d_wlev;
    d.wmask = ctx->d_wmask;
    d.wfull = ctx->d_wfull;
    d.sym = ctx->dense_sym_on ? 1 : 0;
    d.centre0 = ctx->centre0;
    d.kpad = ctx->win_kpad;
    d.win_dwords = ctx->win_dwords;
    d.gbase = ctx->d_gbase;
    d.rare = a.rare;
    d.N = N;
    d.T = a.T;
    d.levels = a.levels;
    d.L = a.L;
    d.k = a.k;
    d.sig_cycles = std::min(kSigCycles, a.L);
    d.strided = strided ? 1 : 0;
    d.check_empty = a.check_empty;
    d.log_hits = a.log_hits;
    d.sig_stride = (N + 127) & ~(long long)127;
    d.partial_stride = ((1 + 5 * a.levels) + 15) & ~15;              // whole 128-byte lines per slot
    d.mask_stride = (((long long)a.T + 3) / 4 + 31) & ~31ll;
    // survivors per group of 64 targets: on diverse reads almost nobody passes 10 cycles (36 x
    // P(<= 2 mismatches in 10) = 0.015 per target; ~2 per group for Levenshtein <= 2), so the
    // region holds mostly duplicate pairs; what does not fit is finished inside k_dense_pairs (or,
    // for Levenshtein, by k_dense_verify): a speed knob, not a limit
    const long long n_groups = (a.T + kWave - 1) / kWave;
    long long q_per = ctx->dense_queue_cap > 0 ? ctx->dense_queue_cap : (lev2 ? 32 : 16);
    // LDS of a k_dense_pairs wave: its signature windows, then 8 bytes per queue entry
    q_per = std::max<long long>(1, std::min<long long>(q_per, kWave));      // one queue entry per lane at most
    d.q_per = (int)q_per;
    // (the one-ended compare keeps the windows of three tiles in flight while LDS lets five workgroups share a CU)
    d.win_bufs = kWinBufs;
    int npc = 0;                                        // pieces of 64 signatures per window, if the kernel is built for that many
    if (d.sym) {
        for (d.win_bufs = kWinBufsSym; d.win_bufs > kWinBufs; d.win_bufs--)
            if ((size_t)kWaves * ((size_t)d.win_bufs * d.win_dwords * 4 + 4 * ((size_t)tile_chunk * (2 * q_per + 1) + 1)) <= 32 * 1024)
                break;
        if (d.win_bufs == kWinBufsSym && d.ginfo)
            for (int cand : {4, 5, 6, 8})
                if (!npc && d.win_dwords <= cand * kWave &&
                    (size_t)kWaves * ((size_t)d.win_bufs * cand * kWave * 4 + 4 * ((size_t)tile_chunk * (2 * q_per + 1) + 1)) <= 32 * 1024)
                    npc = cand;
        if (npc)
            d.win_dwords = npc * kWave;
    }
    const long long win_bytes = (long long)d.win_bufs * d.win_dwords * sizeof(uint32_t);
    d.mw_stride = (((N + 31) / 32) + kMarkBlock - 1) / kMarkBlock * kMarkBlock;
    // scratch of ONE part (the largest); set s of a buffer starts s parts in
    const size_t sig_words = (size_t)d.sig_stride * part * ((lev2 || a.k > 0) ? 2 : 1);
    const size_t mark_words = (size_t)part * d.mw_stride;
    const size_t part_need = (size_t)part * kDenseSlots * d.partial_stride;
    const size_t mask_need = (size_t)part * d.mask_stride;
    const size_t regions = (size_t)part * (size_t)n_groups;
    const size_t cand_words = (kDenseSlots + 1 + 31) & ~(size_t)31;
    if ((rc = dense_reserve(ctx, ctx->d_sig, ctx->sig_cap, sig_words * n_sets, "signature planes")) ||
        (rc = dense_reserve(ctx, ctx->d_partial, ctx->partial_cap, part_need * n_sets, "counter slots")) ||
        (rc = dense_reserve(ctx, ctx->d_mask, ctx->mask_cap, mask_need * n_sets, "hit masks")) ||
        (rc = dense_reserve(ctx, ctx->d_queue, ctx->queue_cap, regions * (size_t)q_per * n_sets, "survivor queue")) ||
        (rc = dense_reserve(ctx, ctx->d_qcnt, ctx->qcnt_cap, regions * n_sets, "survivor counts")) ||
        (rc = dense_reserve(ctx, ctx->d_mark, ctx->mark_cap, 3 * mark_words * n_sets, "marked wells")) ||
        (rc = dense_reserve(ctx, ctx->d_cand, ctx->cand_cap, cand_words * 2, "survivor flags")))
        return rc;
    // packed rows are optional scratch (64 bytes per well): without them every survivor is
    // checked against the planes
    d.pack_mode = ctx->dense_pack;
    d.lev2 = lev2 ? 1 : 0;
    d.nbr = a.nbr;
    const bool want_rows = (d.pack_mode != 0 || lev2) && a.L > d.sig_cycles && a.L <= 40 * kRowGroups &&
                           dense_rows_reserve(ctx, part * n_sets, N);
    if (lev2 && a.L > d.sig_cycles && !want_rows)            // (the caller reserved them)
        return fail(ctx, WD_ERR_NOMEM, "packed rows");
    // Checking one survivor against the planes touches 2 (L - 10) cache lines, one per plane and
    // well; packing touches at most one line per plane and MARKED well (wells of a line share it,
    // lines without a marked well are skipped) and leaves a 64-byte row per marked well: never
    // more lines than the byte-by-byte check, so rows are used whenever there is a survivor
    // (dense_packed, scan_dense.inc).

    // dword loads in k_dense_sig need every plane 4-byte aligned
    bool aligned4 = (a.stride & 3) == 0;
    for (size_t i = 0; i < n_plane_ptrs && aligned4; i++)
        aligned4 = ((uintptr_t)ctx->h_tbl[i] & 3u) == 0;
    const int pmode = lev2 ? 2 : (a.k == 0 ? 0 : 1);
    const unsigned mark_blocks = (unsigned)((n_groups + kWave * kWaves - 1) / (kWave * kWaves));
    const long long dense_bpt = (a.T + kBlock - 1) / kBlock;
    // st: the compare stage's stream, sp: the pack stage's
    hipStream_t st = n_parts > 1 ? ctx->dense_hi : ctx->stream, sp = n_parts > 1 ? ctx->dense_lo : ctx->stream;
    if (n_parts > 1) {
        WD_HIP(ctx, hipEventRecord(ctx->dense_ev_start, ctx->stream));
        WD_HIP(ctx, hipStreamWaitEvent(st, ctx->dense_ev_start, 0));
        WD_HIP(ctx, hipStreamWaitEvent(sp, ctx->dense_ev_start, 0));
    }
    for (int c = 0; c < n_parts; c++) {
        const int t0 = c * part, nt = std::min(part, n_tiles - t0), set = c & 1;
        if (c >= 2)                                 // the scratch set is free when part c - 2 has been verified
            WD_HIP(ctx, hipStreamWaitEvent(st, ctx->dense_ev_pack[set], 0));
        const int tc = std::max(1, std::min(tile_chunk, nt));
        d.planes = a.planes + (strided ? (size_t)t0 : (size_t)t0 * a.L);
        d.filter = a.filter + t0;
        d.out_per_target = a.out_per_target ? a.out_per_target + (size_t)t0 * a.T * a.levels : nullptr;
        d.tile0 = t0;
        d.n_tiles = nt;
        d.tile_chunk = tc;
        d.sig = ctx->d_sig + sig_words * set;
        // distances > 0: the compare stage reads 16-cycle screen words (k_dense_sig)
        d.sig2 = (lev2 || a.k > 0) ? d.sig + (size_t)d.sig_stride * nt : nullptr;
        d.partial = ctx->d_partial + part_need * set;
        d.mask = ctx->d_mask + mask_need * set;
        d.queue = ctx->d_queue + regions * (size_t)q_per * set;
        d.q_cnt = ctx->d_qcnt + regions * set;
        d.mark = ctx->d_mark + 3 * mark_words * set;
        d.wprefix = d.mark + (size_t)nt * d.mw_stride;
        d.bprefix = d.mark + 2 * (size_t)nt * d.mw_stride;
        d.cand = ctx->d_cand + cand_words * set;
        d.rows = want_rows ? ctx->d_rows + (size_t)part * (size_t)N * kRowGroups * set : nullptr;
        // (k_dense_sig clears the part's scratch: dense_clear_scratch)
        const dim3 grid((unsigned)((long long)kXcds * ((dense_bpt + kXcds - 1) / kXcds) * ((nt + tc - 1) / tc)));
        const dim3 grid4((unsigned)((N + 4ll * kBlock - 1) / (4ll * kBlock)), (unsigned)nt);
        const dim3 grid1((unsigned)((N + kBlock - 1) / kBlock), (unsigned)nt);
        // the gather kernel: over the listed target blocks only, when the window kernel takes the rest
        d.pblocks = d.ginfo ? ctx->d_pblocks : nullptr;
        d.n_pblocks = d.ginfo ? ctx->n_pblocks : 0;
        const dim3 pgrid = d.n_pblocks > 0 ? dim3((unsigned)((long long)d.n_pblocks * ((nt + tc - 1) / tc))) : grid;
        if (aligned4 && strided)
            hipLaunchKernelGGL((k_dense_sig<true, true>), grid4, dim3(kBlock), 0, st, d);
        else if (aligned4)
            hipLaunchKernelGGL((k_dense_sig<true, false>), grid4, dim3(kBlock), 0, st, d);
        else if (strided)
            hipLaunchKernelGGL((k_dense_sig<false, true>), grid1, dim3(kBlock), 0, st, d);
        else
            hipLaunchKernelGGL((k_dense_sig<false, false>), grid1, dim3(kBlock), 0, st, d);
        // LDS of a compare-stage wave: its windows, one queue per tile of the chunk, their counts
        const size_t q_lds = (size_t)kWaves * ((size_t)win_bytes + 4 * ((size_t)tc * (2 * d.q_per + 1) + (tc & 1)));
        hipLaunchKernelGGL(k_dense_counts, dim3((unsigned)((a.T + 4 * kBlock - 1) / (4 * kBlock)), (unsigned)((nt + tc - 1) / tc)),
                           dim3(kBlock), 0, st, d);
#define WD_LAUNCH_PAIRS(MODE, SYM)                                                                              \
    do {                                                                                                        \
        if (d.ginfo)                                                                                            \
            hipLaunchKernelGGL((k_dense_pairs_win<MODE, SYM>), grid, dim3(kBlock), q_lds, st, d);               \
        if (ctx->n_window_groups < n_groups || !d.ginfo) {                                                      \
            if (ctx->nbr_t16)                                                                                   \
                hipLaunchKernelGGL((k_dense_pairs<MODE, true, SYM>), pgrid, dim3(kBlock), q_lds, st, d);       \
            else                                                                                                \
                hipLaunchKernelGGL((k_dense_pairs<MODE, false, SYM>), pgrid, dim3(kBlock), q_lds, st, d);     \
        }                                                                                                       \
    } while (0)
#define WD_LAUNCH_PAIRS_N(MODE, NPC)                                                                            \
    do {                                                                                                        \
        hipLaunchKernelGGL((k_dense_pairs_win<MODE, true, NPC>), grid, dim3(kBlock), q_lds, st, d);             \
        if (ctx->n_window_groups < n_groups) {                                                                  \
            if (ctx->nbr_t16)                                                                                   \
                hipLaunchKernelGGL((k_dense_pairs<MODE, true, true>), pgrid, dim3(kBlock), q_lds, st, d);      \
            else                                                                                                \
                hipLaunchKernelGGL((k_dense_pairs<MODE, false, true>), pgrid, dim3(kBlock), q_lds, st, d);    \
        }                                                                                                       \
    } while (0)
#define WD_LAUNCH_PAIRS_SYM(MODE)                                                                               \
    do {                                                                                                        \
        switch (npc) {                                                                                          \
        case 4: WD_LAUNCH_PAIRS_N(MODE, 4); break;                                                              \
        case 5: WD_LAUNCH_PAIRS_N(MODE, 5); break;                                                              \
        case 6: WD_LAUNCH_PAIRS_N(MODE, 6); break;                                                              \
        case 8: WD_LAUNCH_PAIRS_N(MODE, 8); break;                                                              \
        default: WD_LAUNCH_PAIRS(MODE, true); break;                                                            \
        }                                                                                                       \
    } while (0)
        if (pmode == 0 && d.sym)
            WD_LAUNCH_PAIRS_SYM(0);
        else if (pmode == 0)
            WD_LAUNCH_PAIRS(0, false);
        else if (pmode == 1 && d.sym)
            WD_LAUNCH_PAIRS_SYM(1);
        else if (pmode == 1)
            WD_LAUNCH_PAIRS(1, false);
        else if (d.sym)
            WD_LAUNCH_PAIRS_SYM(2);
        else
            WD_LAUNCH_PAIRS(2, false);
#undef WD_LAUNCH_PAIRS_SYM
#undef WD_LAUNCH_PAIRS_N
#undef WD_LAUNCH_PAIRS
        if (lev2)
            hipLaunchKernelGGL(k_dense_mark, dim3(mark_blocks, (unsigned)nt), dim3(kBlock), 0, st, d);
        if (n_parts > 1) {                          // compared: the pack stage may start, the next part's compare stage does
            WD_HIP(ctx, hipEventRecord(ctx->dense_ev_cmp[set], st));
            WD_HIP(ctx, hipStreamWaitEvent(sp, ctx->dense_ev_cmp[set], 0));
        }
        if (d.rows) {
            hipLaunchKernelGGL(k_dense_rank_words, dim3((unsigned)(d.mw_stride / kMarkBlock), (unsigned)nt), dim3(kMarkBlock), 0, sp, d);
            hipLaunchKernelGGL(k_dense_rank_blocks, dim3((unsigned)nt), dim3(1024), 0, sp, d);
            // with a compare stage beside it the pack kernel gets a bounded footprint: dense_pack_blocks
            // workgroups in all (they walk their tiles with a stride), not one per kBlock * VEC wells
            dim3 pg4 = grid4, pg1 = grid1;
            if (n_parts > 1 && ctx->dense_pack_blocks > 0) {
                const unsigned per_tile = (unsigned)std::max(1, ctx->dense_pack_blocks / nt);
                pg4.x = std::min(pg4.x, per_tile);
                pg1.x = std::min(pg1.x, per_tile);
            }
            if (aligned4 && strided && ctx->dense_nt)
                hipLaunchKernelGGL((k_dense_pack<4, true, true>), pg4, dim3(kBlock), 0, sp, d);
            else if (aligned4 && strided)
                hipLaunchKernelGGL((k_dense_pack<4, true>), pg4, dim3(kBlock), 0, sp, d);
            else if (aligned4)
                hipLaunchKernelGGL((k_dense_pack<4, false>), pg4, dim3(kBlock), 0, sp, d);
            else if (strided)
                hipLaunchKernelGGL((k_dense_pack<1, true>), pg1, dim3(kBlock), 0, sp, d);
            else
                hipLaunchKernelGGL((k_dense_pack<1, false>), pg1, dim3(kBlock), 0, sp, d);
        }
        const dim3 vgrid(kXcds * ((mark_blocks + kXcds - 1) / kXcds), (unsigned)nt);
        if (strided)
            hipLaunchKernelGGL((k_dense_verify<true>), vgrid, dim3(kBlock), 0, sp, d);
        else
            hipLaunchKernelGGL((k_dense_verify<false>), vgrid, dim3(kBlock), 0, sp, d);
        hipLaunchKernelGGL(k_dense_reduce, dim3(nt), dim3(kWave), 0, sp, d.partial, d.partial_stride, 1 + 5 * a.levels,
                           a.out_tile + (size_t)t0 * (1 + 5 * a.levels));
        if (n_parts > 1)
            WD_HIP(ctx, hipEventRecord(ctx->dense_ev_pack[set], sp));
    }
    if (n_parts > 1) {                              // the caller's stream has it all behind it
        WD_HIP(ctx, hipEventRecord(ctx->dense_ev_done, sp));     // (every compare stage lies before some pack stage)
        WD_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->dense_ev_done, 0));
    }
    return WD_OK;
}

}  // namespace wd
