// welldup_scan.hip - targets, the scans of sampled targets and the dispatch of wd_scan_async.
// Hot path of count_well_duplicates.py:228-265 (target -> level -> neighbour compare) fused with the gather
// of Tile.get_seqs (bcl_direct_reader.py:158-220, :352-361) and the integer part of output_writer
// (count_well_duplicates.py:63-106).  Interface: include/welldup.h; layout of the sources: welldup_core.hip.
//
// Design in one paragraph (DESIGN.md has the full story): HBM-bound byte/integer work, no MFMA.
// The gather of Tile.get_seqs is fused into the compare and is lazy in the cycle direction: a
// neighbour that already has more than k mismatches (or whose banded edit-distance row is all
// > k) can never become a duplicate, so its remaining bytes are never fetched; HBM hands out
// 128-byte lines, so the kernels are organised around touching few lines and hiding dependent
// round trips (LDS-staged metadata, software pipelining, stream-compacted survivor queues in
// LDS, wave ballots for the per-level counts, LDS histograms for the tallies).  No CUDA shims,
// no dual paths: gfx950 HIP only.
#include "wd_ctx.h"

#ifndef WD_UNIT_ID
#define WD_UNIT_ID "unknown"
#endif
namespace wd { const char *unit_id_scan() { return WD_UNIT_ID; } }      // hash of this unit's sources (wd_build_id)

using namespace wd;

namespace {

#include "device_common.inc"
#include "scan_sequential.inc"
#include "scan_lev_generic.inc"
#include "gen_rings.inc"

// Batch shapes (cycles read unconditionally, then per conditional batch).  The Hamming family
// is instantiated for a few shapes so they can be tuned; the banded edit-distance family
// needs ~2x the cycles before a random neighbour dies, so it uses one deeper shape.
constexpr int kHamShapes[][2] = {{2, 4}, {3, 4}, {4, 4}, {4, 8}, {8, 8}};
constexpr int kLevB2 = 8;

template <bool STRIDED>
void launch_ham(wd_ctx *ctx, const ScanArgs &a, dim3 grid)
{
    const int b1 = ctx->batch_first, b2 = ctx->batch_next;
#define WD_CASE(B1, B2)                                                                   \
    if (b1 == B1 && b2 == B2) {                                                           \
        snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "k_scan<HamState, %s, %d, %d>", STRIDED ? "true" : "false", B1, B2); \
        hipLaunchKernelGGL((k_scan<HamState, STRIDED, B1, B2>), grid, dim3(kBlock), 0,    \
                           ctx->stream, a);                                               \
        return;                                                                           \
    }
    WD_CASE(2, 4)
    WD_CASE(3, 4)
    WD_CASE(4, 4)
    WD_CASE(4, 8)
    WD_CASE(8, 8)
#undef WD_CASE
}

template <int H>
void launch_lev(wd_ctx *ctx, const ScanArgs &a, dim3 grid, bool strided)
{
    snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "k_scan<LevState<%d>, %s, %d, %d>", H, strided ? "true" : "false",
             lev_first(H), kLevB2);
    if (strided)
        hipLaunchKernelGGL((k_scan<LevState<H>, true, lev_first(H), kLevB2>), grid, dim3(kBlock), 0,
                           ctx->stream, a);
    else
        hipLaunchKernelGGL((k_scan<LevState<H>, false, lev_first(H), kLevB2>), grid, dim3(kBlock), 0,
                           ctx->stream, a);
}


// the queue kernels read the targets sorted by centre (install_sorted_view)
void queue_view(const wd_ctx *ctx, ScanArgs &a)
{
    if (ctx->sort_targets && ctx->d_perm) {
        a.centre = ctx->d_centre_q;
        a.lvl_off = ctx->d_lvl_off_q;
        a.perm = ctx->d_perm;
    }
}


// The queue kernels walk the targets in the order of their centre wells, not in file order: a sampled
// targets file is a random permutation of the tile (prepare_cluster_indexes.py:26-30, random.sample), and
// neighbourhoods of different targets overlap - on the bench workload a third of the (plane, line) pairs a
// scan touches are touched by two targets or more.  Sorted, those targets sit in the same workgroup or
// the next one (which the block mapping of k_scan_q puts on the same XCD), and the second touch is an
// L2 hit instead of an HBM line.  Only `centre` and the rows of `lvl_off` are permuted (the rows hold
// absolute offsets into nbr, which stays as it is); perm[sorted position] = target's index in the file,
// for the per-target output and the hit log.  Tallies are sums: order-free.
//
// The order is by column strip, then by well: strips of `sort_strip` wells (four cache lines) of the
// tile's rows, whose length is read off the targets themselves (row_length_of).  In plain well order
// the targets that share lines with a target - those within a few rows AND a line's width of columns -
// are spread over the 20 nearest targets of the order, most of them in other columns; within a strip
// they are the next one or two, in flight at the same time, and the shared line is still in the L2.
int install_sorted_view(wd_ctx *ctx, const int32_t *centre, const int32_t *lvl_off, int T, int levels, long long row_len)
{
    (void)hipFree(ctx->d_centre_q);
    (void)hipFree(ctx->d_lvl_off_q);
    (void)hipFree(ctx->d_perm);
    ctx->d_centre_q = ctx->d_lvl_off_q = ctx->d_perm = nullptr;
    if (T < 2 || T >= 65536 * 64)              // (every well a centre: the dense path's business, and sorted as it is)
        return WD_OK;
    const long long strip = ctx->sort_strip > 0 && row_len >= 2ll * ctx->sort_strip ? ctx->sort_strip : 0;
    auto key = [&](int32_t t) -> long long {
        const long long c = centre[t];
        return strip ? ((c % row_len) / strip) * (1ll << 40) + c : c;
    };
    bool sorted = true;
    for (int t = 1; t < T && sorted; t++)
        sorted = key(t - 1) <= key(t);
    if (sorted)
        return WD_OK;
    const size_t row = (size_t)levels + 1;
    std::vector<int32_t> perm((size_t)T), c2((size_t)T), o2((size_t)T * row);
    for (int t = 0; t < T; t++)
        perm[(size_t)t] = t;
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return key(a) < key(b); });
    for (int t = 0; t < T; t++) {
        c2[(size_t)t] = centre[perm[(size_t)t]];
        memcpy(&o2[(size_t)t * row], lvl_off + (size_t)perm[(size_t)t] * row, row * sizeof(int32_t));
    }
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_centre_q, (size_t)T * sizeof(int32_t)));
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_lvl_off_q, (size_t)T * row * sizeof(int32_t)));
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_perm, (size_t)T * sizeof(int32_t)));
    WD_HIP(ctx, hipMemcpy(ctx->d_centre_q, c2.data(), (size_t)T * sizeof(int32_t), hipMemcpyHostToDevice));
    WD_HIP(ctx, hipMemcpy(ctx->d_lvl_off_q, o2.data(), (size_t)T * row * sizeof(int32_t), hipMemcpyHostToDevice));
    WD_HIP(ctx, hipMemcpy(ctx->d_perm, perm.data(), (size_t)T * sizeof(int32_t), hipMemcpyHostToDevice));
    return WD_OK;
}

// Wells per row of the tile, as the targets show it: a target's nearest neighbours that are not beside
// it in its own row lie one row up or down, about a row's length away in index (prepare_cluster_indexes.py
// bins by distance on the honeycomb).  The median over a few hundred targets; 0 if they do not say.
long long row_length_of(const int32_t *centre, const int32_t *lvl_off, const int32_t *nbr, int T, int levels)
{
    if (!nbr || levels < 1)
        return 0;
    std::vector<long long> est;
    const size_t row = (size_t)levels + 1;
    for (int t = 0; t < T && est.size() < 512; t++) {
        long long best = 0;
        for (int32_t p = lvl_off[(size_t)t * row]; p < lvl_off[(size_t)t * row + 1]; p++) {
            const long long d = llabs((long long)nbr[p] - centre[t]);
            if (d > 8 && (best == 0 || d < best))
                best = d;
        }
        if (best)
            est.push_back(best);
    }
    if (est.size() < 8)
        return 0;
    std::nth_element(est.begin(), est.begin() + (long)est.size() / 2, est.end());
    return est[est.size() / 2];
}


bool valid_batches(int b1, int b2)
{
    for (auto &p : kHamShapes)
        if (p[0] == b1 && p[1] == b2)
            return true;
    return false;
}

}  // namespace

extern "C" {

int wd_set_targets(wd_ctx *ctx, int T, int levels, const int32_t *centre, const int32_t *lvl_off,
                   const int32_t *nbr)
try {
    if (!ctx || T < 0 || levels < 0 || levels > kMaxLevels)
        return fail(ctx, WD_ERR_ARG, "bad T or levels");
    if (T > 0 && (!centre || !lvl_off))
        return fail(ctx, WD_ERR_ARG, "null targets array");
    if (bind_device(ctx))
        return WD_ERR_HIP;
    int64_t P = 0;
    bool empty = false;
    int64_t lo = INT64_MAX, hi = INT64_MIN;
    const size_t row = (size_t)levels + 1;
    for (int t = 0; t < T; t++) {
        const int32_t *o = lvl_off + (size_t)t * row;
        for (int l = 0; l < levels; l++) {
            if (o[l + 1] < o[l] || o[l] < 0)
                return fail(ctx, WD_ERR_ARG, "lvl_off must be non-decreasing and >= 0");
            if (o[l + 1] == o[l])
                empty = true;
        }
        if (o[0] < 0)
            return fail(ctx, WD_ERR_ARG, "lvl_off must be >= 0");
        P = std::max<int64_t>(P, o[levels]);
        lo = std::min<int64_t>(lo, centre[t]);
        hi = std::max<int64_t>(hi, centre[t]);
    }
    if (P > 0 && !nbr)
        return fail(ctx, WD_ERR_ARG, "null nbr array");
    // only slots some target refers to are range-checked (get_all_indices, target.py:93-95)
    for (int t = 0; t < T; t++) {
        const int32_t *o = lvl_off + (size_t)t * row;
        for (int64_t p = o[0]; p < o[levels]; p++) {
            lo = std::min<int64_t>(lo, nbr[p]);
            hi = std::max<int64_t>(hi, nbr[p]);
        }
    }
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->d_centre);
    (void)hipFree(ctx->d_lvl_off);
    (void)hipFree(ctx->d_nbr);
    ctx->d_centre = ctx->d_lvl_off = ctx->d_nbr = nullptr;
    ctx->has_targets = false;
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_centre, std::max<size_t>(1, T) * sizeof(int32_t)));
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_lvl_off, std::max<size_t>(1, T * row) * sizeof(int32_t)));
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_nbr, std::max<int64_t>(1, P) * sizeof(int32_t)));
    if (T > 0) {
        WD_HIP(ctx, hipMemcpy(ctx->d_centre, centre, (size_t)T * sizeof(int32_t), hipMemcpyHostToDevice));
        WD_HIP(ctx, hipMemcpy(ctx->d_lvl_off, lvl_off, (size_t)T * row * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if (P > 0)
        WD_HIP(ctx, hipMemcpy(ctx->d_nbr, nbr, (size_t)P * sizeof(int32_t), hipMemcpyHostToDevice));
    ctx->T = T;
    ctx->levels = levels;
    ctx->P = P;
    ctx->idx_min = T ? lo : 0;
    ctx->idx_max = T ? hi : -1;
    ctx->has_empty_level = empty;
    ctx->k_max = 0;
    for (int t = 0; t < T; t++)
        ctx->k_max = std::max<int64_t>(ctx->k_max, (int64_t)lvl_off[(size_t)t * row + levels] - lvl_off[(size_t)t * row]);
    set_group_bases(ctx, lvl_off, T, levels);
    if (int rc = install_sorted_view(ctx, centre, lvl_off, T, levels, row_length_of(centre, lvl_off, nbr, T, levels)))
        return rc;
    ctx->has_targets = true;
    return WD_OK;
} WD_CATCH

int wd_scan_async(wd_ctx *ctx, int n_tiles, int L, int mode, int k, const uint8_t *const *planes,
                  const uint8_t *const *filter, int64_t N, int64_t *out_tile_dev,
                  uint32_t *out_per_target_dev)
try {
    if (!ctx)
        return WD_ERR_ARG;
    if (!ctx->has_targets)
        return fail(ctx, WD_ERR_STATE, "wd_set_targets has not been called");
    if (n_tiles < 0 || L < 0 || N < 0 || !out_tile_dev)
        return fail(ctx, WD_ERR_ARG, "bad n_tiles, L, N or out_tile");
    if (mode != WD_MODE_EQ && mode != WD_MODE_HAMMING && mode != WD_MODE_LEVENSHTEIN)
        return fail(ctx, WD_ERR_ARG, "bad mode");
    if (n_tiles > 0 && (!filter || (L > 0 && !planes)))
        return fail(ctx, WD_ERR_ARG, "null plane/filter table");
    if (!valid_batches(ctx->batch_first, ctx->batch_next))
        return fail(ctx, WD_ERR_ARG, "unsupported batch_first/batch_next pair");
    if (bind_device(ctx))
        return WD_ERR_HIP;
    // Tile.get_seqs: every requested index must lie inside the tile (bcl_direct_reader.py:186-192)
    if (ctx->T > 0 && n_tiles > 0 && (ctx->idx_min < 0 || ctx->idx_max >= N))
        return fail(ctx, WD_ERR_INDEX, "a target index lies outside [0, N)");

    const int levels = ctx->levels;
    const size_t ncnt = 1 + 5 * (size_t)levels;
    WD_HIP(ctx, hipMemsetAsync(out_tile_dev, 0, (size_t)n_tiles * ncnt * sizeof(int64_t), ctx->stream));
    if (n_tiles == 0 || ctx->T == 0)
        return WD_OK;

    // normalise the compare: equality and Levenshtein <= 1 are Hamming problems
    int kk = k;
    bool lev = false;
    if (mode == WD_MODE_EQ) {
        kk = 0;
    } else if (mode == WD_MODE_LEVENSHTEIN) {
        if (k >= L) {
            // every equal-length pair is within L substitutions: a Hamming problem - unless the
            // hit log wants the true (possibly smaller) edit distance of every pair (:260-262)
            kk = L;
            lev = ctx->hit_cap > 0 && L >= 2 && lev_generic_lds_bytes(L, L / 2) <= 64 * 1024;
        } else if (k >= 2) {
            lev = true;
        }                            // k <= 1: equal lengths, so one edit is one substitution
    }
    if (kk > L)
        kk = L;
    if (kk < -1)
        kk = -1;
    const bool lev_generic = lev && kk / 2 > 8;
    if (lev_generic && lev_generic_lds_bytes(L, kk / 2) > 64 * 1024)
        return fail(ctx, WD_ERR_UNSUPPORTED,
                    "Levenshtein threshold k >= 18 with this read length needs more than 64 KB of LDS");

    // pointer tables: uniform plane stride -> per-tile base only
    bool strided = L > 0;
    int64_t stride = 0;
    const int ws = ctx->well_stride;
    if (ws == 4) {
        // cycles interleaved by four: cycle c of a tile lives at base + (c / 4) * group stride +
        // c % 4, wells 4 bytes apart; the pointers must say exactly that
        if (L > 4)
            stride = (int64_t)(planes[4] - planes[0]);
        for (int i = 0; i < n_tiles; i++) {
            const uint8_t *b = planes[(size_t)i * L];
            if (L > 0 && ((uintptr_t)b & 3u))
                return fail(ctx, WD_ERR_ARG, "interleaved planes must start 4-byte aligned");
            for (int c = 0; c < L; c++)
                if (planes[(size_t)i * L + c] != b + (int64_t)(c >> 2) * stride + (c & 3))
                    return fail(ctx, WD_ERR_ARG, "plane pointers do not describe the interleaved layout");
        }
        if (N > (1ll << 30))
            return fail(ctx, WD_ERR_UNSUPPORTED, "interleaved layout: more than 2^30 wells");
    } else {
        if (L > 1)
            stride = (int64_t)(planes[1] - planes[0]);
        for (int i = 0; i < n_tiles && strided; i++)
            for (int c = 1; c < L; c++)
                if ((int64_t)(planes[(size_t)i * L + c] - planes[(size_t)i * L]) != stride * c) {
                    strided = false;
                    break;
                }
    }
    const size_t n_plane_ptrs = strided ? (size_t)n_tiles : (size_t)n_tiles * L;
    std::vector<const uint8_t *> tbl(n_plane_ptrs + n_tiles);
    for (int i = 0; i < n_tiles; i++) {
        if (strided)
            tbl[i] = planes[(size_t)i * L];
        else
            for (int c = 0; c < L; c++)
                tbl[(size_t)i * L + c] = planes[(size_t)i * L + c];
        tbl[n_plane_ptrs + i] = filter[i];
    }
    if (tbl != ctx->h_tbl) {
        int rc = grow(ctx, ctx->d_tbl, ctx->d_tbl_cap, tbl.size());
        if (rc)
            return rc;
        // the previous table may still be read by queued kernels
        WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        WD_HIP(ctx, hipMemcpy(ctx->d_tbl, tbl.data(), tbl.size() * sizeof(void *), hipMemcpyHostToDevice));
        ctx->h_tbl.swap(tbl);
    }

    ScanArgs a;
    a.planes = ctx->d_tbl;
    a.filter = ctx->d_tbl + n_plane_ptrs;
    a.stride = stride;
    a.centre = ctx->d_centre;
    a.lvl_off = ctx->d_lvl_off;
    a.nbr = ctx->d_nbr;
    a.out_tile = (unsigned long long *)out_tile_dev;
    a.out_per_target = out_per_target_dev;
    a.perm = nullptr;
    {
        ScanRare r{ctx->d_status, ctx->hit_cap > 0 ? ctx->d_hits : nullptr, ctx->d_hit_count,
                   (long long)ctx->hit_cap};
        if (memcmp(&r, &ctx->h_rare, sizeof(r)) != 0) {
            WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            WD_HIP(ctx, hipMemcpy(ctx->d_rare, &r, sizeof(r), hipMemcpyHostToDevice));
            ctx->h_rare = r;
        }
    }
    a.rare = ctx->d_rare;
    a.log_hits = ctx->hit_cap > 0 ? 1 : 0;
    a.T = ctx->T;
    a.levels = levels;
    a.L = L;
    a.k = kk;
    a.tpb = ctx->tpb;
    a.early = ctx->early_exit;
    a.check_empty = ctx->has_empty_level ? 1 : 0;
    if (ctx->hit_cap > 0)
        WD_HIP(ctx, hipMemsetAsync(ctx->d_hit_count, 0, sizeof(unsigned long long), ctx->stream));

    // lane-per-target kernel: many small targets (every well a centre), Hamming family
    // lev2: the reference's default, Levenshtein <= 2, has a closed form for equal-length reads
    // (scan_dense.inc: lev2_window) and needs the packed rows, i.e. L <= 160
    const bool lev2 = lev && kk == 2 && L <= 40 * kRowGroups;
    const bool dense_ok = (!lev || lev2) && ctx->early_exit && L >= 1 && ctx->k_max <= kDenseMaxK &&
                          levels <= 8 && kk <= 2 && kk >= 0 &&    // levels: 8-bit hit masks
                          n_tiles <= 65535;                        // tiles ride in gridDim.y
    bool use_dense = dense_ok && (ctx->dense_kernel == 1 || (ctx->dense_kernel < 0 && ctx->T >= 65536));
    const int tile_chunk = std::max(1, std::min({ctx->dense_tile_chunk, n_tiles, 16}));    // (LDS: one survivor queue per tile)
    if (use_dense && lev2 && L > kSigCycles) {
        // rows for the parts in flight (two scratch sets), not for the whole scan
        const int part = dense_part_size(ctx, n_tiles, tile_chunk);
        if (!dense_rows_reserve(ctx, part * (part < n_tiles ? 2 : 1), N))
            use_dense = false;                               // no room for the rows: queue kernel
    }
    if (ws == 4 && ctx->dense_kernel < 0)
        use_dense = false;                                   // the dense path reads planes
    const int chunks = (ctx->T + ctx->tpb - 1) / ctx->tpb;
    // dense grid: 8 XCDs x (target blocks per XCD) x tile_chunk x (chunks of tiles), see k_dense_pairs
    const long long dense_bpt = (ctx->T + kBlock - 1) / kBlock;
    const long long dense_blocks = (long long)kXcds * ((dense_bpt + kXcds - 1) / kXcds) *
                                   ((n_tiles + tile_chunk - 1) / tile_chunk);
    const long long nblocks = lev_generic ? (long long)ctx->T * n_tiles
                              : use_dense ? dense_blocks
                                          : (long long)chunks * n_tiles;
    if (nblocks > 0x7FFFFFFFll)
        return fail(ctx, WD_ERR_UNSUPPORTED, "grid too large; raise targets_per_block");
    dim3 grid((unsigned)nblocks);

    const bool use_queue = !lev && ctx->queue_kernel && ctx->early_exit && kk <= 254 &&
                           ctx->k_max <= (int64_t)kMaxPasses * kPass;
    // Levenshtein <= 2 / <= 3 in the queue kernel (band half-width 1) reads the interleaved layout too
    const bool lev_queue_il = lev && !lev_generic && kk / 2 == 1 && ctx->queue_kernel && ctx->early_exit &&
                              ctx->k_max <= (int64_t)kMaxPasses * kPass;

    // the line walk (scan_lines.inc): the pairs in the order of their neighbour wells, where the queue kernel
    // would run (planes, early exit, equality / Hamming, or Levenshtein <= 2 by the closed form)
    bool use_lines = false;
    if (line_walk_wanted(ctx) && !use_dense && (ws == 1 || strided) && ctx->queue_kernel && ctx->early_exit && L >= 1 && levels <= 8 &&
        ((!lev && kk <= 254) || (lev && kk == 2 && ctx->lev2_closed && !lev_generic))) {
        // (targets of more slots than the queue kernel's four passes hold, up to 4095, are the walk's too)
        if (int rc = build_line_tables(ctx))
            return rc;
        use_lines = ctx->lw_blocks > 0;
    }
    if (ws == 4 && (use_dense || !strided || !(use_lines || use_queue || lev_queue_il)))
        return fail(ctx, WD_ERR_UNSUPPORTED,
                    "the interleaved layout is read by the queue kernel and the line walk only (equality, Hamming, "
                    "Levenshtein <= 3)");

    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    const bool timed = ctx->profile > 0 && (ctx->profile_seq++ % ctx->profile) == 0;
    if (timed) {
        if (!ctx->free_events.empty()) {
            ev = ctx->free_events.back();
            ctx->free_events.pop_back();
        } else {
            WD_HIP(ctx, hipEventCreate(&ev.first));
            WD_HIP(ctx, hipEventCreate(&ev.second));
        }
        WD_HIP(ctx, hipEventRecord(ev.first, ctx->stream));
    }
    if (use_dense) {
        int rc = launch_dense(ctx, a, n_tiles, N, strided, n_plane_ptrs, tile_chunk, lev2);
        if (rc)
            return rc;
        snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "dense chain v%d, %s%s (k_dense_sig .. k_dense_reduce)",
                 kDenseChainVersion, lev2 ? "Levenshtein <= 2" : (kk > 0 ? "Hamming" : "equality"),
                 ctx->dense_sym_on ? ", pairs from one end" : "");
    } else if (use_lines) {
        const int first = kk <= 0 ? 2 : (kk == 1 ? 3 : (kk == 2 ? 5 : (kk == 3 ? 6 : 8)));
        const int rc = launch_lines(ctx, a, n_tiles, lev, first, strided);
        if (rc)
            return rc;
    } else if (use_queue) {
        queue_view(ctx, a);
        launch_queue(ctx, a, grid, strided);
    } else if (!lev) {
        if (strided)
            launch_ham<true>(ctx, a, grid);
        else
            launch_ham<false>(ctx, a, grid);
    } else if (lev && !lev_generic && kk / 2 <= 3 && ctx->queue_kernel && ctx->early_exit &&
               ctx->k_max <= (int64_t)kMaxPasses * kPass) {
        // Levenshtein <= k, k = 2..7 (the reference's default is 2): queue kernel, DP state in
        // the queue entries
        const int h = kk / 2;
        queue_view(ctx, a);
        launch_queue_lev(ctx, a, grid, strided, h);
    } else if (lev_generic) {
        const int h = kk / 2;
        const size_t lds = lev_generic_lds_bytes(L, h);
        snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "k_scan_lev_generic<%s>", strided ? "true" : "false");
        if (strided)
            hipLaunchKernelGGL((k_scan_lev_generic<true>), grid, dim3(kWave), lds, ctx->stream, a, h);
        else
            hipLaunchKernelGGL((k_scan_lev_generic<false>), grid, dim3(kWave), lds, ctx->stream, a, h);
    } else {
        const int h = kk / 2;
        if (h <= 1) launch_lev<1>(ctx, a, grid, strided);
        else if (h <= 2) launch_lev<2>(ctx, a, grid, strided);
        else if (h <= 3) launch_lev<3>(ctx, a, grid, strided);
        else if (h <= 4) launch_lev<4>(ctx, a, grid, strided);
        else if (h <= 6) launch_lev<6>(ctx, a, grid, strided);
        else launch_lev<8>(ctx, a, grid, strided);
    }
    WD_HIP(ctx, hipGetLastError());
    if (timed) {
        WD_HIP(ctx, hipEventRecord(ev.second, ctx->stream));
        ctx->events.push_back(ev);
    }
    return WD_OK;
} WD_CATCH

int wd_scan_status(wd_ctx *ctx)
{
    if (!ctx)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    WD_HIP(ctx, hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    WD_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(uint32_t), ctx->stream));
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (*ctx->h_status & kStatusEmptyLevel)
        return fail(ctx, WD_ERR_EMPTY_LEVEL, "a target with a valid centre has an empty level");
    return WD_OK;
}

int wd_count_tiles(wd_ctx *ctx, int n_tiles, int L, int mode, int k, const uint8_t *const *planes,
                   const uint8_t *const *filter, int64_t N, int64_t *out_tile,
                   uint32_t *out_per_target)
try {
    if (!ctx || !out_tile)
        return WD_ERR_ARG;
    if (!ctx->has_targets)
        return fail(ctx, WD_ERR_STATE, "wd_set_targets has not been called");
    if (n_tiles < 0)
        return fail(ctx, WD_ERR_ARG, "bad n_tiles");
    if (bind_device(ctx))
        return WD_ERR_HIP;
    const size_t ncnt = 1 + 5 * (size_t)ctx->levels;
    const size_t n_out = std::max<size_t>(1, (size_t)n_tiles * ncnt);
    const size_t n_pt = std::max<size_t>(1, (size_t)n_tiles * ctx->T * ctx->levels);
    int rc = grow(ctx, ctx->d_out_tile, ctx->d_out_tile_cap, n_out);
    if (rc)
        return rc;
    if (out_per_target) {
        rc = grow(ctx, ctx->d_out_pt, ctx->d_out_pt_cap, n_pt);
        if (rc)
            return rc;
    }
    // Planes / filters may also be handed over in host memory (a numpy array's buffer, the bytes
    // of a file just read): those are copied to a staging area first.  Convenient for a caller
    // with no GPU runtime of its own; a caller that cares about time keeps its planes resident.
    std::vector<const uint8_t *> dev_planes, dev_filter;
    if (n_tiles > 0 && planes && filter && L >= 0 && N >= 0) {
        auto on_device = [](const void *p) {
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, p) != hipSuccess) {
                (void)hipGetLastError();                     // plain malloc memory is simply unknown to HIP
                return false;
            }
            return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
        };
        const size_t n_planes = (size_t)n_tiles * (size_t)L;
        const size_t n_pad = ((size_t)N + 255) & ~(size_t)255;
        std::vector<char> host(n_planes + (size_t)n_tiles, 0);
        size_t n_host = 0;
        for (size_t i = 0; i < n_planes + (size_t)n_tiles; i++) {
            const uint8_t *p = i < n_planes ? planes[i] : filter[i - n_planes];
            if (!p)
                return fail(ctx, WD_ERR_ARG, "null plane/filter pointer");
            if (N > 0 && !on_device(p)) {
                host[i] = 1;
                n_host++;
            }
        }
        if (n_host) {
            rc = grow(ctx, ctx->d_stage, ctx->d_stage_cap, n_host * n_pad);
            if (rc)
                return rc;
            dev_planes.assign(planes, planes + n_planes);
            dev_filter.assign(filter, filter + n_tiles);
            size_t slot = 0;
            for (size_t i = 0; i < n_planes + (size_t)n_tiles; i++) {
                if (!host[i])
                    continue;
                uint8_t *d = ctx->d_stage + slot++ * n_pad;
                const uint8_t *src = i < n_planes ? planes[i] : filter[i - n_planes];
                WD_HIP(ctx, hipMemcpyAsync(d, src, (size_t)N, hipMemcpyHostToDevice, ctx->stream));
                (i < n_planes ? dev_planes[i] : dev_filter[i - n_planes]) = d;
            }
            planes = dev_planes.data();
            filter = dev_filter.data();
        }
    }
    rc = wd_scan_async(ctx, n_tiles, L, mode, k, planes, filter, N, (int64_t *)ctx->d_out_tile,
                       out_per_target ? ctx->d_out_pt : nullptr);
    if (rc)
        return rc;
    rc = wd_scan_status(ctx);
    if (rc)
        return rc;
    if (n_tiles > 0) {
        WD_HIP(ctx, hipMemcpy(out_tile, ctx->d_out_tile, (size_t)n_tiles * ncnt * sizeof(int64_t), hipMemcpyDeviceToHost));
        if (out_per_target && ctx->T > 0 && ctx->levels > 0)
            WD_HIP(ctx, hipMemcpy(out_per_target, ctx->d_out_pt,
                                  (size_t)n_tiles * ctx->T * ctx->levels * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return WD_OK;
} WD_CATCH

// ---- targets from coordinates ----------------------------------------------------------
int wd_targets_from_coords(wd_ctx *ctx, const int32_t *x, const int32_t *y, int64_t n,
                           const int32_t *centres, int64_t n_centres, int levels,
                           const int32_t *max_dists, int64_t *P_out)
try {
    if (!ctx || !x || !y || n <= 0 || levels < 1 || levels > kMaxLevels || !max_dists)
        return fail(ctx, WD_ERR_ARG, "bad coordinates, levels or ring table");
    if (!centres)
        n_centres = n;
    if (n_centres < 0 || n_centres > 0x7FFFFFFF)
        return fail(ctx, WD_ERR_ARG, "bad number of centres");
    for (int r = 0; r <= levels; r++)
        if (max_dists[r] < 0 || max_dists[r] > 30000 || (r && max_dists[r] <= max_dists[r - 1]))
            return fail(ctx, WD_ERR_ARG, "ring boundaries must be increasing and <= 30000");
    if (centres)
        for (int64_t i = 0; i < n_centres; i++)
            if (centres[i] < 0 || centres[i] >= n)
                return fail(ctx, WD_ERR_INDEX, "centre outside the s.locs table");
    if (bind_device(ctx))
        return WD_ERR_HIP;
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int T = (int)n_centres;
    int32_t *d_x = nullptr, *d_y = nullptr, *d_counts = nullptr;
    int32_t *d_off = nullptr, *d_nbr = nullptr, *d_centre = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_x); (void)hipFree(d_y); (void)hipFree(d_counts);
    };
    auto bail = [&](int code, const std::string &msg) {
        cleanup();
        (void)hipFree(d_off); (void)hipFree(d_nbr); (void)hipFree(d_centre);
        return fail(ctx, code, msg);
    };
#define WD_GEN_HIP(call)                                                                  \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return bail(e_ == hipErrorOutOfMemory ? WD_ERR_NOMEM : WD_ERR_HIP,            \
                        std::string(#call) + ": " + hipGetErrorString(e_));               \
    } while (0)
    WD_GEN_HIP(hipMalloc((void **)&d_x, (size_t)n * 4));
    WD_GEN_HIP(hipMalloc((void **)&d_y, (size_t)n * 4));
    WD_GEN_HIP(hipMemcpy(d_x, x, (size_t)n * 4, hipMemcpyHostToDevice));
    WD_GEN_HIP(hipMemcpy(d_y, y, (size_t)n * 4, hipMemcpyHostToDevice));
    WD_GEN_HIP(hipMalloc((void **)&d_centre, std::max<size_t>(1, T) * 4));
    if (centres) {
        WD_GEN_HIP(hipMemcpy(d_centre, centres, (size_t)T * 4, hipMemcpyHostToDevice));
    } else {
        std::vector<int32_t> iota((size_t)T);
        for (int i = 0; i < T; i++)
            iota[i] = i;
        WD_GEN_HIP(hipMemcpy(d_centre, iota.data(), (size_t)T * 4, hipMemcpyHostToDevice));
    }
    WD_GEN_HIP(hipMalloc((void **)&d_counts, std::max<size_t>(1, (size_t)T * levels) * 4));
    WD_GEN_HIP(hipMemsetAsync(ctx->d_tblflags, 0, sizeof(uint32_t), ctx->stream));
    GenArgs a;
    a.x = d_x;
    a.y = d_y;
    a.centres = centres ? d_centre : nullptr;
    a.n = n;
    a.n_centres = T;
    a.levels = levels;
    for (int r = 0; r <= levels; r++)
        a.md2[r] = max_dists[r] * max_dists[r];
    a.counts = d_counts;
    a.lvl_off = nullptr;
    a.nbr = nullptr;
    a.status = ctx->d_tblflags;
    std::vector<int32_t> off((size_t)T * (levels + 1) + 1, 0);
    int64_t P = 0;
    if (T > 0) {
        hipLaunchKernelGGL((k_gen_rings<false>), dim3(T), dim3(kBlock), 0, ctx->stream, a);
        WD_GEN_HIP(hipGetLastError());
        std::vector<int32_t> counts((size_t)T * levels);
        WD_GEN_HIP(hipMemcpyAsync(counts.data(), d_counts, counts.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        WD_GEN_HIP(hipMemcpyAsync(ctx->h_status, ctx->d_tblflags, 4, hipMemcpyDeviceToHost, ctx->stream));
        WD_GEN_HIP(hipStreamSynchronize(ctx->stream));
        if (*ctx->h_status & 2u) {
            (void)hipMemsetAsync(ctx->d_tblflags, 0, sizeof(uint32_t), ctx->stream);
            return bail(WD_ERR_NO_WELLS, "Got no wells for some cluster at some level");
        }
        for (int t = 0; t < T; t++) {
            for (int l = 0; l < levels; l++) {
                off[(size_t)t * (levels + 1) + l] = (int32_t)P;
                P += counts[(size_t)t * levels + l];
                if (P > 0x7FFFFFFF)
                    return bail(WD_ERR_UNSUPPORTED, "more than 2^31 neighbour slots");
            }
            off[(size_t)t * (levels + 1) + levels] = (int32_t)P;
        }
    }
    WD_GEN_HIP(hipMalloc((void **)&d_off, std::max<size_t>(1, (size_t)T * (levels + 1)) * 4));
    WD_GEN_HIP(hipMalloc((void **)&d_nbr, std::max<int64_t>(1, P) * 4));
    if (T > 0) {
        WD_GEN_HIP(hipMemcpy(d_off, off.data(), (size_t)T * (levels + 1) * 4, hipMemcpyHostToDevice));
        a.lvl_off = d_off;
        a.nbr = d_nbr;
        hipLaunchKernelGGL((k_gen_rings<true>), dim3(T), dim3(kBlock), 0, ctx->stream, a);
        WD_GEN_HIP(hipGetLastError());
        WD_GEN_HIP(hipMemcpyAsync(ctx->h_status, ctx->d_tblflags, 4, hipMemcpyDeviceToHost, ctx->stream));
        WD_GEN_HIP(hipStreamSynchronize(ctx->stream));
        if (*ctx->h_status & 4u) {
            (void)hipMemsetAsync(ctx->d_tblflags, 0, sizeof(uint32_t), ctx->stream);
            return bail(WD_ERR_UNSUPPORTED, "a target has more than 2048 wells inside the outermost ring");
        }
    }
#undef WD_GEN_HIP
    cleanup();
    // install as the context's targets (as wd_set_targets would)
    (void)hipFree(ctx->d_centre);
    (void)hipFree(ctx->d_lvl_off);
    (void)hipFree(ctx->d_nbr);
    ctx->d_centre = d_centre;
    ctx->d_lvl_off = d_off;
    ctx->d_nbr = d_nbr;
    ctx->T = T;
    ctx->levels = levels;
    ctx->P = P;
    ctx->idx_min = 0;
    ctx->idx_max = n - 1 >= 0 && T > 0 ? n - 1 : -1;   // every emitted index lies inside the table
    ctx->has_empty_level = false;
    ctx->k_max = 0;
    for (int t = 0; t < T; t++)
        ctx->k_max = std::max<int64_t>(ctx->k_max, (int64_t)off[(size_t)t * (levels + 1) + levels] -
                                                        off[(size_t)t * (levels + 1)]);
    set_group_bases(ctx, off.data(), T, levels);
    if (centres) {
        // (the rows of the table: wells at the first well's height)
        long long row_len = 0;
        while (row_len < n && y[row_len] == y[0])
            row_len++;
        if (int rc = install_sorted_view(ctx, centres, off.data(), T, levels, row_len < n ? row_len : 0))
            return rc;
    } else {
        install_sorted_view(ctx, nullptr, nullptr, 0, levels, 0);    // every well a centre: sorted as it is
    }
    ctx->has_targets = true;
    if (P_out)
        *P_out = P;
    return WD_OK;
} WD_CATCH

int wd_targets_info(wd_ctx *ctx, int *T, int *levels, int64_t *P)
{
    if (!ctx)
        return WD_ERR_ARG;
    if (!ctx->has_targets)
        return fail(ctx, WD_ERR_STATE, "no targets set");
    if (T) *T = ctx->T;
    if (levels) *levels = ctx->levels;
    if (P) *P = ctx->P;
    return WD_OK;
}

int wd_get_targets(wd_ctx *ctx, int32_t *centre, int32_t *lvl_off, int32_t *nbr)
try {
    if (!ctx)
        return WD_ERR_ARG;
    if (!ctx->has_targets)
        return fail(ctx, WD_ERR_STATE, "no targets set");
    if (bind_device(ctx))
        return WD_ERR_HIP;
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (centre && ctx->T)
        WD_HIP(ctx, hipMemcpy(centre, ctx->d_centre, (size_t)ctx->T * 4, hipMemcpyDeviceToHost));
    if (lvl_off && ctx->T)
        WD_HIP(ctx, hipMemcpy(lvl_off, ctx->d_lvl_off, (size_t)ctx->T * (ctx->levels + 1) * 4, hipMemcpyDeviceToHost));
    if (nbr && ctx->P)
        WD_HIP(ctx, hipMemcpy(nbr, ctx->d_nbr, (size_t)ctx->P * 4, hipMemcpyDeviceToHost));
    return WD_OK;
} WD_CATCH

}  // extern "C"

