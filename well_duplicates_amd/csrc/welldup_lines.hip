// welldup_lines.hip - the line walk (scan_lines.inc): its tables and the launches of k_scan_lines.  For
// sampled targets with large neighbourhoods (BASELINE configs[3]); called from wd_scan_async.
#include "wd_ctx.h"

#ifndef WD_UNIT_ID
#define WD_UNIT_ID "unknown"
#endif
namespace wd { const char *unit_id_lines() { return WD_UNIT_ID; } }      // hash of this unit's sources (wd_build_id)

using namespace wd;

namespace {

#include "device_common.inc"
#include "scan_sequential.inc"
#include "lev2_stream.inc"
#include "scan_queue.inc"
#include "scan_lines.inc"

}  // namespace

namespace wd {

// k_scan_lines, equality / Levenshtein <= 2):
//     7 rings, 163 slots per target:  2 500 targets 0.191 -> 0.188 / 0.515 -> 0.390 ms;  5 000: 0.348 -> 0.314 /
//                                     0.895 -> 0.617;  10 000 (BASELINE configs[3]): 0.57 -> 0.50 / 1.43 -> 0.99
//     5 rings,  86 slots per target:  2 500 targets (the bench workload) 0.124 -> 0.136 / 0.262 -> 0.277;  5 000:
//                                     0.208 -> 0.239 / 0.442 -> 0.483;  20 000: 0.606 -> 0.623 / 1.241 -> 1.207
// It is the size of the neighbourhoods that decides, not how densely the targets lie: the queue kernel pays
// per (target, pass of 127 slots) - a target of 163 slots is two passes, the second a quarter full - the line
// walk per pair, plus a prologue per (target, block) that 86 pairs do not amortise.  Hence: the line walk for
// targets of more than one pass on average.
bool line_walk_wanted(const wd_ctx *ctx)
{
    if (ctx->line_walk >= 0)
        return ctx->line_walk != 0;
    return ctx->T >= 512 && ctx->P >= (1 << 16) && ctx->P <= (1ll << 24) && ctx->P > (int64_t)kPass * ctx->T;
}


void drop_line_tables(wd_ctx *ctx)
{
    (void)hipFree(ctx->d_lw_well);
    (void)hipFree(ctx->d_lw_meta);
    (void)hipFree(ctx->d_lw_btgt);
    (void)hipFree(ctx->d_lw_blk);
    (void)hipFree(ctx->d_lw_bcen);
    (void)hipFree(ctx->d_lw_boff);
    ctx->d_lw_bcen = ctx->d_lw_boff = nullptr;
    ctx->d_lw_well = nullptr;
    ctx->d_lw_meta = ctx->d_lw_btgt = nullptr;
    ctx->d_lw_blk = nullptr;
    ctx->lw_blocks = -1;
}


// The line walk's tables (scan_lines.inc), built on the first scan that wants them: every (target, slot)
// pair of the current targets, sorted by neighbour well, cut into blocks of at most kLwPairs pairs that
// involve at most kLwTargets targets.  lw_blocks = 0 if the walk does not apply (an empty ring, more than
// 4095 slots in a target, no pairs, too many of them).
int build_line_tables(wd_ctx *ctx)
{
    if (ctx->lw_blocks >= 0)
        return WD_OK;
    ctx->lw_blocks = 0;                      // "does not apply", unless the end of this function is reached
    const int T = ctx->T, levels = ctx->levels;
    const int64_t P = ctx->P;
    if (T < 1 || levels < 1 || P < 1 || P > (1ll << 26) || ctx->has_empty_level || ctx->k_max > 4095)
        return WD_OK;
    const size_t row = (size_t)levels + 1;
    std::vector<int32_t> off((size_t)T * row), nbr((size_t)P), cen((size_t)T);
    // (a HIP error is not "does not apply": the tables stay unbuilt, -1, and the next scan tries again)
#define WD_LW_HIP(call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            drop_line_tables(ctx);                                                        \
            return fail(ctx, e_ == hipErrorOutOfMemory ? WD_ERR_NOMEM : WD_ERR_HIP,       \
                        std::string(#call) + ": " + hipGetErrorString(e_));               \
        }                                                                                 \
    } while (0)
    WD_LW_HIP(hipStreamSynchronize(ctx->stream));
    WD_LW_HIP(hipMemcpy(cen.data(), ctx->d_centre, cen.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    WD_LW_HIP(hipMemcpy(off.data(), ctx->d_lvl_off, off.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    WD_LW_HIP(hipMemcpy(nbr.data(), ctx->d_nbr, nbr.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    // (well, target, slot) of every pair; a target's slots are [off[t][0], off[t][levels]) of nbr
    struct Pair { int32_t well, t; uint32_t slot; };
    std::vector<Pair> pairs;
    pairs.reserve((size_t)P);
    for (int t = 0; t < T; t++) {
        const int32_t b = off[(size_t)t * row], e = off[(size_t)t * row + levels];
        if (b < 0 || e > P || e < b)
            return WD_OK;
        for (int32_t i = b; i < e; i++)
            pairs.push_back(Pair{nbr[(size_t)i], t, (uint32_t)(i - b)});
    }
    if (pairs.empty())
        return WD_OK;
    {
        // by well, stable (equal wells stay in target order): three counting passes of 11 bits - a comparison
        // sort of a few million pairs would be a tenth of a second of every run's start-up
        std::vector<Pair> tmp(pairs.size());
        int32_t lo_well = pairs[0].well;
        for (const Pair &q : pairs)
            lo_well = std::min(lo_well, q.well);
        for (int pass = 0; pass < 3; pass++) {
            std::vector<size_t> cnt(2049, 0);
            const int sh = 11 * pass;
            auto digit = [&](const Pair &q) { return (size_t)(((uint32_t)(q.well - lo_well) >> sh) & 2047u); };
            for (const Pair &q : pairs)
                cnt[digit(q) + 1]++;
            for (int d = 0; d < 2048; d++)
                cnt[(size_t)d + 1] += cnt[(size_t)d];
            for (const Pair &q : pairs)
                tmp[cnt[digit(q)]++] = q;
            pairs.swap(tmp);
        }
        // (33 bits of spread would need a fourth pass: wells are int32 and tiles hold a few million)
        if ((uint32_t)(pairs.back().well - lo_well) >> 31)
            return WD_OK;
        for (size_t i = 1; i < pairs.size(); i++)
            if (pairs[i - 1].well > pairs[i].well) {                 // (spread above 2^33 cannot happen; belt and braces)
                std::stable_sort(pairs.begin(), pairs.end(), [](const Pair &x, const Pair &y) { return x.well < y.well; });
                break;
            }
    }
    const size_t n = pairs.size();
    std::vector<int32_t> well(n);
    std::vector<uint32_t> meta(n), btgt;
    std::vector<int4> blk;
    std::vector<int> local((size_t)T, -1), seen_in((size_t)T, -1);
    std::vector<char> counted((size_t)T, 0);
    size_t first = 0;
    const size_t per_block = (size_t)(ctx->line_pairs > 0 ? ctx->line_pairs : kLwPairs);
    int tmax = 0;
    while (first < n) {
        const int b = (int)blk.size();
        const size_t tgt0 = btgt.size();
        size_t i = first;
        for (; i < n && i - first < per_block; i++) {
            const int t = pairs[i].t;
            if (seen_in[(size_t)t] != b) {
                if (btgt.size() - tgt0 == (size_t)kLwTargets)
                    break;                                       // the block's target table is full
                seen_in[(size_t)t] = b;
                local[(size_t)t] = (int)(btgt.size() - tgt0);
                btgt.push_back((uint32_t)t | (counted[(size_t)t] ? 0u : 0x80000000u));   // the first block to see it owns it
                counted[(size_t)t] = 1;
            }
            well[i] = pairs[i].well;
            meta[i] = ((uint32_t)local[(size_t)t] << 16) | pairs[i].slot;
        }
        blk.push_back(make_int4((int)first, (int)(i - first), (int)tgt0, (int)(btgt.size() - tgt0)));
        tmax = std::max(tmax, (int)(btgt.size() - tgt0));
        first = i;
    }
    ctx->lw_tmax = (tmax + 3) & ~3;
    // (a target without a single pair would never be counted: has_empty_level excludes it)
    std::vector<int32_t> bcen(btgt.size()), boff(btgt.size() * row);
    for (size_t i = 0; i < btgt.size(); i++) {
        const size_t t = (size_t)(btgt[i] & 0x7FFFFFFFu);
        bcen[i] = cen[t];
        memcpy(&boff[i * row], &off[t * row], row * sizeof(int32_t));
    }
    // all five tables or none: a failure half way frees what it has
    auto upload = [&](auto *&dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc((void **)&dst, bytes);
        return e != hipSuccess ? e : hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    };
    WD_LW_HIP(upload(ctx->d_lw_well, well.data(), n * sizeof(int32_t)));
    WD_LW_HIP(upload(ctx->d_lw_meta, meta.data(), n * sizeof(uint32_t)));
    WD_LW_HIP(upload(ctx->d_lw_btgt, btgt.data(), btgt.size() * sizeof(uint32_t)));
    WD_LW_HIP(upload(ctx->d_lw_blk, blk.data(), blk.size() * sizeof(int4)));
    WD_LW_HIP(upload(ctx->d_lw_bcen, bcen.data(), bcen.size() * sizeof(int32_t)));
    WD_LW_HIP(upload(ctx->d_lw_boff, boff.data(), boff.size() * sizeof(int32_t)));
#undef WD_LW_HIP
    ctx->lw_blocks = (int)blk.size();
    return WD_OK;
}

// k_scan_lines for equality / Hamming <= k (first round of `first` cycles) or Levenshtein <= 2 (closed form)
template <bool STRIDED>
int launch_lines_t(wd_ctx *ctx, const ScanArgs &sa, int n_tiles, bool lev2, int first)
{
    LineArgs a;
    a.s = sa;
    a.s.perm = nullptr;
    a.pw = ctx->d_lw_well;
    a.pm = ctx->d_lw_meta;
    a.blk = ctx->d_lw_blk;
    a.btgt = ctx->d_lw_btgt;
    a.bcen = ctx->d_lw_bcen;
    a.boff = ctx->d_lw_boff;
    a.n_blk = ctx->lw_blocks;
    a.tmax = ctx->lw_tmax;
    a.mask_stride = (((long long)sa.T + 3) / 4 + 31) & ~31ll;
    if (int rc = dense_reserve(ctx, ctx->d_mask, ctx->mask_cap, (size_t)n_tiles * (size_t)a.mask_stride, "hit masks"))
        return rc;
    a.mask = ctx->d_mask;
    WD_HIP(ctx, hipMemsetAsync(a.mask, 0, (size_t)n_tiles * (size_t)a.mask_stride * sizeof(uint32_t), ctx->stream));
    if (sa.out_per_target)
        WD_HIP(ctx, hipMemsetAsync(sa.out_per_target, 0, (size_t)n_tiles * sa.T * sa.levels * sizeof(uint32_t), ctx->stream));
    const long long nblocks = (long long)ctx->lw_blocks * n_tiles;
    if (nblocks > 0x7FFFFFFFll)
        return fail(ctx, WD_ERR_UNSUPPORTED, "grid too large");
    const dim3 grid((unsigned)nblocks);
    const size_t lds = (size_t)scan_lines_lds_dwords(sa.levels, ctx->lw_tmax) * sizeof(uint32_t);
#define WD_LAUNCH_L(B1_, LEVH_)                                                                          \
    do {                                                                                                 \
        snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "k_scan_lines<%s, %d, %d>", STRIDED ? "true" : "false", \
                 (int)(B1_), (int)(LEVH_));                                                              \
        hipLaunchKernelGGL((k_scan_lines<STRIDED, (B1_), (LEVH_)>), grid, dim3(kBlock), lds, ctx->stream, a); \
    } while (0)
    if constexpr (STRIDED) {
        if (ctx->well_stride == 4) {        // interleaved: the first round is one dword per pair, two for Levenshtein <= 2
            if (lev2) {
                snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "k_scan_lines<true, 8, %d, 4>", kLev2Closed);
                hipLaunchKernelGGL((k_scan_lines<true, 8, kLev2Closed, 4>), grid, dim3(kBlock), lds, ctx->stream, a);
            } else {
                snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "k_scan_lines<true, 4, 0, 4>");
                hipLaunchKernelGGL((k_scan_lines<true, 4, 0, 4>), grid, dim3(kBlock), lds, ctx->stream, a);
            }
            return WD_OK;
        }
    }
    if (lev2) {
        WD_LAUNCH_L(5, kLev2Closed);
    } else {
        switch (first) {
        case 2: WD_LAUNCH_L(2, 0); break;
        case 3: WD_LAUNCH_L(3, 0); break;
        case 5: WD_LAUNCH_L(5, 0); break;
        case 6: WD_LAUNCH_L(6, 0); break;
        default: WD_LAUNCH_L(8, 0); break;
        }
    }
#undef WD_LAUNCH_L
    return WD_OK;
}

int launch_lines(wd_ctx *ctx, const ScanArgs &sa, int n_tiles, bool lev2, int first, bool strided)
{
    return strided ? launch_lines_t<true>(ctx, sa, n_tiles, lev2, first) : launch_lines_t<false>(ctx, sa, n_tiles, lev2, first);
}

}  // namespace wd
