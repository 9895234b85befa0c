// welldup_queue.hip - launches of k_scan_q (scan_queue.inc): the default kernel for sampled targets -
// equality / Hamming <= k, and Levenshtein <= 7 with the DP (or, for k = 2, the closed form's 12 bits) in
// the queue entries; plane-per-cycle and interleaved-by-four input.  Called from wd_scan_async.
#include "wd_ctx.h"

#ifndef WD_UNIT_ID
#define WD_UNIT_ID "unknown"
#endif
namespace wd { const char *unit_id_queue() { return WD_UNIT_ID; } }      // hash of this unit's sources (wd_build_id)

using namespace wd;

namespace {

#include "device_common.inc"
#include "scan_sequential.inc"
#include "lev2_stream.inc"
#include "scan_queue.inc"

// One launch of the queue kernel; its template name - as the code object spells it - is kept for
// wd_last_kernel(), so that a counter profile can be tied to the kernel that really ran.
#define WD_LAUNCH_Q(STR, B1_, LEVH_, WS_, LDS_)                                                          \
    do {                                                                                                 \
        snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "k_scan_q<%s, %d, %d, %d>%s", (STR) ? "true" : "false", \
                 (int)(B1_), (int)(LEVH_), (int)(WS_), a.perm ? ", targets sorted by centre" : "");      \
        hipLaunchKernelGGL((k_scan_q<(STR), (B1_), (LEVH_), (WS_)>), grid, dim3(kBlock), (LDS_), ctx->stream, a); \
    } while (0)


template <bool STRIDED>
int launch_queue_t(wd_ctx *ctx, const ScanArgs &a, dim3 grid)
{
    const size_t lds = (size_t)scan_q_lds_dwords(a.levels, a.tpb) * sizeof(uint32_t);
    if (STRIDED && ctx->well_stride == 4) {                 // interleaved: one dword = the first round
        WD_LAUNCH_Q(true, 4, 0, STRIDED ? 4 : 1, lds);
        return 0;
    }
    // A random neighbour survives r cycles with <= k mismatches with probability
    // sum_{i<=k} C(r,i) 0.75^i 0.25^(r-i); the first round should leave a few percent alive.
    int first = ctx->queue_first;
    if (first == 0)
        first = a.k <= 0 ? 2 : (a.k == 1 ? 3 : (a.k == 2 ? 5 : (a.k == 3 ? 6 : 8)));   // measured on MI355X
    switch (first) {
    case 1: WD_LAUNCH_Q(STRIDED, 1, 0, 1, lds); break;
    case 2: WD_LAUNCH_Q(STRIDED, 2, 0, 1, lds); break;
    case 3: WD_LAUNCH_Q(STRIDED, 3, 0, 1, lds); break;
    case 4: WD_LAUNCH_Q(STRIDED, 4, 0, 1, lds); break;
    case 5: WD_LAUNCH_Q(STRIDED, 5, 0, 1, lds); break;
    case 6: WD_LAUNCH_Q(STRIDED, 6, 0, 1, lds); break;
    case 7: WD_LAUNCH_Q(STRIDED, 7, 0, 1, lds); break;
    default: WD_LAUNCH_Q(STRIDED, 8, 0, 1, lds); break;
    }
    return 0;
}

template <bool STRIDED, int H>
void launch_queue_lev_t(wd_ctx *ctx, const ScanArgs &a, dim3 grid)
{
    if constexpr (H == 1) {
        // k = 2, the reference's default: the streaming closed form (lev2_stream.inc) instead of the
        // banded DP - 8-byte queue entries, six waves per SIMD ("lev2_closed" = 0 keeps the DP: tests)
        if (a.k == 2 && ctx->lev2_closed) {
            const size_t lds2 = (size_t)scan_q_lds_dwords(a.levels, a.tpb, 2) * sizeof(uint32_t);
            if constexpr (STRIDED) {
                if (ctx->well_stride == 4) {
                    WD_LAUNCH_Q(true, 8, kLev2Closed, 4, lds2);
                    return;
                }
            }
            switch (ctx->queue_first) {
            case 4: WD_LAUNCH_Q(STRIDED, 4, kLev2Closed, 1, lds2); break;
            case 6: WD_LAUNCH_Q(STRIDED, 6, kLev2Closed, 1, lds2); break;
            default: WD_LAUNCH_Q(STRIDED, 5, kLev2Closed, 1, lds2); break;
            }
            return;
        }
    }
    const size_t lds = (size_t)scan_q_lds_dwords(a.levels, a.tpb, 4) * sizeof(uint32_t);
    if constexpr (STRIDED && H == 1) {
        if (ctx->well_stride == 4) {                    // interleaved: the first round is two dwords per well
            WD_LAUNCH_Q(true, 8, 1, 4, lds);
            return;
        }
    }
    // The first round (every neighbour, lane per slot) is shorter than the sequential kernel's: since
    // the survivors are drained in rounds of 1, 2, 4, 8 cycles, what the first round leaves alive is
    // cheap, and every cycle it reads costs a line per row segment.  Measured for H = 1 on the bench
    // workload (k = 2): 7 cycles 0.317 ms, 6: 0.291, 5: 0.280, 4: 0.303, 3: 0.349.
    // (Wider bands: 10 - 2 cycles 0.453 ms against 0.484 at k = 4, but 0.550 against 0.542 at k = 5, and
    // worse beyond - only k = 4 takes the shorter round.)
    constexpr int first_even = lev_first(H) - (H <= 2 ? 2 : 0), first_odd = lev_first(H) + 1 - (H == 1 ? 2 : 0);
    // an odd threshold (k = 2H + 1) keeps random neighbours alive about one cycle longer
    if (a.k & 1)
        WD_LAUNCH_Q(STRIDED, first_odd, H, 1, lds);
    else
        WD_LAUNCH_Q(STRIDED, first_even, H, 1, lds);
}


}  // namespace

namespace wd {

int launch_queue(wd_ctx *ctx, const ScanArgs &a, dim3 grid, bool strided)
{
    return strided ? launch_queue_t<true>(ctx, a, grid) : launch_queue_t<false>(ctx, a, grid);
}

// Levenshtein <= k, k = 2 .. 7, band half-width h = k / 2 (1 .. 3)
void launch_queue_lev(wd_ctx *ctx, const ScanArgs &a, dim3 grid, bool strided, int h)
{
    if (h <= 1) { if (strided) launch_queue_lev_t<true, 1>(ctx, a, grid); else launch_queue_lev_t<false, 1>(ctx, a, grid); }
    else if (h == 2) { if (strided) launch_queue_lev_t<true, 2>(ctx, a, grid); else launch_queue_lev_t<false, 2>(ctx, a, grid); }
    else { if (strided) launch_queue_lev_t<true, 3>(ctx, a, grid); else launch_queue_lev_t<false, 3>(ctx, a, grid); }
}

}  // namespace wd
