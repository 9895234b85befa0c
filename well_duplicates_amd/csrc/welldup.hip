// welldup.hip - MI355X (gfx950 / CDNA4) well-duplicate scanner: kernels + C ABI.
//
// Hot path of EdinburghGenomics/well_duplicates' count_well_duplicates.py:228-265 (target ->
// level -> neighbour compare) fused with the gather of Tile.get_seqs
// (bcl_direct_reader.py:158-220, :352-361) and the integer part of output_writer
// (count_well_duplicates.py:63-106).  Interface: include/welldup.h.
//
// Design (DESIGN.md has the full story)
//   * One wave64 per target, one lane per neighbour slot ("entry").  Everything per-target
//     (centre index, filter bit, ring offsets, the centre's base at each cycle) is
//     wave-uniform; the per-level dup counts fall out of one __ballot + popcount per pass.
//   * The gather is fused into the compare and is *lazy in the cycle direction*: a lane
//     reads its well's byte for the first few cycles only; a neighbour that has already
//     accumulated more than k mismatches (or whose banded edit-distance row is all > k) can
//     never become a duplicate, so its remaining L - few bytes are never fetched.  On
//     sequencing data almost every neighbour dies within 3-4 cycles, which removes ~90 % of
//     the HBM sectors a full gather touches.  "early_exit"=0 gives the full gather.
//   * HBM-bound byte/integer work: no MFMA, no LDS staging of sequences (each byte is used
//     once).  LDS holds the block's tally histogram (LDS atomics), flushed with one global
//     64-bit atomic per counter per block.
//   * No CUDA shims, no dual paths: this file is gfx950 HIP only.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <zlib.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "welldup.h"

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWaves = kBlock / kWave;
constexpr int kMaxLevels = WD_MAX_LEVELS;
constexpr int kCounters = 1 + 5 * kMaxLevels;

constexpr uint32_t kStatusEmptyLevel = 1u;

// -------------------------------------------------------------------------------------
// Synthetic data: device twin of well_duplicates_amd/synth.py (same constants, same bits)
// -------------------------------------------------------------------------------------
constexpr uint64_t K_SEED = 0x9E3779B97F4A7C15ull;
constexpr uint64_t K_LANE = 0xD1B54A32D192ED03ull;
constexpr uint64_t K_TILE = 0x8CB92BA72F3D8DD7ull;
constexpr uint64_t K_CYCLE = 0xDB4F0B9175AE2165ull;
constexpr uint64_t K_CLUSTER = 0xA24BAED4963EE407ull;
constexpr uint64_t SALT_PLANT = 0x5851F42D4C957F2Dull;
constexpr uint64_t SALT_FILTER = 0x2545F4914F6CDD1Dull;

__host__ __device__ inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

__device__ inline uint8_t synth_raw(uint64_t plane_key, uint64_t cluster, uint32_t nocall)
{
    uint64_t h = mix64(plane_key + cluster * K_CLUSTER);
    if ((h & 0xFFFF) < nocall)
        return 0;
    uint32_t base = (uint32_t)(h >> 16) & 3u;
    uint32_t qual = 2u + ((uint32_t)(h >> 18) & 0xFFFFu) % 39u;
    return (uint8_t)((qual << 2) | base);
}

struct SynthPlaneArgs {
    uint8_t *dst;
    int64_t n, row;
    uint64_t key_here, key_next, key_plant;
    uint32_t nocall, plant, far;
    int cycle;
};

__global__ __launch_bounds__(kBlock) void k_synth_plane(SynthPlaneArgs a)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
        uint64_t g = mix64(a.key_plant + (uint64_t)i * K_CLUSTER);
        bool planted = (g & 0xFFFF) < a.plant;
        int64_t delta;
        if (a.far) {
            const uint32_t sel = (uint32_t)(g >> 16) & 7u;
            delta = sel < 3 ? (int64_t)sel + 1 : (int64_t)(sel - 2) * a.row;
        } else {
            const uint32_t sel = ((uint32_t)(g >> 16) & 0xFFu) % 3u;
            delta = sel == 0 ? 1 : (sel == 1 ? a.row : 2 * a.row);
        }
        int64_t src = i - delta;
        planted = planted && src >= 0;
        uint8_t b;
        if (!planted) {
            b = synth_raw(a.key_here, (uint64_t)i, a.nocall);
        } else {
            uint32_t variant = (uint32_t)(g >> 24) & 7u;
            int sub1 = (int)(((uint32_t)(g >> 32) & 0xFFFFu) % 128u);
            int sub2 = (int)(((uint32_t)(g >> 48) & 0xFFFFu) % 128u);
            bool subst = ((variant == 5 || variant == 6) && sub1 == a.cycle) ||
                         (variant == 6 && sub2 == a.cycle);
            if (subst)
                b = synth_raw(a.key_here, (uint64_t)i, a.nocall);
            else if (variant == 7)
                b = synth_raw(a.key_next, (uint64_t)src, a.nocall);
            else
                b = synth_raw(a.key_here, (uint64_t)src, a.nocall);
        }
        a.dst[i] = b;
    }
}

struct SynthFilterArgs {
    uint8_t *dst;
    int64_t n;
    uint64_t key;
    uint32_t pass, noise, dead;
};

__global__ __launch_bounds__(kBlock) void k_synth_filter(SynthFilterArgs a)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
        uint64_t f = mix64(a.key + (uint64_t)i * K_CLUSTER);
        uint8_t b = (!a.dead && (f & 0xFFFF) < a.pass) ? 1 : 0;
        if (a.noise)
            b |= (uint8_t)(((f >> 16) & 1) << 1);
        a.dst[i] = b;
    }
}

// -------------------------------------------------------------------------------------
// Scan kernels
// -------------------------------------------------------------------------------------
struct ScanArgs {
    const uint8_t *const *planes;   // device table: [n_tiles*L], or [n_tiles] bases if strided
    const uint8_t *const *filter;   // device table [n_tiles]
    int64_t stride;                 // bytes between consecutive cycle planes of a tile
    const int32_t *centre;
    const int32_t *lvl_off;
    const int32_t *nbr;
    unsigned long long *out_tile;   // [n_tiles][1 + 5*levels]
    uint32_t *out_per_target;       // nullable [n_tiles][T][levels]
    const struct ScanRare *rare;    // rarely-touched arguments, read only on the rare paths
    const int32_t *nbr_t;           // dense kernel: neighbour lists transposed per 64-target group
    const long long *gbase;         // dense kernel: start of each group's block in nbr_t
    int T, levels, L, k, tpb, early, check_empty, log_hits;
};

// Kept out of the kernel-argument registers: only duplicates and malformed targets need them.
struct ScanRare {
    uint32_t *status;
    wd_hit *hits;
    unsigned long long *hit_count;
    long long hit_cap;
};

// Plane / filter pointers reach the kernels through pointer tables in memory, so the compiler
// only knows them as generic ("flat") addresses: flat loads need a VGPR address pair each and
// force vmcnt(0)+lgkmcnt(0) waits.  They are device-global by contract (include/welldup.h), so
// say so: global_load_ubyte with a scalar base, a 32-bit lane offset and counted vmcnt.
using gbytes = const __attribute__((address_space(1))) uint8_t *;
__device__ inline gbytes as_global(const uint8_t *p) { return (gbytes)p; }

// Cache policy of the plane-byte loads.  On a pure one-byte-per-line gather non-temporal loads
// (`global_load_ubyte ... nt`) reach 52.0 vs 47.1 G lines/s (tools/microbench_modes.hip; sc0 /
// sc1 change nothing and no policy fetches less than the full 128-byte line) - but the scan
// re-reads lines through L1/L2 (both slots of a pass, the centre's line, the drain rounds), and
// with nt it ran 0.163 ms instead of 0.137 ms.  So: default policy; -DWD_NT_LOADS=1 to retry.
#ifndef WD_NT_LOADS
#define WD_NT_LOADS 0
#endif
__device__ inline uint32_t ldb(gbytes p, uint32_t i)
{
#if WD_NT_LOADS
    return __builtin_nontemporal_load(p + i);
#else
    return p[i];
#endif
}

// Symbol code of a BCL byte: 0 -> 4 ('N'), else byte & 3 (bcl_direct_reader.py:352-361).
__device__ inline uint32_t code_of(uint32_t b)
{
    return (b & 3u) | (((b - 1u) >> 29) & 4u);
}

// Bits [lo, hi) of a 64-bit mask, lo/hi clamped to [0, 64].
__device__ inline uint64_t range_mask(int lo, int hi)
{
    lo = lo < 0 ? 0 : (lo > 64 ? 64 : lo);
    hi = hi < 0 ? 0 : (hi > 64 ? 64 : hi);
    uint64_t mh = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
    uint64_t ml = lo >= 64 ? ~0ull : ((1ull << lo) - 1ull);
    return mh & ~ml;
}

// ---- per-entry compare state: Hamming (also equality, k = 0) -------------------------
struct HamState {
    // register budget hint: 7 waves/SIMD (<= 72 VGPRs) measured best on MI355X (8 spills)
    static constexpr int kMinWavesPerSimd = 7;
    int mm;
    __device__ void init(int) { mm = 0; }
    // p = 1-based cycle just pushed; cc/wc = centre / well codes of that cycle
    __device__ void push(int, uint64_t, uint32_t cc, uint32_t wc, int, int) { mm += (cc != wc); }
    __device__ void finish(int, uint64_t, int, int) {}
    __device__ bool alive(int k) const { return mm <= k; }
    __device__ bool dup(int k) const { return mm <= k; }
    __device__ int dist() const { return mm; }
};

// ---- per-entry compare state: Levenshtein <= k by a banded row DP ---------------------
// Equal-length strings: a path of cost <= k never leaves diagonals |j - i| <= H = k / 2, so a
// band of 2H+1 cells per row is exact for every pair with distance <= k and over-estimates
// (never under-estimates) the rest.  Row i needs the well's codes w[i-H .. i+H], so it is
// processed H cycles after cycle i arrives; values saturate at cap = k + 1.
template <int H>
struct LevState {
    static constexpr int kMinWavesPerSimd = 1;
    static constexpr int W = 2 * H + 1;
    int r[W];
    uint64_t wh;   // well codes, newest in bits [0,3)

    __device__ void init(int cap)
    {
#pragma unroll
        for (int d = 0; d < W; d++)
            r[d] = d >= H ? min(d - H, cap) : cap;
        wh = ~0ull;
    }
    // Row i (1-based) with centre code ci; wh's newest code is w[i + H].
    __device__ void row(int i, uint32_t ci, int L, int cap)
    {
        int nw[W];
#pragma unroll
        for (int d = 0; d < W; d++) {
            const int j = i + d - H;
            const uint32_t wj = (uint32_t)(wh >> (3 * (2 * H - d))) & 7u;
            int v = r[d] + (ci != wj ? 1 : 0);
            if (d + 1 < W)
                v = min(v, r[d + 1] + 1);
            if (d > 0)
                v = min(v, nw[d - 1] + 1);
            v = (j == 0) ? i : v;
            v = (j < 0 || j > L) ? cap : v;
            nw[d] = min(v, cap);
        }
#pragma unroll
        for (int d = 0; d < W; d++)
            r[d] = nw[d];
    }
    // ch: centre codes, newest (cycle p) in bits [0,3)
    __device__ void push(int p, uint64_t ch, uint32_t, uint32_t wc, int L, int cap)
    {
        wh = (wh << 3) | wc;
        if (p > H)
            row(p - H, (uint32_t)(ch >> (3 * H)) & 7u, L, cap);
    }
    // after cycle L: rows L-H+1 .. L still need processing; the missing codes never match
    __device__ void finish(int L, uint64_t ch, int cap, int)
    {
#pragma unroll
        for (int q = 1; q <= H; q++) {
            wh = (wh << 3) | 7u;
            ch = (ch << 3) | 6u;
            const int i = L + q - H;
            if (i >= 1)
                row(i, (uint32_t)(ch >> (3 * H)) & 7u, L, cap);
        }
    }
    __device__ bool alive(int k) const
    {
        bool a = false;
#pragma unroll
        for (int d = 0; d < W; d++)
            a = a || (r[d] + (d > H ? d - H : H - d) <= k);
        return a;
    }
    __device__ bool dup(int k) const { return r[H] <= k; }
    __device__ int dist() const { return r[H]; }
};

// One wave = one target at a time; one lane = one neighbour slot (two slots per lane per
// pass, 128 slots per pass).  B1 cycles are read unconditionally, then batches of B2 cycles
// only by lanes that can still become a duplicate.
//
// Latency structure (the kernel is bound by dependent HBM round trips, not by bytes):
//   * the block's target metadata (centre, ring offsets) is staged once into LDS;
//   * while target t's plane bytes are in flight the wave already issues target t+1's filter
//     byte and neighbour-index loads (one-target-ahead software pipeline), so the only
//     dependent HBM trip left per target is the plane gather itself;
//   * the centre's byte is taken from an idle lane of the second slot (idle lanes shadow the
//     centre index) instead of a third load per cycle;
//   * tallies accumulate in registers (lane l = level l) and reach LDS once per wave.
constexpr int kMaxTpb = 64;

template <class State, bool STRIDED, int B1, int B2>
__global__ __launch_bounds__(kBlock, State::kMinWavesPerSimd) void k_scan(ScanArgs a)
{
    __shared__ uint32_t s_cnt[kCounters];
    __shared__ int32_t s_centre[kMaxTpb];
    __shared__ int32_t s_off[kMaxTpb * (kMaxLevels + 1)];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int levels = a.levels;
    const int L = a.L;
    const int k = a.k;
    const int cap = k + 1;
    const int ncnt = 1 + 5 * levels;
    const int chunks = (a.T + a.tpb - 1) / a.tpb;
    const int tile = blockIdx.x / chunks;
    const int chunk = blockIdx.x - tile * chunks;
    const int t_first = chunk * a.tpb;
    const int n_t = min(a.tpb, a.T - t_first);

    for (int i = threadIdx.x; i < ncnt; i += kBlock)
        s_cnt[i] = 0;
    for (int i = threadIdx.x; i < n_t; i += kBlock)
        s_centre[i] = a.centre[t_first + i];
    for (int i = threadIdx.x; i < n_t * (levels + 1); i += kBlock)
        s_off[i] = a.lvl_off[(size_t)t_first * (levels + 1) + i];
    __syncthreads();

    gbytes filt = as_global(a.filter[tile]);
    const uint8_t *const *ptab = STRIDED ? nullptr : a.planes + (size_t)tile * L;
    gbytes base0 = STRIDED ? as_global(a.planes[tile]) : nullptr;
    const int64_t stride = a.stride;
    auto plane_ptr = [&](int j) -> gbytes {
        return STRIDED ? base0 + (int64_t)j * stride : as_global(ptab[j]);
    };

    // per-wave tallies, lane l = level l
    uint32_t acc_wells = 0, acc_dups = 0, acc_hit = 0, acc_first = 0, acc_last = 0, acc_valid = 0;

    // Registers of the target one step ahead: wave-uniform c / off0 / K, the centre's filter
    // byte and the first-pass neighbour indices (the centre index in idle lanes).
    int n_c = 0, n_off0 = 0, n_K = 0;
    uint32_t n_fb = 0, n_i0 = 0, n_i1 = 0;
#define WD_FETCH(TL)                                                                      \
    do {                                                                                  \
        const int tl_ = (TL);                                                             \
        n_c = __builtin_amdgcn_readfirstlane(s_centre[tl_]);                              \
        n_off0 = __builtin_amdgcn_readfirstlane(s_off[tl_ * (levels + 1)]);               \
        n_K = __builtin_amdgcn_readfirstlane(s_off[tl_ * (levels + 1) + levels]) - n_off0; \
        n_fb = filt[(uint32_t)n_c];                                                       \
        n_i0 = lane < n_K ? (uint32_t)a.nbr[n_off0 + lane] : (uint32_t)n_c;               \
        n_i1 = lane + kWave < n_K ? (uint32_t)a.nbr[n_off0 + kWave + lane] : (uint32_t)n_c; \
    } while (0)

    if (wave < n_t)
        WD_FETCH(wave);
    for (int tl = wave; tl < n_t; tl += kWaves) {
        const int t = t_first + tl;
        const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane(n_c);
        const int off0 = __builtin_amdgcn_readfirstlane(n_off0);
        const int K = __builtin_amdgcn_readfirstlane(n_K);
        const bool valid = __builtin_amdgcn_readfirstlane(n_fb) & 1u;   // :236-237
        const uint32_t c_i0 = n_i0, c_i1 = n_i1;
        // the wave's next target (clamped: the last one re-fetches itself, harmlessly)
        const int tl_next = min(tl + kWaves, n_t - 1);
        uint32_t *opt = a.out_per_target
                            ? a.out_per_target + ((size_t)tile * a.T + t) * levels
                            : nullptr;
        if (!valid) {
            WD_FETCH(tl_next);
            if (opt && lane < levels)
                opt[lane] = WD_INVALID_TARGET;
            continue;
        }
        int my_lo = 0, my_hi = 0;   // lane l < levels: ring l+1 is slots [my_lo, my_hi)
        if (lane < levels) {
            my_lo = s_off[tl * (levels + 1) + lane] - off0;
            my_hi = s_off[tl * (levels + 1) + lane + 1] - off0;
        }
        bool skip = false;
        if (a.check_empty) {        // count_well_duplicates.py:249
            if (__ballot(lane < levels && my_hi <= my_lo)) {
                if (lane == 0)
                    atomicOr(a.rare->status, kStatusEmptyLevel);
                skip = true;
            }
        }
        uint32_t my_d = 0;
        bool prefetched = false;

        for (int base = 0; base < K && !skip; base += 2 * kWave) {
            const int e0 = base + lane, e1 = e0 + kWave;
            const bool a0 = e0 < K, a1 = e1 < K;
            uint32_t i0 = c_i0, i1 = c_i1;
            if (base > 0) {
                // idle lanes shadow the centre well: their loads hit the centre's own line
                i0 = a0 ? (uint32_t)a.nbr[off0 + e0] : c;
                i1 = a1 ? (uint32_t)a.nbr[off0 + e1] : c;
            }
            // lane 63 of the second slot is idle (and so reads the centre) unless the pass is full
            const bool centre_in_lane = base + 2 * kWave > K;
            State s0, s1;
            s0.init(cap);
            s1.init(cap);
            uint64_t ch = ~0ull;          // centre codes, newest in bits [0,3)
            bool l0 = a0, l1 = a1;        // still worth loading for
            int j = 0;

            // ---- first batch: unconditional ----
            if (L > 0) {
                uint32_t cb[B1], w0[B1], w1[B1];
#pragma unroll
                for (int q = 0; q < B1; q++) {
                    gbytes p = plane_ptr(min(j + q, L - 1));
                    w0[q] = p[i0];
                    w1[q] = p[i1];
                    if (!centre_in_lane)
                        cb[q] = p[c];
                }
                if (!prefetched) {                  // next target's metadata rides behind
                    WD_FETCH(tl_next);
                    prefetched = true;
                }
#pragma unroll
                for (int q = 0; q < B1; q++) {
                    if (j + q < L) {
                        const uint32_t cbyte = centre_in_lane
                                                   ? (uint32_t)__builtin_amdgcn_readlane((int)w1[q], kWave - 1)
                                                   : cb[q];
                        const uint32_t cc = code_of(cbyte);
                        ch = (ch << 3) | cc;
                        s0.push(j + q + 1, ch, cc, code_of(w0[q]), L, cap);
                        s1.push(j + q + 1, ch, cc, code_of(w1[q]), L, cap);
                    }
                }
                j = min(L, B1);
                if (a.early) {
                    l0 = a0 && s0.alive(k);
                    l1 = a1 && s1.alive(k);
                }
            }
            // ---- later batches: only lanes that can still become a duplicate ----
            while (j < L && __ballot(l0 || l1)) {
                uint32_t cb[B2], w0[B2], w1[B2];
                gbytes pp[B2];
#pragma unroll
                for (int q = 0; q < B2; q++) {
                    pp[q] = plane_ptr(min(j + q, L - 1));
                    cb[q] = pp[q][c];
                }
                if (l0) {
#pragma unroll
                    for (int q = 0; q < B2; q++)
                        w0[q] = pp[q][i0];
                }
                if (l1) {
#pragma unroll
                    for (int q = 0; q < B2; q++)
                        w1[q] = pp[q][i1];
                }
#pragma unroll
                for (int q = 0; q < B2; q++) {
                    if (j + q < L) {
                        const uint32_t cc = code_of(cb[q]);
                        ch = (ch << 3) | cc;
                        if (l0)
                            s0.push(j + q + 1, ch, cc, code_of(w0[q]), L, cap);
                        if (l1)
                            s1.push(j + q + 1, ch, cc, code_of(w1[q]), L, cap);
                    }
                }
                j = min(L, j + B2);
                if (a.early) {
                    l0 = l0 && s0.alive(k);
                    l1 = l1 && s1.alive(k);
                }
            }
            // lanes that made it through all L cycles
            bool d0 = false, d1 = false;
            if (j >= L) {
                if (l0) {
                    s0.finish(L, ch, cap, k);
                    d0 = s0.dup(k);
                }
                if (l1) {
                    s1.finish(L, ch, cap, k);
                    d1 = s1.dup(k);
                }
            }
            const uint64_t m0 = __ballot(d0), m1 = __ballot(d1);
            if (m0 | m1) {
                if (lane < levels) {
                    my_d += __popcll(m0 & range_mask(my_lo - base, my_hi - base));
                    my_d += __popcll(m1 & range_mask(my_lo - base - kWave, my_hi - base - kWave));
                }
                if (a.log_hits) {
                    const ScanRare r = *a.rare;
                    if (d0) {
                        unsigned long long h = atomicAdd(r.hit_count, 1ull);
                        if ((long long)h < r.hit_cap)
                            r.hits[h] = wd_hit{tile, t, off0 + e0, s0.dist()};
                    }
                    if (d1) {
                        unsigned long long h = atomicAdd(r.hit_count, 1ull);
                        if ((long long)h < r.hit_cap)
                            r.hits[h] = wd_hit{tile, t, off0 + e1, s1.dist()};
                    }
                }
            }
        }
        if (!prefetched)
            WD_FETCH(tl_next);
        if (skip)
            continue;

        // ---- tally (count_well_duplicates.py:80-95 as histograms; include/welldup.h) ----
        const uint64_t hm = __ballot(lane < levels && my_d > 0);
        acc_valid += 1;
        acc_wells += (uint32_t)(my_hi - my_lo);
        acc_dups += my_d;
        acc_hit += my_d ? 1u : 0u;
        if (hm) {
            acc_first += (lane == __ffsll((long long)hm) - 1) ? 1u : 0u;
            acc_last += (lane == 63 - __clzll((long long)hm)) ? 1u : 0u;
        }
        if (opt && lane < levels)
            opt[lane] = my_d;
    }

    if (lane < levels) {
        if (acc_wells) atomicAdd(&s_cnt[1 + lane], acc_wells);
        if (acc_dups) atomicAdd(&s_cnt[1 + levels + lane], acc_dups);
        if (acc_hit) atomicAdd(&s_cnt[1 + 2 * levels + lane], acc_hit);
        if (acc_first) atomicAdd(&s_cnt[1 + 3 * levels + lane], acc_first);
        if (acc_last) atomicAdd(&s_cnt[1 + 4 * levels + lane], acc_last);
    }
    if (lane == 0 && acc_valid)
        atomicAdd(&s_cnt[0], acc_valid);
#undef WD_FETCH
    __syncthreads();
    for (int i = threadIdx.x; i < ncnt; i += kBlock) {
        const uint32_t v = s_cnt[i];
        if (v)
            atomicAdd(&a.out_tile[(size_t)tile * ncnt + i], (unsigned long long)v);
    }
}

// -------------------------------------------------------------------------------------
// Queue kernel (Hamming family with early exit): the default for equality / Hamming <= k
// -------------------------------------------------------------------------------------
// k_scan above pays one dependent HBM round trip per target per round, so reading fewer cycles
// in the first round (less traffic) only adds rounds (more latency).  This kernel makes the
// round, not the target, the unit of latency:
//   phase 0  the block stages its targets' metadata in LDS, reads all their filter bytes in
//            one go and builds the list of (valid target, pass) work items;
//   phase 1  each wave streams over its items with a 4-stage software pipeline (neighbour
//            indices two items ahead, plane bytes one item ahead, all loads unconditional so
//            the compiler's counted vmcnt keeps two gathers in flight).  Only B1 cycles are
//            read; neighbours still within k mismatches ("survivors", ~1/4^B1 of them) are
//            compacted into a small per-wave LDS queue {well index, target, slot, mismatches};
//   phase 2  the wave drains its queue in rounds of 2, 4, 8, 8, ... further cycles, one lane
//            per survivor (dense), re-compacting after each round; whoever survives all L
//            cycles is a duplicate and bumps its target's per-level counter in LDS;
//   phase 3  per-target tallies -> block histogram (LDS atomics) -> one global atomic each.
// A pass holds 127 slots: lane 63 of the second slot always shadows the centre well, so the
// centre's byte comes out of the same load as the neighbours'.
constexpr int kPass = 127;
constexpr int kQCap = 256;          // >= kPass: after a drain one pass always fits
constexpr int kMaxPasses = 4;       // host falls back to k_scan for targets with more slots

// ---- survivor queue shared by k_scan_q and k_scan_dense -------------------------------
// A queue entry is {well index, tag}: tag = target-in-block << 24 | mismatches << 16 | slot.
struct QEnv {
    const int32_t *s_centre;      // LDS: centre index of each target of the block
    const int32_t *s_off;         // LDS: ring offsets, levels + 1 per target
    uint32_t *s_d;                // LDS: per-target per-level duplicate counters
    gbytes base0;                 // strided layout: cycle 0 plane of this tile
    const uint8_t *const *ptab;   // pointer-table layout: this tile's L plane pointers
    int64_t stride;
    const ScanRare *rare;
    int levels, L, k, tile, t_first, log_hits;
};

template <bool STRIDED>
__device__ inline gbytes q_plane(const QEnv &v, int j)
{
    return STRIDED ? v.base0 + (int64_t)j * v.stride : as_global(v.ptab[j]);
}

// a duplicate found: bump its target's per-level counter (and the optional hit log)
__device__ inline void q_record_dup(const QEnv &v, int tl, int e, int dist)
{
    const int32_t *o = v.s_off + tl * (v.levels + 1);
    const int rel0 = o[0];
    int lev = 0;
    for (int l = 1; l < v.levels; l++)
        lev += (e >= o[l] - rel0) ? 1 : 0;
    atomicAdd(&v.s_d[tl * v.levels + lev], 1u);
    if (v.log_hits) {
        const ScanRare r = *v.rare;
        unsigned long long h = atomicAdd(r.hit_count, 1ull);
        if ((long long)h < r.hit_cap)
            r.hits[h] = wd_hit{v.tile, v.t_first + tl, rel0 + e, dist};
    }
}

__device__ inline void q_wave_sync()
{
    // LDS traffic of one wave is processed in order; this only stops the compiler from
    // moving queue reads above the pushes of other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Drain a wave's queue: every entry has seen cycles [0, j0); rounds of 2, 4, 8 ... more
// cycles, one lane per survivor, re-compacting after each round.  Whoever is still alive
// after kFinishFrom cycles is almost certainly a true duplicate (a random neighbour gets
// there with probability 4^-8): those are finished one entry at a time with one LANE PER
// CYCLE - a single ballot + popcount adds up 64 cycles' mismatches, so a duplicate costs
// ceil((L - 8) / 64) round trips instead of (L - 8) / 8.  qn is reset to 0.
constexpr int kFinishFrom = 8;
template <bool STRIDED, int MAXB = 8>
__device__ inline void q_drain(const QEnv &v, uint2 *q_a, uint2 *q_b, int &qn, int j0, int lane)
{
    q_wave_sync();
    const int L = v.L, k = v.k;
    int j = min(j0, L);
    int nb = 2;
    uint2 *qa = q_a, *qb = q_b;
    int n = qn;
    while (n > 0 && j < L) {
        if (j >= kFinishFrom) {
            for (int i = 0; i < n; i++) {
                const uint2 ent = qa[i];                       // same address in every lane
                const uint32_t idx = ent.x;
                const int tl = (int)(ent.y >> 24);
                const uint32_t c = (uint32_t)v.s_centre[tl];
                int mm = (int)((ent.y >> 16) & 0xFFu);
                for (int j0 = j; j0 < L && mm <= k; j0 += kWave) {
                    const int jj = j0 + lane;
                    gbytes p = q_plane<STRIDED>(v, min(jj, L - 1));
                    const bool diff = code_of(ldb(p, idx)) != code_of(ldb(p, c));
                    mm += __popcll(__ballot(jj < L && diff));
                }
                if (mm <= k && lane == 0)
                    q_record_dup(v, tl, (int)(ent.y & 0xFFFFu), mm);
            }
            break;
        }
        const bool last = j + nb >= L;
        int n2 = 0;
        for (int p0 = 0; p0 < n; p0 += kWave) {
            const bool have = p0 + lane < n;
            uint2 ent = make_uint2(0u, 0u);
            if (have)
                ent = qa[p0 + lane];
            const uint32_t idx = ent.x;
            const int tl = (int)(ent.y >> 24);
            const int e = (int)(ent.y & 0xFFFFu);
            int mm = (int)((ent.y >> 16) & 0xFFu);
            if (have) {
                const uint32_t c = (uint32_t)v.s_centre[tl];
                uint32_t w[MAXB], cb[MAXB];
#pragma unroll
                for (int q = 0; q < MAXB; q++) {
                    if (q < nb) {
                        gbytes p = q_plane<STRIDED>(v, min(j + q, L - 1));
                        w[q] = ldb(p, idx);
                        cb[q] = ldb(p, c);
                    }
                }
#pragma unroll
                for (int q = 0; q < MAXB; q++) {
                    if (q < nb && j + q < L)
                        mm += code_of(w[q]) != code_of(cb[q]) ? 1 : 0;
                }
            }
            const bool alive = have && mm <= k;
            if (last) {
                if (alive)
                    q_record_dup(v, tl, e, mm);
            } else {
                const uint64_t m = __ballot(alive);
                if (alive) {
                    const int pos = n2 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                             __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    qb[pos] = make_uint2(idx, (ent.y & 0xFF00FFFFu) | ((uint32_t)min(mm, 255) << 16));
                }
                n2 += __popcll(m);
            }
        }
        q_wave_sync();
        uint2 *t = qa; qa = qb; qb = t;
        n = last ? 0 : n2;
        j += nb;
        nb = min(MAXB, nb * 2);
    }
    qn = 0;
}

// Push the lanes flagged `alive` (ballot m) behind the qn entries already queued.
__device__ inline void q_push(uint2 *q_a, int qn, uint64_t m, bool alive, uint32_t idx, uint32_t tag)
{
    if (alive) {
        const int pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                 __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        q_a[pos] = make_uint2(idx, tag);
    }
}

__host__ __device__ inline int scan_q_lds_dwords(int levels, int tpb)
{
    int n = (1 + 5 * levels) + tpb + tpb * (levels + 1) + tpb * levels + tpb + kMaxPasses * tpb + 4;
    n = (n + 1) & ~1;               // queue entries are 8-byte pairs
    return n + kWaves * 2 * kQCap * 2;
}

template <bool STRIDED, int B1>
__global__ __launch_bounds__(kBlock, 6) void k_scan_q(ScanArgs a)
{
    extern __shared__ uint32_t smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int levels = a.levels;
    const int L = a.L;
    const int k = a.k;
    const int tpb = a.tpb;
    const int ncnt = 1 + 5 * levels;
    const int chunks = (a.T + tpb - 1) / tpb;
    const int tile = blockIdx.x / chunks;
    const int chunk = blockIdx.x - tile * chunks;
    const int t_first = chunk * tpb;
    const int n_t = min(tpb, a.T - t_first);

    uint32_t *s_cnt = smem;
    int32_t *s_centre = (int32_t *)(s_cnt + ncnt);
    int32_t *s_off = s_centre + tpb;
    uint32_t *s_d = (uint32_t *)(s_off + tpb * (levels + 1));
    uint32_t *s_valid = s_d + tpb * levels;
    uint32_t *s_items = s_valid + tpb;
    uint32_t *s_misc = s_items + kMaxPasses * tpb;
    const int q_base = (int)((((s_misc + 4) - smem) + 1) & ~1);
    uint2 *q_a = (uint2 *)(smem + q_base) + wave * 2 * kQCap;
    uint2 *q_b = q_a + kQCap;

    gbytes filt = as_global(a.filter[tile]);
    const uint8_t *const *ptab = STRIDED ? nullptr : a.planes + (size_t)tile * L;
    gbytes base0 = STRIDED ? as_global(a.planes[tile]) : nullptr;
    const int64_t stride = a.stride;
    auto plane_ptr = [&](int j) -> gbytes {
        return STRIDED ? base0 + (int64_t)j * stride : as_global(ptab[j]);
    };

    // ---------------- phase 0: metadata, filter bytes, work items ----------------
    for (int i = threadIdx.x; i < ncnt; i += kBlock)
        s_cnt[i] = 0;
    for (int i = threadIdx.x; i < n_t * levels; i += kBlock)
        s_d[i] = 0;
    for (int i = threadIdx.x; i < n_t; i += kBlock)
        s_centre[i] = a.centre[t_first + i];
    for (int i = threadIdx.x; i < n_t * (levels + 1); i += kBlock)
        s_off[i] = a.lvl_off[(size_t)t_first * (levels + 1) + i];
    __syncthreads();
    if (wave == 0) {
        const bool in = lane < n_t;
        const int32_t *o = s_off + (in ? lane : 0) * (levels + 1);
        const uint32_t c = in ? (uint32_t)s_centre[lane] : 0u;
        const uint32_t fb = in ? (uint32_t)filt[c] : 0u;
        bool valid = in && (fb & 1u);                          // :236-237
        if (a.check_empty && valid) {                           // :249
            bool empty = false;
            for (int l = 0; l < levels; l++)
                empty = empty || (o[l + 1] <= o[l]);
            if (empty) {
                atomicOr(a.rare->status, kStatusEmptyLevel);
                valid = false;
            }
        }
        const int K = in ? o[levels] - o[0] : 0;
        const int np = valid ? (K + kPass - 1) / kPass : 0;
        int incl = np;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d)
                incl += v;
        }
        if (in)
            s_valid[lane] = valid ? 1u : 0u;
        for (int p = 0; p < np; p++)
            s_items[incl - np + p] = ((uint32_t)lane << 8) | (uint32_t)p;
        if (lane == kWave - 1)
            s_misc[0] = (uint32_t)incl;
    }
    __syncthreads();

    QEnv env;
    env.s_centre = s_centre;
    env.s_off = s_off;
    env.s_d = s_d;
    env.base0 = base0;
    env.ptab = ptab;
    env.stride = stride;
    env.rare = a.rare;
    env.levels = levels;
    env.L = L;
    env.k = k;
    env.tile = tile;
    env.t_first = t_first;
    env.log_hits = a.log_hits;
    int qn = 0;

    // ---------------- phase 1: pipelined first round over this wave's items -----------------
    const int nitems = (int)s_misc[0];
    const int n_my = nitems > wave ? (nitems - wave + kWaves - 1) / kWaves : 0;
    if (n_my > 0) {
        // plane pointers of the first round (clamped: cycles >= L are masked out below)
        gbytes pp[B1];
#pragma unroll
        for (int q = 0; q < B1; q++)
            pp[q] = plane_ptr(min(q, max(L - 1, 0)));

        // item registers by age: 0 = indices being loaded ... 3 = being consumed
        uint32_t it0 = 0, it1 = 0, it2 = 0, it3 = 0;
        uint32_t i0_0 = 0, i1_0 = 0, i0_1 = 0, i1_1 = 0, i0_2 = 0, i1_2 = 0, i0_3 = 0, i1_3 = 0;
        uint32_t w0_2[B1], w1_2[B1], w0_3[B1], w1_3[B1];
#pragma unroll
        for (int q = 0; q < B1; q++)
            w0_2[q] = w1_2[q] = w0_3[q] = w1_3[q] = 0;

        for (int s = 0; s < n_my + 3; s++) {
            // rotate ages
            it3 = it2; i0_3 = i0_2; i1_3 = i1_2;
#pragma unroll
            for (int q = 0; q < B1; q++) {
                w0_3[q] = w0_2[q];
                w1_3[q] = w1_2[q];
            }
            it2 = it1; i0_2 = i0_1; i1_2 = i1_1;
            it1 = it0; i0_1 = i0_0; i1_1 = i1_0;

            // stage B: plane bytes of item s-2 (its indices were requested two steps ago)
            if (L > 0) {
#pragma unroll
                for (int q = 0; q < B1; q++) {
                    w0_2[q] = ldb(pp[q], i0_2);
                    w1_2[q] = ldb(pp[q], i1_2);
                }
            }
            // stage A: neighbour indices of item s (clamped to the wave's last item)
            {
                it0 = s_items[wave + kWaves * min(s, n_my - 1)];
                const int tl = (int)(it0 >> 8);
                const int base = (int)(it0 & 0xFFu) * kPass;
                const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane(s_centre[tl]);
                const int off0 = __builtin_amdgcn_readfirstlane(s_off[tl * (levels + 1)]);
                const int K = __builtin_amdgcn_readfirstlane(s_off[tl * (levels + 1) + levels]) - off0;
                const int e0 = base + lane, e1 = e0 + kWave;
                // idle lanes (and always lane 63 of the second slot) shadow the centre well
                // (loads are unconditional - clamped slot, then select - so that the number of
                // loads per step is static and the compiler can keep counted vmcnt waits)
                const uint32_t r0 = (uint32_t)a.nbr[off0 + min(e0, K - 1)];
                const uint32_t r1 = (uint32_t)a.nbr[off0 + min(e1, K - 1)];
                i0_0 = e0 < K ? r0 : c;
                i1_0 = (lane < kWave - 1 && e1 < K) ? r1 : c;
            }
            // stage C: consume item s-3
            if (s >= 3) {
                const int tl = (int)(it3 >> 8);
                const int base = (int)(it3 & 0xFFu) * kPass;
                const int off0 = __builtin_amdgcn_readfirstlane(s_off[tl * (levels + 1)]);
                const int K = __builtin_amdgcn_readfirstlane(s_off[tl * (levels + 1) + levels]) - off0;
                const int e0 = base + lane, e1 = e0 + kWave;
                const bool a0 = e0 < K, a1 = lane < kWave - 1 && e1 < K;
                int mm0 = 0, mm1 = 0;
#pragma unroll
                for (int q = 0; q < B1; q++) {
                    if (q < L) {
                        const uint32_t cc = code_of((uint32_t)__builtin_amdgcn_readlane((int)w1_3[q], kWave - 1));
                        mm0 += code_of(w0_3[q]) != cc ? 1 : 0;
                        mm1 += code_of(w1_3[q]) != cc ? 1 : 0;
                    }
                }
                const bool al0 = a0 && mm0 <= k, al1 = a1 && mm1 <= k;
                const uint64_t m0 = __ballot(al0), m1 = __ballot(al1);
                if (m0 | m1) {
                    if (B1 >= L) {
                        if (al0)
                            q_record_dup(env, tl, e0, mm0);
                        if (al1)
                            q_record_dup(env, tl, e1, mm1);
                    } else {
                        const int n0 = __popcll(m0), n1 = __popcll(m1);
                        if (qn + n0 + n1 > kQCap)
                            q_drain<STRIDED>(env, q_a, q_b, qn, B1, lane);
                        q_push(q_a, qn, m0, al0, i0_3,
                               ((uint32_t)tl << 24) | ((uint32_t)min(mm0, 255) << 16) | (uint32_t)e0);
                        q_push(q_a, qn + n0, m1, al1, i1_3,
                               ((uint32_t)tl << 24) | ((uint32_t)min(mm1, 255) << 16) | (uint32_t)e1);
                        qn += n0 + n1;
                    }
                }
            }
        }
        if (qn > 0)
            q_drain<STRIDED>(env, q_a, q_b, qn, B1, lane);
    }
    __syncthreads();

    // ---------------- phase 3: tallies (count_well_duplicates.py:80-95 as histograms) -------
    uint32_t acc_wells = 0, acc_dups = 0, acc_hit = 0, acc_first = 0, acc_last = 0, acc_valid = 0;
    for (int tl = wave; tl < n_t; tl += kWaves) {
        const bool valid = __builtin_amdgcn_readfirstlane(s_valid[tl]) != 0;
        uint32_t *opt = a.out_per_target
                            ? a.out_per_target + ((size_t)tile * a.T + t_first + tl) * levels
                            : nullptr;
        if (!valid) {
            if (opt && lane < levels)
                opt[lane] = WD_INVALID_TARGET;
            continue;
        }
        uint32_t my_d = 0, my_w = 0;
        if (lane < levels) {
            my_d = s_d[tl * levels + lane];
            my_w = (uint32_t)(s_off[tl * (levels + 1) + lane + 1] - s_off[tl * (levels + 1) + lane]);
        }
        const uint64_t hm = __ballot(lane < levels && my_d > 0);
        acc_valid += 1;
        acc_wells += my_w;
        acc_dups += my_d;
        acc_hit += my_d ? 1u : 0u;
        if (hm) {
            acc_first += (lane == __ffsll((long long)hm) - 1) ? 1u : 0u;
            acc_last += (lane == 63 - __clzll((long long)hm)) ? 1u : 0u;
        }
        if (opt && lane < levels)
            opt[lane] = my_d;
    }
    if (lane < levels) {
        if (acc_wells) atomicAdd(&s_cnt[1 + lane], acc_wells);
        if (acc_dups) atomicAdd(&s_cnt[1 + levels + lane], acc_dups);
        if (acc_hit) atomicAdd(&s_cnt[1 + 2 * levels + lane], acc_hit);
        if (acc_first) atomicAdd(&s_cnt[1 + 3 * levels + lane], acc_first);
        if (acc_last) atomicAdd(&s_cnt[1 + 4 * levels + lane], acc_last);
    }
    if (lane == 0 && acc_valid)
        atomicAdd(&s_cnt[0], acc_valid);
    __syncthreads();
    for (int i = threadIdx.x; i < ncnt; i += kBlock) {
        const uint32_t v = s_cnt[i];
        if (v)
            atomicAdd(&a.out_tile[(size_t)tile * ncnt + i], (unsigned long long)v);
    }
}

// -------------------------------------------------------------------------------------
// Dense kernel (Hamming family): one LANE per target - for "every well is a centre" scans
// -------------------------------------------------------------------------------------
// With millions of targets of ~36 neighbours each (BASELINE config 5) a wave per target wastes
// most lanes and pays hundreds of instructions per target.  Here consecutive lanes own
// consecutive centres: for the q-th neighbour the 64 lanes read 64 mostly consecutive wells,
// so plane loads coalesce and the planes' first cycles are effectively streamed once through
// L2, while each lane walks its own index list.  Round 1 reads 2 cycles of every neighbour;
// survivors go through the same per-wave LDS queue and dense drain rounds as k_scan_q
// (walking them lane by lane instead costs ~350 k cycles of dependent loads per wave).
// Neighbour lists regrouped for the lane-per-target kernel: for each group of 64 consecutive
// targets, nbr_t[gbase + q*64 + lane] is the q-th neighbour of target 64*group + lane (the
// target's own centre where q >= K), so that one wave-load reads 256 contiguous bytes.
__global__ __launch_bounds__(kWave) void k_transpose_nbr(const int32_t *centre, const int32_t *lvl_off,
                                                         const int32_t *nbr, const long long *gbase,
                                                         int32_t *nbr_t, int T, int levels)
{
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    const int t = g * kWave + lane;
    const bool in = t < T;
    const int off0 = in ? lvl_off[(size_t)t * (levels + 1)] : 0;
    const int K = in ? lvl_off[(size_t)t * (levels + 1) + levels] - off0 : 0;
    const int c = in ? centre[t] : 0;
    const long long b0 = gbase[g], b1 = gbase[g + 1];
    const int kmax = (int)((b1 - b0) / kWave);
    for (int q = 0; q < kmax; q++)
        nbr_t[b0 + (long long)q * kWave + lane] = q < K ? nbr[off0 + q] : c;
}

constexpr int kDenseMaxK = 16384;   // slot index must fit the queue tag's 16 bits
constexpr int kDenseQCap = 256;

__host__ __device__ inline int scan_dense_lds_dwords(int levels)
{
    int n = (1 + 5 * levels) + kBlock + kBlock * (levels + 1) + kBlock * levels;
    n = (n + 1) & ~1;
    return n + kWaves * 2 * kDenseQCap * 2;
}

template <bool STRIDED>
__global__ __launch_bounds__(kBlock) void k_scan_dense(ScanArgs a)
{
    extern __shared__ uint32_t smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int levels = a.levels;
    const int L = a.L;
    const int k = a.k;
    const int ncnt = 1 + 5 * levels;
    const int bpt = (a.T + kBlock - 1) / kBlock;
    const int tile = blockIdx.x / bpt;
    const int t_first = (blockIdx.x - tile * bpt) * kBlock;
    const int tl = threadIdx.x;
    const int t = t_first + tl;
    const bool in = t < a.T;

    uint32_t *s_cnt = smem;
    int32_t *s_centre = (int32_t *)(s_cnt + ncnt);
    int32_t *s_off = s_centre + kBlock;
    uint32_t *s_d = (uint32_t *)(s_off + kBlock * (levels + 1));
    const int q_base = (int)((((s_d + kBlock * levels) - smem) + 1) & ~1);
    uint2 *q_a = (uint2 *)(smem + q_base) + wave * 2 * kDenseQCap;
    uint2 *q_b = q_a + kDenseQCap;

    gbytes filt = as_global(a.filter[tile]);
    const uint8_t *const *ptab = STRIDED ? nullptr : a.planes + (size_t)tile * L;
    gbytes base0 = STRIDED ? as_global(a.planes[tile]) : nullptr;
    const int64_t stride = a.stride;
    auto plane_ptr = [&](int j) -> gbytes {
        return STRIDED ? base0 + (int64_t)j * stride : as_global(ptab[j]);
    };

    // ---- phase 0: this lane's target ----
    for (int i = threadIdx.x; i < ncnt; i += kBlock)
        s_cnt[i] = 0;
    for (int l = 0; l < levels; l++)
        s_d[tl * levels + l] = 0;
    const int32_t *off = a.lvl_off + (size_t)(in ? t : 0) * (levels + 1);
    const uint32_t c = in ? (uint32_t)a.centre[t] : 0u;
    s_centre[tl] = (int32_t)c;
    for (int l = 0; l <= levels; l++)
        s_off[tl * (levels + 1) + l] = off[l];
    bool valid = in && ((uint32_t)filt[c] & 1u);                       // :236-237
    if (a.check_empty && valid) {                                      // :249
        bool empty = false;
        for (int l = 0; l < levels; l++)
            empty = empty || (off[l + 1] <= off[l]);
        if (empty) {
            atomicOr(a.rare->status, kStatusEmptyLevel);
            valid = false;
        }
    }
    const int off0 = off[0];
    const int K = valid ? off[levels] - off0 : 0;
    __syncthreads();

    QEnv env;
    env.s_centre = s_centre;
    env.s_off = s_off;
    env.s_d = s_d;
    env.base0 = base0;
    env.ptab = ptab;
    env.stride = stride;
    env.rare = a.rare;
    env.levels = levels;
    env.L = L;
    env.k = k;
    env.tile = tile;
    env.t_first = t_first;
    env.log_hits = a.log_hits;
    int qn = 0;

    // ---- phase 1: cycles 0 and 1 of every neighbour, one lane per target ----
    gbytes p0 = plane_ptr(0);
    gbytes p1 = plane_ptr(min(1, L - 1));
    const uint32_t cc0 = code_of(p0[c]);
    const uint32_t cc1 = code_of(p1[c]);
    int kmax = K;                                     // wave-wide trip count
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        kmax = max(kmax, __shfl_xor(kmax, d));
    // this wave's group of 64 targets in the transposed neighbour table
    const long long gb = a.gbase[(t_first >> 6) + wave];
    const int gk = (int)((a.gbase[(t_first >> 6) + wave + 1] - gb) >> 6);     // >= kmax
    const int32_t *nt = a.nbr_t + gb + lane;
#ifndef WD_DENSE_U
#define WD_DENSE_U 4
#endif
#ifndef WD_DENSE_MAXB
#define WD_DENSE_MAXB 8
#endif
    constexpr int U = WD_DENSE_U;                      // neighbours per step
    uint32_t idx[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; u++)
        idx[u] = (uint32_t)nt[(size_t)min(u, gk - 1) * kWave];           // own centre where q >= K
    for (int q0 = 0; q0 < kmax; q0 += U) {
        uint32_t w0[U], w1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {                  // next step's indices ride behind
            nxt[u] = (uint32_t)nt[(size_t)min(q0 + U + u, gk - 1) * kWave];
            w0[u] = p0[idx[u]];
            w1[u] = p1[idx[u]];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int mm = (code_of(w0[u]) != cc0 ? 1 : 0) + ((L > 1 && code_of(w1[u]) != cc1) ? 1 : 0);
            const bool alive = q0 + u < K && mm <= k;
            const uint64_t m = __ballot(alive);
            if (m) {
                if (L <= 2) {
                    if (alive)
                        q_record_dup(env, tl, q0 + u, mm);
                } else {
                    const int n = __popcll(m);
                    if (qn + n > kDenseQCap)
                        q_drain<STRIDED, WD_DENSE_MAXB>(env, q_a, q_b, qn, 2, lane);
                    q_push(q_a, qn, m, alive, idx[u],
                           ((uint32_t)tl << 24) | ((uint32_t)mm << 16) | (uint32_t)(q0 + u));
                    qn += n;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            idx[u] = nxt[u];
    }
    // ---- phase 2: survivors, dense lanes ----
    if (qn > 0)
        q_drain<STRIDED, WD_DENSE_MAXB>(env, q_a, q_b, qn, 2, lane);
    __syncthreads();

    // ---- phase 3: tallies, one lane per target ----
    uint32_t hm = 0;
    for (int l = 0; l < levels; l++)
        hm |= (valid && s_d[tl * levels + l]) ? (1u << l) : 0u;
    const uint64_t vmask = __ballot(valid);
    if (lane == 0 && vmask)
        atomicAdd(&s_cnt[0], (uint32_t)__popcll(vmask));
    for (int l = 0; l < levels; l++) {
        int w = valid ? off[l + 1] - off[l] : 0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1)
            w += __shfl_xor(w, d);
        if (lane == 0 && w)
            atomicAdd(&s_cnt[1 + l], (uint32_t)w);
    }
    if (hm) {
        for (int l = 0; l < levels; l++) {
            const uint32_t d = s_d[tl * levels + l];
            if (d) {
                atomicAdd(&s_cnt[1 + levels + l], d);
                atomicAdd(&s_cnt[1 + 2 * levels + l], 1u);
            }
        }
        atomicAdd(&s_cnt[1 + 3 * levels + (__ffs((int)hm) - 1)], 1u);
        atomicAdd(&s_cnt[1 + 4 * levels + (31 - __clz((int)hm))], 1u);
    }
    if (a.out_per_target && in) {
        uint32_t *opt = a.out_per_target + ((size_t)tile * a.T + t) * levels;
        for (int l = 0; l < levels; l++)
            opt[l] = valid ? s_d[tl * levels + l] : WD_INVALID_TARGET;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncnt; i += kBlock) {
        const uint32_t v = s_cnt[i];
        if (v)
            atomicAdd(&a.out_tile[(size_t)tile * ncnt + i], (unsigned long long)v);
    }
}

// -------------------------------------------------------------------------------------
// Generic Levenshtein kernel: any threshold (band half-width H = k/2 > 8), correct not fast
// -------------------------------------------------------------------------------------
// One wave per (tile, target), one lane per neighbour slot, 64 slots per pass.  The well's L
// codes and the banded DP row live in LDS ([.][lane] layout, conflict-free); no early exit.
// Same recurrence as LevState (row-wise, band |j - i| <= H, saturating at k + 1).
__host__ __device__ inline size_t lev_generic_lds_bytes(int L, int H)
{
    return (size_t)(2 * H + 1) * kWave * sizeof(uint16_t) + (size_t)L * kWave + (size_t)L + 8;
}

template <bool STRIDED>
__global__ __launch_bounds__(kWave) void k_scan_lev_generic(ScanArgs a, int H)
{
    extern __shared__ uint32_t smem[];
    const int lane = threadIdx.x;
    const int levels = a.levels;
    const int L = a.L;
    const int k = a.k;
    const int cap = k + 1;
    const int W = 2 * H + 1;
    const int ncnt = 1 + 5 * levels;
    const int tile = blockIdx.x / a.T;
    const int t = blockIdx.x - tile * a.T;
    uint16_t *row = (uint16_t *)smem;
    uint8_t *wcode = (uint8_t *)(row + (size_t)W * kWave);
    uint8_t *ccode = wcode + (size_t)L * kWave;

    gbytes filt = as_global(a.filter[tile]);
    const uint8_t *const *ptab = STRIDED ? nullptr : a.planes + (size_t)tile * L;
    gbytes base0 = STRIDED ? as_global(a.planes[tile]) : nullptr;
    const int64_t stride = a.stride;
    auto plane_ptr = [&](int j) -> gbytes {
        return STRIDED ? base0 + (int64_t)j * stride : as_global(ptab[j]);
    };

    const int32_t *off = a.lvl_off + (size_t)t * (levels + 1);
    const uint32_t c = (uint32_t)a.centre[t];
    const int off0 = off[0];
    const int K = off[levels] - off0;
    uint32_t *opt = a.out_per_target ? a.out_per_target + ((size_t)tile * a.T + t) * levels : nullptr;
    if (!(__builtin_amdgcn_readfirstlane((uint32_t)filt[c]) & 1u)) {        // :236-237
        if (opt && lane < levels)
            opt[lane] = WD_INVALID_TARGET;
        return;
    }
    int my_lo = 0, my_hi = 0;
    if (lane < levels) {
        my_lo = off[lane] - off0;
        my_hi = off[lane + 1] - off0;
    }
    if (a.check_empty && __ballot(lane < levels && my_hi <= my_lo)) {         // :249
        if (lane == 0)
            atomicOr(a.rare->status, kStatusEmptyLevel);
        return;
    }
    for (int j = 0; j < L; j++) {
        const uint32_t b = plane_ptr(j)[c];
        if (lane == 0)
            ccode[j] = (uint8_t)code_of(b);
    }
    __syncthreads();
    uint32_t my_d = 0;
    for (int base = 0; base < K; base += kWave) {
        const int e = base + lane;
        const bool act = e < K;
        const uint32_t idx = act ? (uint32_t)a.nbr[off0 + e] : c;
        for (int j = 0; j < L; j++)
            wcode[(size_t)j * kWave + lane] = (uint8_t)code_of(plane_ptr(j)[idx]);
        for (int d = 0; d < W; d++)
            row[(size_t)d * kWave + lane] = (uint16_t)(d >= H ? min(d - H, cap) : cap);
        for (int i = 1; i <= L; i++) {
            const uint32_t ci = ccode[i - 1];
            int left = cap;
            for (int d = 0; d < W; d++) {
                const int j = i + d - H;
                int v;
                if (j < 0 || j > L) {
                    v = cap;
                } else if (j == 0) {
                    v = min(i, cap);
                } else {
                    v = row[(size_t)d * kWave + lane] + (ci != wcode[(size_t)(j - 1) * kWave + lane] ? 1 : 0);
                    if (d + 1 < W)
                        v = min(v, row[(size_t)(d + 1) * kWave + lane] + 1);
                    v = min(min(v, left + 1), cap);
                }
                row[(size_t)d * kWave + lane] = (uint16_t)v;
                left = v;
            }
        }
        const int dist = row[(size_t)H * kWave + lane];
        const bool dup = act && dist <= k;
        const uint64_t m = __ballot(dup);
        if (m) {
            if (lane < levels)
                my_d += __popcll(m & range_mask(my_lo - base, my_hi - base));
            if (a.log_hits && dup) {
                const ScanRare r = *a.rare;
                unsigned long long h = atomicAdd(r.hit_count, 1ull);
                if ((long long)h < r.hit_cap)
                    r.hits[h] = wd_hit{tile, t, off0 + e, dist};
            }
        }
    }
    const uint64_t hm = __ballot(lane < levels && my_d > 0);
    unsigned long long *ot = a.out_tile + (size_t)tile * ncnt;
    if (lane < levels) {
        atomicAdd(&ot[1 + lane], (unsigned long long)(my_hi - my_lo));
        if (my_d) {
            atomicAdd(&ot[1 + levels + lane], (unsigned long long)my_d);
            atomicAdd(&ot[1 + 2 * levels + lane], 1ull);
        }
        if (opt)
            opt[lane] = my_d;
    }
    if (lane == 0) {
        atomicAdd(&ot[0], 1ull);
        if (hm) {
            atomicAdd(&ot[1 + 3 * levels + (__ffsll((long long)hm) - 1)], 1ull);
            atomicAdd(&ot[1 + 4 * levels + (63 - __clzll((long long)hm))], 1ull);
        }
    }
}

// -------------------------------------------------------------------------------------
// Neighbour-index generator (the producer of the targets): prepare_cluster_indexes.py on device
// -------------------------------------------------------------------------------------
// get_indexes() (prepare_cluster_indexes.py:38-78) scans records max(0, c-20000) .. c+20001
// and bins each by pixel distance to the centre: ring r holds md[r] < dist <= md[r+1], compared
// here on exact integer squares.  One block per centre; two launches: count per (centre,
// ring), host prefix sum, then fill - matches are collected in LDS and ranked by (ring, index)
// so every ring comes out in ascending index order, as the reference's scan emits it.
constexpr int kGenWindow = 20000;       // MAX_SEARCH_AREA (:43)
constexpr int kGenMaxMatches = 2048;

struct GenArgs {
    const int32_t *x, *y;
    const int32_t *centres;   // nullable: centre i is well i
    int64_t n;
    int n_centres;
    int levels;
    int md2[kMaxLevels + 1];  // squared ring boundaries
    int32_t *counts;          // [n_centres][levels]            (pass 1 output)
    const int32_t *lvl_off;   // [n_centres][levels+1] absolute (pass 2 input)
    int32_t *nbr;             // pass 2 output
    uint32_t *status;
};

template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_gen_rings(GenArgs a)
{
    __shared__ uint32_t s_cnt[kMaxLevels];
    __shared__ uint32_t s_n;
    __shared__ int32_t s_idx[FILL ? kGenMaxMatches : 1];
    __shared__ uint8_t s_lev[FILL ? kGenMaxMatches : 1];
    const int t = blockIdx.x;
    const int64_t c = a.centres ? (int64_t)a.centres[t] : (int64_t)t;
    const int levels = a.levels;
    if (threadIdx.x < kMaxLevels)
        s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0)
        s_n = 0;
    __syncthreads();
    const int cx = a.x[c], cy = a.y[c];
    const int64_t lo = c > kGenWindow ? c - kGenWindow : 0;
    const int64_t hi = min(a.n, c + kGenWindow + 2);        // the record at c+20001 is examined (:66)
    const int far2 = a.md2[levels];
    const int near2 = a.md2[0];
    for (int64_t j = lo + threadIdx.x; j < hi; j += kBlock) {
        const int64_t dx = (int64_t)a.x[j] - cx, dy = (int64_t)a.y[j] - cy;
        const int64_t d2l = dx * dx + dy * dy;
        if (d2l > far2 || d2l <= near2)
            continue;
        const int d2 = (int)d2l;
        int lev = 0;
        for (int r = 1; r < levels; r++)
            lev += d2 > a.md2[r] ? 1 : 0;
        if (FILL) {
            const uint32_t pos = atomicAdd(&s_n, 1u);
            if (pos < kGenMaxMatches) {
                s_idx[pos] = (int32_t)j;
                s_lev[pos] = (uint8_t)lev;
            }
        } else {
            atomicAdd(&s_cnt[lev], 1u);
        }
    }
    __syncthreads();
    if (!FILL) {
        if (threadIdx.x < levels) {
            a.counts[(size_t)t * levels + threadIdx.x] = (int32_t)s_cnt[threadIdx.x];
            if (s_cnt[threadIdx.x] == 0)
                atomicOr(a.status, 2u);                       // :70-76 RuntimeError
        }
        return;
    }
    const uint32_t m = s_n;
    if (m > kGenMaxMatches) {
        if (threadIdx.x == 0)
            atomicOr(a.status, 4u);
        return;
    }
    const int32_t *off = a.lvl_off + (size_t)t * (levels + 1);
    for (uint32_t i = threadIdx.x; i < m; i += kBlock) {
        const int32_t me = s_idx[i];
        const int lev = s_lev[i];
        int rank = 0;                                         // matches of my ring with a smaller index
        for (uint32_t q = 0; q < m; q++)
            rank += (s_lev[q] == lev && s_idx[q] < me) ? 1 : 0;
        a.nbr[off[lev] + rank] = me;
    }
}

// -------------------------------------------------------------------------------------
// Ingest helpers
// -------------------------------------------------------------------------------------
// out[w * L + c] = plane_c[idx[w]]: the bytes of a few wells over all scanned cycles (what the
// stderr duplicate log prints), so whole planes never have to exist in host memory.
__global__ __launch_bounds__(kBlock) void k_gather_wells(const uint8_t *const *planes, int L,
                                                         const int32_t *idx, long long n, uint8_t *out)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n * L)
        return;
    const long long w = i / L;
    const int c = (int)(i - w * L);
    out[i] = as_global(planes[c])[(uint32_t)idx[w]];
}

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

int g_create_status = WD_OK;

}  // namespace

// -------------------------------------------------------------------------------------
// Context
// -------------------------------------------------------------------------------------
struct wd_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;

    // options
    int early_exit = 1;
    int tpb = 64;
    int batch_first = 4;
    int batch_next = 4;
    int queue_kernel = 1;      // equality / Hamming with early exit: use k_scan_q
    int queue_first = 0;       // cycles of its first round; 0 = choose from k
    int dense_kernel = -1;     // lane-per-target kernel: -1 = when the targets look dense
    int profile = 0;

    // targets (device)
    int T = 0, levels = 0;
    int64_t P = 0;
    int32_t *d_centre = nullptr, *d_lvl_off = nullptr, *d_nbr = nullptr;
    int64_t idx_min = 0, idx_max = -1;
    int64_t k_max = 0;         // most neighbour slots of any target
    std::vector<long long> h_gbase;   // per 64-target group: start in the transposed table
    int32_t *d_nbr_t = nullptr;       // built on first use of the dense kernel
    long long *d_gbase = nullptr;
    bool has_targets = false;
    bool has_empty_level = false;

    // per-call tables
    std::vector<const uint8_t *> h_tbl;      // last uploaded pointer table (planes then filter)
    const uint8_t **d_tbl = nullptr;
    size_t d_tbl_cap = 0;
    uint32_t *d_status = nullptr;
    uint32_t *h_status = nullptr;            // pinned
    ScanRare *d_rare = nullptr;
    ScanRare h_rare = {nullptr, nullptr, nullptr, 0};

    // sync-call scratch
    unsigned long long *d_out_tile = nullptr;
    size_t d_out_tile_cap = 0;
    uint32_t *d_out_pt = nullptr;
    size_t d_out_pt_cap = 0;

    // hit log
    wd_hit *d_hits = nullptr;
    unsigned long long *d_hit_count = nullptr;
    int64_t hit_cap = 0;

    // profile
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
    double prof_ms = 0.0;
    int64_t prof_launches = 0;

    // comm
    void *comm = nullptr;

    // ingest: pinned staging buffers + copy streams, one per concurrently loading thread
    struct IngestSlot {
        uint8_t *pinned = nullptr;
        size_t cap = 0;
        hipStream_t stream = nullptr;
        bool busy = false;
    };
    std::mutex ingest_mu;
    std::vector<IngestSlot *> ingest_slots;
};

namespace {

int fail(wd_ctx *ctx, int code, const std::string &msg)
{
    if (ctx)
        ctx->err = msg;
    return code;
}

#define WD_HIP(ctx, call)                                                                 \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail((ctx), e_ == hipErrorOutOfMemory ? WD_ERR_NOMEM : WD_ERR_HIP,     \
                        std::string(#call) + ": " + hipGetErrorString(e_));               \
    } while (0)

int bind_device(wd_ctx *ctx)
{
    WD_HIP(ctx, hipSetDevice(ctx->device));
    return WD_OK;
}

template <class T>
int grow(wd_ctx *ctx, T *&ptr, size_t &cap, size_t need)
{
    if (need <= cap)
        return WD_OK;
    if (ptr) {
        WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        WD_HIP(ctx, hipFree(ptr));
        ptr = nullptr;
        cap = 0;
    }
    WD_HIP(ctx, hipMalloc((void **)&ptr, need * sizeof(T)));
    cap = need;
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

// Batch shapes (cycles read unconditionally, then per conditional batch).  The Hamming family
// is instantiated for a few shapes so they can be tuned; the banded edit-distance family
// needs ~2x the cycles before a random neighbour dies, so it uses one deeper shape.
constexpr int kHamShapes[][2] = {{2, 4}, {3, 4}, {4, 4}, {4, 8}, {8, 8}};
constexpr int kLevB1 = 8, kLevB2 = 8;

template <bool STRIDED>
void launch_ham(wd_ctx *ctx, const ScanArgs &a, dim3 grid)
{
    const int b1 = ctx->batch_first, b2 = ctx->batch_next;
#define WD_CASE(B1, B2)                                                                   \
    if (b1 == B1 && b2 == B2) {                                                           \
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
    if (strided)
        hipLaunchKernelGGL((k_scan<LevState<H>, true, kLevB1, kLevB2>), grid, dim3(kBlock), 0,
                           ctx->stream, a);
    else
        hipLaunchKernelGGL((k_scan<LevState<H>, false, kLevB1, kLevB2>), grid, dim3(kBlock), 0,
                           ctx->stream, a);
}

template <bool STRIDED>
int launch_queue(wd_ctx *ctx, const ScanArgs &a, dim3 grid)
{
    const size_t lds = (size_t)scan_q_lds_dwords(a.levels, a.tpb) * sizeof(uint32_t);
    // A random neighbour survives r cycles with <= k mismatches with probability
    // sum_{i<=k} C(r,i) 0.75^i 0.25^(r-i); the first round should leave a few percent alive.
    int first = ctx->queue_first;
    if (first == 0)
        first = a.k <= 0 ? 2 : (a.k <= 2 ? 4 : (a.k == 3 ? 6 : 8));   // measured on MI355X
    switch (first) {
    case 1: hipLaunchKernelGGL((k_scan_q<STRIDED, 1>), grid, dim3(kBlock), lds, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL((k_scan_q<STRIDED, 2>), grid, dim3(kBlock), lds, ctx->stream, a); break;
    case 3: hipLaunchKernelGGL((k_scan_q<STRIDED, 3>), grid, dim3(kBlock), lds, ctx->stream, a); break;
    case 4: hipLaunchKernelGGL((k_scan_q<STRIDED, 4>), grid, dim3(kBlock), lds, ctx->stream, a); break;
    case 6: hipLaunchKernelGGL((k_scan_q<STRIDED, 6>), grid, dim3(kBlock), lds, ctx->stream, a); break;
    default: hipLaunchKernelGGL((k_scan_q<STRIDED, 8>), grid, dim3(kBlock), lds, ctx->stream, a); break;
    }
    return 0;
}

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
    (void)hipFree(ctx->d_nbr_t);
    (void)hipFree(ctx->d_gbase);
    ctx->d_nbr_t = nullptr;
    ctx->d_gbase = nullptr;
}

// Build the device copy of the transposed table (first dense scan after new targets).
int ensure_dense_tables(wd_ctx *ctx)
{
    if (ctx->d_nbr_t)
        return WD_OK;
    const int groups = (ctx->T + kWave - 1) / kWave;
    const long long total = ctx->h_gbase.empty() ? 0 : ctx->h_gbase.back();
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_gbase, (size_t)(groups + 1) * sizeof(long long)));
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_nbr_t, std::max<long long>(1, total) * sizeof(int32_t)));
    WD_HIP(ctx, hipMemcpyAsync(ctx->d_gbase, ctx->h_gbase.data(), (size_t)(groups + 1) * sizeof(long long),
                               hipMemcpyHostToDevice, ctx->stream));
    if (groups > 0)
        hipLaunchKernelGGL(k_transpose_nbr, dim3(groups), dim3(kWave), 0, ctx->stream, ctx->d_centre,
                           ctx->d_lvl_off, ctx->d_nbr, ctx->d_gbase, ctx->d_nbr_t, ctx->T, ctx->levels);
    WD_HIP(ctx, hipGetLastError());
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));       // h_gbase may be reused by the caller
    return WD_OK;
}

bool valid_batches(int b1, int b2)
{
    for (auto &p : kHamShapes)
        if (p[0] == b1 && p[1] == b2)
            return true;
    return false;
}

}  // namespace

// -------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------
extern "C" {

int wd_version(void) { return 100; }

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
    wd_ctx *ctx = new wd_ctx();
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&ctx->d_status, sizeof(uint32_t)) != hipSuccess ||
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
    (void)hipFree(ctx->d_gbase);
    (void)hipHostFree(ctx->h_status);
    (void)hipFree(ctx->d_out_tile);
    (void)hipFree(ctx->d_out_pt);
    (void)hipFree(ctx->d_hits);
    (void)hipFree(ctx->d_hit_count);
    for (auto *sl : ctx->ingest_slots) {
        (void)hipHostFree(sl->pinned);
        if (sl->stream)
            (void)hipStreamDestroy(sl->stream);
        delete sl;
    }
    if (ctx->own_stream)
        (void)hipStreamDestroy(ctx->own_stream);
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
{
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
        ctx->profile = value ? 1 : 0;
    } else if (n == "null_stream") {
        // run on the HIP null (legacy default) stream, e.g. to order with a framework that
        // uses it; 0 returns to the context's own stream
        if (bind_device(ctx))
            return WD_ERR_HIP;
        WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->stream = value ? (hipStream_t) nullptr : ctx->own_stream;
    } else if (n == "dense_kernel") {
        ctx->dense_kernel = value < 0 ? -1 : (value ? 1 : 0);
    } else if (n == "queue_kernel") {
        ctx->queue_kernel = value ? 1 : 0;
    } else if (n == "queue_first") {
        if (value != 0 && value != 1 && value != 2 && value != 3 && value != 4 && value != 6 && value != 8)
            return fail(ctx, WD_ERR_ARG, "queue_first must be 0 (auto), 1, 2, 3, 4, 6 or 8");
        ctx->queue_first = (int)value;
    } else {
        return fail(ctx, WD_ERR_ARG, "unknown option " + n);
    }
    return WD_OK;
}

int wd_get_option(wd_ctx *ctx, const char *name, int64_t *value)
{
    if (!ctx || !name || !value)
        return WD_ERR_ARG;
    std::string n(name);
    if (n == "early_exit") *value = ctx->early_exit;
    else if (n == "targets_per_block") *value = ctx->tpb;
    else if (n == "batch_first") *value = ctx->batch_first;
    else if (n == "batch_next") *value = ctx->batch_next;
    else if (n == "profile") *value = ctx->profile;
    else if (n == "queue_kernel") *value = ctx->queue_kernel;
    else if (n == "dense_kernel") *value = ctx->dense_kernel;
    else if (n == "null_stream") *value = ctx->stream == nullptr ? 1 : 0;
    else if (n == "queue_first") *value = ctx->queue_first;
    else return fail(ctx, WD_ERR_ARG, "unknown option " + n);
    return WD_OK;
}

int wd_malloc(wd_ctx *ctx, size_t bytes, void **out_dev)
{
    if (!ctx || !out_dev)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    *out_dev = nullptr;
    WD_HIP(ctx, hipMalloc(out_dev, bytes ? bytes : 1));
    return WD_OK;
}

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

int wd_set_targets(wd_ctx *ctx, int T, int levels, const int32_t *centre, const int32_t *lvl_off,
                   const int32_t *nbr)
{
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
    ctx->has_targets = true;
    return WD_OK;
}

int wd_scan_async(wd_ctx *ctx, int n_tiles, int L, int mode, int k, const uint8_t *const *planes,
                  const uint8_t *const *filter, int64_t N, int64_t *out_tile_dev,
                  uint32_t *out_per_target_dev)
{
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
            kk = L;                 // every equal-length pair is within L substitutions
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
    if (L > 1)
        stride = (int64_t)(planes[1] - planes[0]);
    for (int i = 0; i < n_tiles && strided; i++)
        for (int c = 1; c < L; c++)
            if ((int64_t)(planes[(size_t)i * L + c] - planes[(size_t)i * L]) != stride * c) {
                strided = false;
                break;
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
    a.nbr_t = nullptr;
    a.gbase = nullptr;
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
    const bool dense_ok = !lev && ctx->early_exit && L >= 1 && ctx->k_max <= kDenseMaxK &&
                          levels <= 8 && kk <= 2 && kk >= 0;      // levels: LDS budget (35 KB)
    // (with k >= 2 nothing can die within the 2-cycle first round, so only on request)
    const bool use_dense = dense_ok && (ctx->dense_kernel == 1 ||
                                        (ctx->dense_kernel < 0 && ctx->T >= 65536 && kk <= 1));
    const int chunks = (ctx->T + ctx->tpb - 1) / ctx->tpb;
    const long long nblocks = lev_generic ? (long long)ctx->T * n_tiles
                              : use_dense ? (long long)((ctx->T + kBlock - 1) / kBlock) * n_tiles
                                          : (long long)chunks * n_tiles;
    if (nblocks > 0x7FFFFFFFll)
        return fail(ctx, WD_ERR_UNSUPPORTED, "grid too large; raise targets_per_block");
    dim3 grid((unsigned)nblocks);

    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (ctx->profile) {
        if (!ctx->free_events.empty()) {
            ev = ctx->free_events.back();
            ctx->free_events.pop_back();
        } else {
            WD_HIP(ctx, hipEventCreate(&ev.first));
            WD_HIP(ctx, hipEventCreate(&ev.second));
        }
        WD_HIP(ctx, hipEventRecord(ev.first, ctx->stream));
    }
    const bool use_queue = !lev && ctx->queue_kernel && ctx->early_exit && kk <= 254 &&
                           ctx->k_max <= (int64_t)kMaxPasses * kPass;
    if (use_dense) {
        int rc = ensure_dense_tables(ctx);
        if (rc)
            return rc;
        a.nbr_t = ctx->d_nbr_t;
        a.gbase = ctx->d_gbase;
        const size_t lds = (size_t)scan_dense_lds_dwords(levels) * sizeof(uint32_t);
        if (strided)
            hipLaunchKernelGGL((k_scan_dense<true>), grid, dim3(kBlock), lds, ctx->stream, a);
        else
            hipLaunchKernelGGL((k_scan_dense<false>), grid, dim3(kBlock), lds, ctx->stream, a);
    } else if (use_queue) {
        if (strided)
            launch_queue<true>(ctx, a, grid);
        else
            launch_queue<false>(ctx, a, grid);
    } else if (!lev) {
        if (strided)
            launch_ham<true>(ctx, a, grid);
        else
            launch_ham<false>(ctx, a, grid);
    } else if (lev_generic) {
        const int h = kk / 2;
        const size_t lds = lev_generic_lds_bytes(L, h);
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
    if (ctx->profile) {
        WD_HIP(ctx, hipEventRecord(ev.second, ctx->stream));
        ctx->events.push_back(ev);
    }
    return WD_OK;
}

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
{
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
}

int wd_hitlog_enable(wd_ctx *ctx, int64_t capacity)
{
    if (!ctx || capacity < 0)
        return WD_ERR_ARG;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->d_hits);
    ctx->d_hits = nullptr;
    ctx->hit_cap = 0;
    if (capacity > 0) {
        WD_HIP(ctx, hipMalloc((void **)&ctx->d_hits, (size_t)capacity * sizeof(wd_hit)));
        ctx->hit_cap = capacity;
    }
    WD_HIP(ctx, hipMemsetAsync(ctx->d_hit_count, 0, sizeof(unsigned long long), ctx->stream));
    return WD_OK;
}

int wd_hitlog_fetch(wd_ctx *ctx, wd_hit *out_host, int64_t max_records, int64_t *total_out)
{
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
}

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

// ---- targets from coordinates ----------------------------------------------------------
int wd_targets_from_coords(wd_ctx *ctx, const int32_t *x, const int32_t *y, int64_t n,
                           const int32_t *centres, int64_t n_centres, int levels,
                           const int32_t *max_dists, int64_t *P_out)
{
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
    int32_t *d_x = nullptr, *d_y = nullptr, *d_c = nullptr, *d_counts = nullptr;
    int32_t *d_off = nullptr, *d_nbr = nullptr, *d_centre = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_x); (void)hipFree(d_y); (void)hipFree(d_c); (void)hipFree(d_counts);
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
        d_c = nullptr;
    } else {
        std::vector<int32_t> iota((size_t)T);
        for (int i = 0; i < T; i++)
            iota[i] = i;
        WD_GEN_HIP(hipMemcpy(d_centre, iota.data(), (size_t)T * 4, hipMemcpyHostToDevice));
    }
    WD_GEN_HIP(hipMalloc((void **)&d_counts, std::max<size_t>(1, (size_t)T * levels) * 4));
    WD_GEN_HIP(hipMemsetAsync(ctx->d_status, 0, sizeof(uint32_t), ctx->stream));
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
    a.status = ctx->d_status;
    std::vector<int32_t> off((size_t)T * (levels + 1) + 1, 0);
    int64_t P = 0;
    if (T > 0) {
        hipLaunchKernelGGL((k_gen_rings<false>), dim3(T), dim3(kBlock), 0, ctx->stream, a);
        WD_GEN_HIP(hipGetLastError());
        std::vector<int32_t> counts((size_t)T * levels);
        WD_GEN_HIP(hipMemcpyAsync(counts.data(), d_counts, counts.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        WD_GEN_HIP(hipMemcpyAsync(ctx->h_status, ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
        WD_GEN_HIP(hipStreamSynchronize(ctx->stream));
        if (*ctx->h_status & 2u) {
            (void)hipMemsetAsync(ctx->d_status, 0, sizeof(uint32_t), ctx->stream);
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
        WD_GEN_HIP(hipMemcpyAsync(ctx->h_status, ctx->d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
        WD_GEN_HIP(hipStreamSynchronize(ctx->stream));
        if (*ctx->h_status & 4u) {
            (void)hipMemsetAsync(ctx->d_status, 0, sizeof(uint32_t), ctx->stream);
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
    ctx->has_targets = true;
    if (P_out)
        *P_out = P;
    return WD_OK;
}

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
{
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
}

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

int slot_reserve(wd_ctx::IngestSlot *s, size_t need)
{
    if (!s->stream && hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess)
        return WD_ERR_HIP;
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

// These two may be called from several host threads at once on one context (each call leases
// its own pinned buffer and copy stream); they do not touch the context's error string.
int wd_load_bcl_gz(wd_ctx *ctx, const char *path, uint8_t *dst_dev, int64_t n_clusters)
{
    if (!ctx || !path || !dst_dev || n_clusters < 0)
        return WD_ERR_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess)
        return WD_ERR_HIP;
    std::vector<uint8_t> raw;
    if (!slurp(path, raw))
        return WD_ERR_IO;                                  // FileNotFoundError in the reference
    SlotLease lease(ctx);
    const size_t want = (size_t)n_clusters + 4;
    int rc = slot_reserve(lease.slot, want + 64);
    if (rc)
        return rc;
    // gunzip (possibly several concatenated members) straight into the pinned buffer
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK)
        return WD_ERR_NOMEM;
    zs.next_in = raw.data();
    zs.avail_in = (uInt)std::min<size_t>(raw.size(), 0xFFFFFFFFu);
    size_t produced = 0;
    bool bad = raw.size() > 0xFFFFFFFFu;
    while (!bad) {
        zs.next_out = lease.slot->pinned + produced;
        zs.avail_out = (uInt)std::min<size_t>(want + 64 - produced, 0x7FFFFFFFu);
        const int zr = inflate(&zs, Z_NO_FLUSH);
        produced = (size_t)(zs.next_out - lease.slot->pinned);
        if (zr == Z_STREAM_END) {
            if (zs.avail_in == 0)
                break;
            if (inflateReset(&zs) != Z_OK)
                bad = true;
            continue;
        }
        if (zr != Z_OK || zs.avail_out == 0) {
            bad = zr != Z_OK;                              // avail_out == 0: more data than a plane
            break;
        }
        if (zs.avail_in == 0)
            break;                                         // truncated stream
    }
    inflateEnd(&zs);
    if (bad)
        return WD_ERR_IO;
    if (produced < 4)
        return WD_ERR_FORMAT;
    uint32_t header;
    memcpy(&header, lease.slot->pinned, 4);
    if ((int64_t)header != n_clusters)                     // bcl_direct_reader.py:338
        return WD_ERR_FORMAT;
    if (produced < want)
        return WD_ERR_INDEX;                               // the reference fails at slurped_file[idx]
    if (n_clusters > 0) {
        if (hipMemcpyAsync(dst_dev, lease.slot->pinned + 4, (size_t)n_clusters, hipMemcpyHostToDevice,
                           lease.slot->stream) != hipSuccess ||
            hipStreamSynchronize(lease.slot->stream) != hipSuccess)
            return WD_ERR_HIP;
    }
    return WD_OK;
}

int wd_load_filter(wd_ctx *ctx, const char *path, uint8_t *dst_dev, int64_t n_clusters)
{
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
    int rc = slot_reserve(lease.slot, (size_t)n_clusters + 64);
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
}

int wd_gather_wells(wd_ctx *ctx, const uint8_t *const *planes, int L, const int32_t *idx, int64_t n,
                    int64_t n_clusters, uint8_t *out_host)
{
    if (!ctx || L < 0 || n < 0 || (n > 0 && L > 0 && (!planes || !idx || !out_host)))
        return fail(ctx, WD_ERR_ARG, "bad gather arguments");
    for (int64_t i = 0; i < n; i++)
        if (idx[i] < 0 || idx[i] >= n_clusters)
            return fail(ctx, WD_ERR_INDEX, "well index outside the tile");
    if (n == 0 || L == 0)
        return WD_OK;
    if (bind_device(ctx))
        return WD_ERR_HIP;
    const uint8_t **d_pl = nullptr;
    int32_t *d_idx = nullptr;
    uint8_t *d_out = nullptr;
    int rc = WD_OK;
    auto done = [&](int code, const char *msg) {
        (void)hipFree(d_pl); (void)hipFree(d_idx); (void)hipFree(d_out);
        return code == WD_OK ? WD_OK : fail(ctx, code, msg);
    };
    if (hipMalloc((void **)&d_pl, (size_t)L * sizeof(void *)) != hipSuccess ||
        hipMalloc((void **)&d_idx, (size_t)n * 4) != hipSuccess ||
        hipMalloc((void **)&d_out, (size_t)n * L) != hipSuccess)
        return done(WD_ERR_NOMEM, "gather workspace");
    if (hipMemcpyAsync(d_pl, planes, (size_t)L * sizeof(void *), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d_idx, idx, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return done(WD_ERR_HIP, "gather upload");
    const long long total = (long long)n * L;
    hipLaunchKernelGGL(k_gather_wells, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       ctx->stream, d_pl, L, d_idx, (long long)n, d_out);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(out_host, d_out, (size_t)total, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return done(WD_ERR_HIP, "gather kernel");
    (void)rc;
    return done(WD_OK, "");
}

// ---- RCCL ----------------------------------------------------------------------------
int wd_comm_unique_id(void *out128)
{
    std::string err;
    if (!out128 || !rccl_load(err))
        return WD_ERR_COMM;
    return g_rccl.GetUniqueId(out128) == 0 ? WD_OK : WD_ERR_COMM;
}

int wd_comm_init(wd_ctx *ctx, int rank, int world, const void *id128)
{
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
}

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
