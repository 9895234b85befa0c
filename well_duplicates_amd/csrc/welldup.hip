// welldup.hip - MI355X (gfx950 / CDNA4) well-duplicate scanner: kernels + C ABI.
//
// Hot path of EdinburghGenomics/well_duplicates' count_well_duplicates.py:228-265 (target ->
// level -> neighbour compare) fused with the gather of Tile.get_seqs
// (bcl_direct_reader.py:158-220, :352-361) and the integer part of output_writer
// (count_well_duplicates.py:63-106).  Interface: include/welldup.h.
//
// Layout of the source (one translation unit; the .inc files are included below)
//   synth_kernels.inc     device twin of the synthetic-data spec
//   scan_sequential.inc   ScanArgs, HamState / LevState<H>, k_scan (Levenshtein, full gather)
//   scan_queue.inc        survivor queue + k_scan_q: the default equality / Hamming kernel
//   scan_dense.inc        dense path: one lane per target ("every well is a centre")
//   scan_lev_generic.inc  Levenshtein with any threshold (DP row in LDS)
//   gen_rings.inc         neighbour-index generator (prepare_cluster_indexes.py on device)
//   ingest_kernels.inc    gather of a few wells' bytes for the duplicate log
//   gpu_inflate.inc       DEFLATE on the GPU: one wave per .bcl.gz member (+ _kernels.inc: launch, CRC-32)
//   welldup.hip           RCCL binding, context, C ABI
//
// Design in one paragraph (DESIGN.md has the full story): HBM-bound byte/integer work, no MFMA.
// The gather of Tile.get_seqs is fused into the compare and is lazy in the cycle direction: a
// neighbour that already has more than k mismatches (or whose banded edit-distance row is all
// > k) can never become a duplicate, so its remaining bytes are never fetched; HBM hands out
// 128-byte lines, so the kernels are organised around touching few lines and hiding dependent
// round trips (LDS-staged metadata, software pipelining, stream-compacted survivor queues in
// LDS, wave ballots for the per-level counts, LDS histograms for the tallies).  No CUDA shims,
// no dual paths: gfx950 HIP only.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <zlib.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fcntl.h>
#include <emmintrin.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <new>
#include <system_error>
#include <memory>
#include <mutex>
#include <thread>
#include <type_traits>
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

#include "synth_kernels.inc"
#include "scan_sequential.inc"
#include "lev2_stream.inc"
#include "scan_queue.inc"
#include "scan_lines.inc"
#include "scan_dense.inc"
#include "scan_lev_generic.inc"
#include "gen_rings.inc"
#include "ingest_kernels.inc"
#include "gpu_inflate.inc"
#include "gpu_inflate_kernels.inc"

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
    int dense_tile_chunk = 8;  // dense kernel: tiles a group of targets is taken through by one wave
    uint32_t *d_sig = nullptr; // dense path: signature planes [n_tiles][sig_stride]
    size_t sig_cap = 0;        // elements
    unsigned long long *d_partial = nullptr;   // dense path: counter slots [n_tiles][kDenseSlots][stride]
    size_t partial_cap = 0;    // elements
    uint32_t *d_mask = nullptr;                // dense path: per-target hit masks, 4 targets per word
    size_t mask_cap = 0;       // words
    uint2 *d_queue = nullptr;                  // dense path: survivors of the signature round
    size_t queue_cap = 0;      // entries
    uint32_t *d_qcnt = nullptr;                // dense path: entries used per block region
    size_t qcnt_cap = 0;
    long long dense_queue_cap = 0;             // option: entries per 256-target block; 0 = from k
    uint32_t *d_cand = nullptr;                // dense path: survivor flags of a part (kDenseSlots + 1), two sets
    size_t cand_cap = 0;
    hipStream_t dense_hi = nullptr, dense_lo = nullptr;   // dense path: compare stages (high priority) / pack stages
    hipEvent_t dense_ev_start = nullptr, dense_ev_done = nullptr, dense_ev_cmp[2] = {nullptr, nullptr},
               dense_ev_pack[2] = {nullptr, nullptr};
    int dense_overlap = 0;                     // option: a dense scan as a pipeline of parts over two streams (measured: no gain, see launch_dense)
    int dense_part_tiles = 0;                  // option: tiles per part (0 = by the scan's size)
    int dense_pack_blocks = 1024;              // option: workgroups of the pack kernel beside a compare stage (0 = one per block of wells)
    uint4 *d_rows = nullptr;                   // dense path: packed cycles of the marked wells [n_tiles][N][kRowGroups]
    size_t rows_cap = 0;       // uint4 elements
    uint32_t *d_mark = nullptr;                // dense path: [3][n_tiles][mw_stride]: mark bits, word prefixes, block prefixes
    size_t mark_cap = 0;       // words
    int dense_pack = -1;                       // option: -1 = by survivor count, 0 = never, 1 = always
    int dense_windows = 1;                     // option: 0 = no LDS windows, every group gathers through L1
    int dense_nt = 1;                          // option: the pack kernel streams the planes with non-temporal loads (0.97 -> 0.83 ms per 8 tiles)
    int fast_inflate = 1;                      // option: own gunzip first, zlib as referee (0 = zlib only)
    int well_stride = 1;                       // option: 1 = a plane per cycle, 4 = cycles interleaved by four
    int profile = 0;           // HIP events around every n-th scan (0 = off)
    long long profile_seq = 0;

    // targets (device)
    int T = 0, levels = 0;
    int64_t P = 0;
    int32_t *d_centre = nullptr, *d_lvl_off = nullptr, *d_nbr = nullptr;
    int64_t idx_min = 0, idx_max = -1;
    int64_t k_max = 0;         // most neighbour slots of any target
    std::vector<long long> h_gbase;   // per 64-target group: start in the transposed table
    void *d_nbr_t = nullptr;          // built on first use of the dense path: int16 offsets or int32 indices
    bool nbr_t16 = false;
    int32_t *d_rel_t = nullptr;       // dense path: ring ends per target, level-major
    void *d_udelta = nullptr;         // dense path: shared neighbour offsets of uniform groups (type of d_nbr_t)
    uint8_t *d_guni = nullptr;        // dense path: which groups are uniform
    int32_t *d_ginfo = nullptr;       // dense path: window groups (k_dense_windows): union size | runs << 9
    uint16_t *d_uoff = nullptr;       // ... a union element's place in the wave's LDS window
    int2 *d_useg = nullptr;           // ... the runs of the window
    int32_t *d_wdelta = nullptr;      // ... the union's offsets
    uint8_t *d_wlev = nullptr;        // ... and rings
    uint32_t *d_wmask = nullptr;      // ... which elements each target has
    uint8_t *d_wfull = nullptr;       // ... place in the whole union of the elements the compare stage walks
    int dense_sym = 1;                // option: a symmetric neighbour relation (every well a centre) is compared from one end
    bool dense_sym_on = false;        // the tables built last are those of the one-ended compare
    int32_t centre0 = 0;              // ... and target t's centre is centre0 + t
    int win_kpad = 0;                 // row length of d_uoff / d_wdelta
    int win_dwords = 0;               // largest window of any group, in dwords
    long long n_uniform_groups = -1, n_window_groups = -1;   // -1: tables not built yet
    int32_t *d_pblocks = nullptr;              // dense path: target blocks with a group for the gather kernel (window groups in use)
    int n_pblocks = 0;
    uint32_t *d_tblflags = nullptr;   // scratch of the table builders: [0] offsets need 32 bits, [1] largest window
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
    uint8_t *d_stage = nullptr;       // wd_count_tiles: device copies of planes handed over in host memory
    size_t d_stage_cap = 0;

    // hit log
    wd_hit *d_hits = nullptr;
    unsigned long long *d_hit_count = nullptr;
    int64_t hit_cap = 0;
    uint8_t *d_gather = nullptr;               // wd_gather_wells' workspace (grow-only)
    size_t gather_cap = 0;
    size_t hit_alloc = 0;                      // records d_hits has room for (>= hit_cap: the buffer only grows)

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
        uint8_t *dev = nullptr;        // device scratch (packed CBCL block + chunk sums)
        size_t dev_cap = 0;
        uint8_t *file = nullptr;       // the compressed file, kept between calls (no mmap churn)
        size_t file_cap = 0;
        hipStream_t stream = nullptr;
        bool busy = false;
    };
    std::mutex ingest_mu;
    std::vector<IngestSlot *> ingest_slots;
    // the slots' streams: a few, shared (creating and destroying a stream costs milliseconds, and the
    // copies of all slots cross the same PCIe link anyway)
    static constexpr int kSlotStreams = 4;
    hipStream_t slot_streams[kSlotStreams] = {};

    // ingest through the GPU decoder (wd_load_bcl_gz_batch): the reader threads fill a ring of
    // pinned chunks; each chunk's files are copied into the batch's arena in device memory and
    // decoded by a launch of their own, on one of a pool of streams, while the next chunk is read
    struct InflateChunk {
        uint8_t *pinned = nullptr;
        hipEvent_t copied = nullptr;                   // the chunk's H2D copy is done: it may be refilled
    };
    static constexpr int kInflateChunks = 4;
    // The files are decoded in launches of kInflateLaunchFiles (two rounds of what the chip holds at
    // four waves per file; launches of 512 were no faster on 1600 files and slower on 800: a large
    // batch is bound by reading its files) or whatever the batch has, all on one stream: a wave's time per file does not
    // depend on how many files a launch holds, launches that share a hardware queue run one after the
    // other anyway (HIP multiplexes its streams onto ~4 of them; 4 launches on 4 streams took 3 kernel
    // times), and with a stream of their own for the chunk copies the next launch's files arrive while
    // this one decodes.  Consecutive batches take turns on two decode streams, so that the next batch's
    // workgroups move in as this batch's retire (400-file batches: 36 -> 33 ms per batch).
    static constexpr int kInflateStreams = 4;          // decode streams that exist; `inflate_decode_streams` of them are used
    int inflate_decode_streams = 2;                    // option / WD_DECODE_STREAMS
    int inflate_launch_files = 1024;                   // option / WD_LAUNCH_FILES: files per decoder launch within a batch
    unsigned inflate_launch_seq = 0;                   // launches so far: they take the decode streams in turn
    hipEvent_t inflate_joined[kInflateStreams] = {};   // a batch's launches on a stream are done
    size_t inflate_chunk_bytes = 16u << 20;            // option "inflate_chunk_mb" (pinning memory costs time: keep the ring small)
    InflateChunk inflate_chunks[kInflateChunks];
    std::mutex inflate_shared_mu;                      // the ring, the streams and their events are set up by one thread at a time
    size_t inflate_chunk_cap = 0;                      // bytes the chunks were allocated with
    char last_kernel[96] = "";                         // template name of the compare kernel of the last scan
    // the queue kernel's view of the targets: sorted by centre well, so that targets whose neighbourhoods
    // share cache lines sit in the same workgroup (install_sorted_view); null = the file's order is sorted
    int32_t *d_centre_q = nullptr, *d_lvl_off_q = nullptr, *d_perm = nullptr;
    // the line walk's view (build_line_tables, scan_lines.inc): the (target, slot) pairs sorted by neighbour well
    int line_walk = -1;                                // option: 1 = k_scan_lines where it applies, 0 = never, -1 = where the
                                                       // targets are dense enough for it to pay (line_walk_wanted)
    int32_t *d_lw_well = nullptr;
    uint32_t *d_lw_meta = nullptr, *d_lw_btgt = nullptr;
    int4 *d_lw_blk = nullptr;
    int32_t *d_lw_bcen = nullptr;
    int lw_blocks = -1;                                // -1: not built for the current targets; 0: does not apply to them
    int lw_tmax = 0;                                   // most targets of any block
    int line_pairs = 0;                                // option: pairs per block of the line walk (0 = kLwPairs)
    int sort_targets = 1;                              // option: use it (0: file order, as rounds 1 and 2)
    int sort_strip = 256;                              // option: width of the column strips of that order (0: plain well order)
    int lev2_closed = 1;                               // option: Levenshtein <= 2 by the closed form (0: banded DP)
    long long test_thread_limit = -1;                  // option (tests): pretend thread creation fails after this many per crew
    int inflate_waves = 0;                             // option: waves per file (1, 4, 8; 0 = by the launch's size)
    hipStream_t inflate_streams[kInflateStreams + 1] = {};
    hipEvent_t inflate_ready[kInflateStreams] = {};    // a launch's files are all in the arena
    // what a batch keeps until its last kernel is done; several, so that the next batch's files are read
    // and copied while this batch's are still being decoded
    struct InflateSlot {
        std::mutex mu;
        uint8_t *arena = nullptr;                      // compressed files of the batch (device)
        size_t arena_cap = 0;
        InfJob *h_jobs = nullptr, *d_jobs = nullptr;   // one entry per file of the batch
        InfResult *h_res = nullptr, *d_res = nullptr;
        size_t jobs_cap = 0;
        hipEvent_t done = nullptr;                     // the batch's results are on the host
    };
    static constexpr int kInflateSlots = 3;            // one batch read, one decoded, one waiting for its results
    InflateSlot inflate_slots[kInflateSlots];
    // One batch at a time reads, copies and launches, in the order the calls came in (a ticket each).
    std::atomic<unsigned> inflate_calls{0};
    unsigned inflate_serving = 0;                      // under inflate_mu
    std::mutex inflate_mu;
    std::condition_variable inflate_cv;
    std::atomic<long long> inflate_files_gpu{0}, inflate_files_host{0};   // how the files of all batches were decoded
    std::atomic<long long> inflate_files_early{0};     // ... of the host's: decoded by a reader thread while the batch was still being read
    std::atomic<long long> inflate_us_per_file{0};     // last batch: a file's time in the decode kernel, mean, microseconds
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
// first-round depth per band half-width H: where ~2-4 % of random neighbours are still alive
// under LevState::alive's lag-free criterion (k = 2H or 2H+1)
constexpr int lev_first(int H) { return H == 1 ? 7 : H == 2 ? 10 : H == 3 ? 13 : H == 4 ? 16 : H <= 6 ? 20 : 24; }
constexpr int kLevB2 = 8;

// One launch of the queue kernel; its template name - as the code object spells it - is kept for
// wd_last_kernel(), so that a counter profile can be tied to the kernel that really ran.
#define WD_LAUNCH_Q(STR, B1_, LEVH_, WS_, LDS_)                                                          \
    do {                                                                                                 \
        snprintf(ctx->last_kernel, sizeof(ctx->last_kernel), "k_scan_q<%s, %d, %d, %d>%s", (STR) ? "true" : "false", \
                 (int)(B1_), (int)(LEVH_), (int)(WS_), a.perm ? ", targets sorted by centre" : "");      \
        hipLaunchKernelGGL((k_scan_q<(STR), (B1_), (LEVH_), (WS_)>), grid, dim3(kBlock), (LDS_), ctx->stream, a); \
    } while (0)

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

template <bool STRIDED>
int launch_queue(wd_ctx *ctx, const ScanArgs &a, dim3 grid)
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

void drop_dense_tables(wd_ctx *ctx);
void drop_line_tables(wd_ctx *ctx);
int inflate_prepare_shared(wd_ctx *ctx, int n_chunks);

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

template <bool STRIDED, int H>
void launch_queue_lev(wd_ctx *ctx, const ScanArgs &a, dim3 grid)
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

// grow-only device scratch of the dense path
template <typename T>
int dense_reserve(wd_ctx *ctx, T *&ptr, size_t &cap, size_t need, const char *what)
{
    if (need <= cap)
        return WD_OK;
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    if (hipMalloc((void **)&ptr, need * sizeof(T)) != hipSuccess)
        return fail(ctx, WD_ERR_NOMEM, what);
    cap = need;
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
        WD_HIP(ctx, hipMemsetAsync(d.cand, 0, (kDenseSlots + 1) * sizeof(uint32_t), st));
        WD_HIP(ctx, hipMemsetAsync(d.partial, 0, (size_t)nt * kDenseSlots * d.partial_stride * sizeof(unsigned long long), st));
        WD_HIP(ctx, hipMemsetAsync(d.mask, 0, (size_t)nt * d.mask_stride * sizeof(uint32_t), st));
        if (d.rows)
            WD_HIP(ctx, hipMemsetAsync(d.mark, 0, (size_t)nt * d.mw_stride * sizeof(uint32_t), st));
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

// Automatic choice (option line_walk = -1).  Measured on MI355X (tools/line_probe.py, 96 tiles, k_scan_q ->
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
    ctx->d_lw_bcen = nullptr;
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
    ctx->lw_blocks = 0;
    const int T = ctx->T, levels = ctx->levels;
    const int64_t P = ctx->P;
    if (T < 1 || levels < 1 || P < 1 || P > (1ll << 26) || ctx->has_empty_level || ctx->k_max > 4095)
        return WD_OK;
    const size_t row = (size_t)levels + 1;
    std::vector<int32_t> off((size_t)T * row), nbr((size_t)P), cen((size_t)T);
    WD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    WD_HIP(ctx, hipMemcpy(cen.data(), ctx->d_centre, cen.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    WD_HIP(ctx, hipMemcpy(off.data(), ctx->d_lvl_off, off.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    WD_HIP(ctx, hipMemcpy(nbr.data(), ctx->d_nbr, nbr.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
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
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_lw_well, n * sizeof(int32_t)));
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_lw_meta, n * sizeof(uint32_t)));
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_lw_btgt, btgt.size() * sizeof(uint32_t)));
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_lw_blk, blk.size() * sizeof(int4)));
    std::vector<int32_t> bcen(btgt.size());
    for (size_t i = 0; i < btgt.size(); i++)
        bcen[i] = cen[(size_t)(btgt[i] & 0x7FFFFFFFu)];
    WD_HIP(ctx, hipMalloc((void **)&ctx->d_lw_bcen, bcen.size() * sizeof(int32_t)));
    WD_HIP(ctx, hipMemcpy(ctx->d_lw_bcen, bcen.data(), bcen.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    WD_HIP(ctx, hipMemcpy(ctx->d_lw_well, well.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    WD_HIP(ctx, hipMemcpy(ctx->d_lw_meta, meta.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    WD_HIP(ctx, hipMemcpy(ctx->d_lw_btgt, btgt.data(), btgt.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    WD_HIP(ctx, hipMemcpy(ctx->d_lw_blk, blk.data(), blk.size() * sizeof(int4), hipMemcpyHostToDevice));
    ctx->lw_blocks = (int)blk.size();
    return WD_OK;
}

// k_scan_lines for equality / Hamming <= k (first round of `first` cycles) or Levenshtein <= 2 (closed form)
template <bool STRIDED>
int launch_lines(wd_ctx *ctx, const ScanArgs &sa, int n_tiles, bool lev2, int first)
{
    LineArgs a;
    a.s = sa;
    a.s.perm = nullptr;
    a.pw = ctx->d_lw_well;
    a.pm = ctx->d_lw_meta;
    a.blk = ctx->d_lw_blk;
    a.btgt = ctx->d_lw_btgt;
    a.bcen = ctx->d_lw_bcen;
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
// Every entry point that allocates on the host is a function-try-block: a C++ exception ends at the
// boundary as an error code (a caller in C, ctypes or cgo cannot catch it: it would be std::terminate).
#define WD_CATCH                                                                                  \
    catch (const std::bad_alloc &) { return WD_ERR_NOMEM; }                                       \
    catch (...) { return WD_ERR_STATE; }

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
    if (ws == 4 && (use_dense || !strided || !(use_queue || lev_queue_il)))
        return fail(ctx, WD_ERR_UNSUPPORTED,
                    "the interleaved layout is read by the queue kernel only (equality, Hamming, Levenshtein <= 3)");

    // the line walk (scan_lines.inc): the pairs in the order of their neighbour wells, where the queue kernel
    // would run (planes, early exit, equality / Hamming, or Levenshtein <= 2 by the closed form)
    bool use_lines = false;
    if (line_walk_wanted(ctx) && !use_dense && ws == 1 && ctx->queue_kernel && ctx->early_exit && L >= 1 && levels <= 8 &&
        ((!lev && kk <= 254) || (lev && kk == 2 && ctx->lev2_closed && !lev_generic))) {
        // (targets of more slots than the queue kernel's four passes hold, up to 4095, are the walk's too)
        if (int rc = build_line_tables(ctx))
            return rc;
        use_lines = ctx->lw_blocks > 0;
    }

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
        const int rc = strided ? launch_lines<true>(ctx, a, n_tiles, lev, first) : launch_lines<false>(ctx, a, n_tiles, lev, first);
        if (rc)
            return rc;
    } else if (use_queue) {
        queue_view(ctx, a);
        if (strided)
            launch_queue<true>(ctx, a, grid);
        else
            launch_queue<false>(ctx, a, grid);
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
        if (h <= 1) { if (strided) launch_queue_lev<true, 1>(ctx, a, grid); else launch_queue_lev<false, 1>(ctx, a, grid); }
        else if (h == 2) { if (strided) launch_queue_lev<true, 2>(ctx, a, grid); else launch_queue_lev<false, 2>(ctx, a, grid); }
        else { if (strided) launch_queue_lev<true, 3>(ctx, a, grid); else launch_queue_lev<false, 3>(ctx, a, grid); }
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
int inflate_prepare_shared(wd_ctx *ctx, int n_chunks)
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
try {
    if (!ctx || !path || !dst_dev || !filter_dev || n_clusters < 0)
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
                       filter_dev, sums, (long long)n_clusters, excluded, dst_dev);
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
                                const uint8_t *const *filter_dev, uint8_t *const *dst_dev, int64_t n_clusters, int threads,
                                int *rc_out);

int wd_load_cbcl_batch(wd_ctx *ctx, int n, const char *const *paths, const int *tile_number,
                       const uint8_t *const *filter_dev, uint8_t *const *dst_dev, int64_t n_clusters, int threads,
                       int *rc_out)
{
    return guarded(ctx, [&] {
        return load_cbcl_batch_impl(ctx, n, paths, tile_number, filter_dev, dst_dev, n_clusters, threads, rc_out);
    });
}

static int load_cbcl_batch_impl(wd_ctx *ctx, int n, const char *const *paths, const int *tile_number,
                                const uint8_t *const *filter_dev, uint8_t *const *dst_dev, int64_t n_clusters, int threads,
                                int *rc_out)
{
    if (!ctx || n < 0 || (n && (!paths || !tile_number || !filter_dev || !dst_dev)) || n_clusters < 0 ||
        n_clusters > 0x7FFFFFF0ll)
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
                               filter_dev[i], sums, (long long)n_clusters, e.excluded, dst_dev[i]);
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
                rc[(size_t)i] = wd_load_cbcl_tile(ctx, paths[i], tile_number[i], filter_dev[i], n_clusters, dst_dev[i]);
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
