// wd_ctx.h - the context behind the C ABI's opaque wd_ctx, and the helpers the translation units
// share (namespace wd: defined once, in the unit named beside each).
#ifndef WD_CTX_H
#define WD_CTX_H
#include "wd_shared.h"

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
    int dense_tile_chunk = 16; // dense kernel: tiles a group of targets is taken through by one wave
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
    int fast_exit = 0;         // option: wd_destroy only waits for the device (the process is about to exit)
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
    int32_t *d_lw_bcen = nullptr, *d_lw_boff = nullptr;
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

#define WD_HIP(ctx, call)                                                                 \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return wd::fail((ctx), e_ == hipErrorOutOfMemory ? WD_ERR_NOMEM : WD_ERR_HIP, \
                            std::string(#call) + ": " + hipGetErrorString(e_));           \
    } while (0)

// Every entry point that allocates on the host is a function-try-block: a C++ exception ends at the
// boundary as an error code (a caller in C, ctypes or cgo cannot catch it: it would be std::terminate).
#define WD_CATCH                                                                                  \
    catch (const std::bad_alloc &) { return WD_ERR_NOMEM; }                                       \
    catch (...) { return WD_ERR_STATE; }

namespace wd {

// one per translation unit: the hash of the sources it was compiled from (wd_build_id lists them)
const char *unit_id_scan();
const char *unit_id_queue();
const char *unit_id_lines();
const char *unit_id_dense();
const char *unit_id_ingest();

// welldup_core.hip
int fail(wd_ctx *ctx, int code, const std::string &msg);
int bind_device(wd_ctx *ctx);
void drain_events(wd_ctx *ctx);

// welldup_dense.hip
void set_group_bases(wd_ctx *ctx, const int32_t *lvl_off, int T, int levels);
void drop_dense_tables(wd_ctx *ctx);
bool dense_rows_reserve(wd_ctx *ctx, int n_tiles, int64_t N);
int dense_part_size(const wd_ctx *ctx, int n_tiles, int tile_chunk);
int launch_dense(wd_ctx *ctx, const ScanArgs &a, int n_tiles, int64_t N, bool strided, size_t n_plane_ptrs,
                 int tile_chunk, bool lev2);

// welldup_queue.hip
int launch_queue(wd_ctx *ctx, const ScanArgs &a, dim3 grid, bool strided);
void launch_queue_lev(wd_ctx *ctx, const ScanArgs &a, dim3 grid, bool strided, int h);

// welldup_lines.hip
bool line_walk_wanted(const wd_ctx *ctx);
void drop_line_tables(wd_ctx *ctx);
int build_line_tables(wd_ctx *ctx);
int launch_lines(wd_ctx *ctx, const ScanArgs &sa, int n_tiles, bool lev2, int first, bool strided);

// welldup_ingest.hip
int inflate_prepare_shared(wd_ctx *ctx, int n_chunks);

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

// grow-only device scratch of the dense path and the line walk
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

}  // namespace wd

#endif
