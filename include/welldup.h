/*
 * welldup.h - C ABI of libwelldup.so, the MI355X (gfx950) well-duplicate scanner.
 *
 * The reference (EdinburghGenomics/well_duplicates) has no plugin or FFI interface: the
 * scan path is inline Python.  This ABI is cut at the seams a replacement has to honour
 * (SURVEY.md section 8b); each entry point cites the reference code it stands in for
 * (paths relative to /root/reference).  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add to count_well_duplicates.py.
 *
 * Conventions
 *   - plain C: opaque context, plain pointers and sizes, caller-allocated outputs.
 *   - every function returns WD_OK (0) or a negative WD_ERR_* code; wd_strerror() names the
 *     code, wd_last_error() gives the detail (e.g. the HIP error string) for that context.
 *   - one context per GPU; a context is thread-compatible (no hidden globals), not
 *     thread-safe: serialise calls on one context.
 *   - "device pointer" = address valid on the context's GPU (hipMalloc, a torch tensor's
 *     data_ptr(), or wd_malloc below).  "host pointer" = ordinary process memory.
 *   - error codes map back to the reference's exception classes:
 *       WD_ERR_INDEX        -> IndexError       (bcl_direct_reader.py:186-192)
 *       WD_ERR_EMPTY_LEVEL  -> AssertionError   (count_well_duplicates.py:249)
 *       WD_ERR_ARG          -> ValueError
 *       WD_ERR_NO_WELLS     -> RuntimeError     (prepare_cluster_indexes.py:70-76)
 *       WD_ERR_IO           -> FileNotFoundError (bcl_direct_reader.py:207-216: the file cannot be opened)
 *       WD_ERR_CORRUPT      -> gzip.BadGzipFile / zlib.error (:208-209: what gzip.open().read() raises on bad data)
 *       WD_ERR_TRUNCATED    -> EOFError         (:208-209: ... on a file that ends early)
 *       WD_ERR_FORMAT       -> AssertionError   (bcl_direct_reader.py:151, :236, :338)
 *       everything else     -> RuntimeError
 */
#ifndef WELLDUP_H
#define WELLDUP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WD_OK 0
#define WD_ERR_ARG (-1)
#define WD_ERR_INDEX (-2)
#define WD_ERR_EMPTY_LEVEL (-3)
#define WD_ERR_HIP (-4)
#define WD_ERR_NOMEM (-5)
#define WD_ERR_STATE (-6)
#define WD_ERR_UNSUPPORTED (-7)
#define WD_ERR_COMM (-8)
#define WD_ERR_NO_WELLS (-9)
#define WD_ERR_IO (-10)
#define WD_ERR_FORMAT (-11)
#define WD_ERR_CORRUPT (-12)
#define WD_ERR_TRUNCATED (-13)

/* Compare modes.  The reference counts a duplicate when dist <= edit_distance
 * (count_well_duplicates.py:258) with dist = Levenshtein.distance, or Levenshtein.hamming
 * under --hamming (:200).  WD_MODE_EQ is the `-e 0` corner (plain string equality under
 * either metric); k is ignored there. */
#define WD_MODE_EQ 0
#define WD_MODE_HAMMING 1
#define WD_MODE_LEVENSHTEIN 2

#define WD_MAX_LEVELS 32
#define WD_INVALID_TARGET 0xFFFFFFFFu

typedef struct wd_ctx wd_ctx;

/* ---- library / context ------------------------------------------------------------- */
int wd_version(void);                    /* major*10000 + minor*100 + patch */
/* Hashes of the sources the library was built from, set at compile time: "<id> core=<id> scan=<id> queue=<id>
 * lines=<id> dense=<id> ingest=<id>" - sha256 (first 16 hex digits) of all of csrc/ and this header, then of
 * each translation unit's own sources (csrc/welldup_<unit>.hip and what it includes).  Counter profiles and
 * resource tables carry them, so that a measurement is tied to the code that produced it and not merely to a
 * kernel's name: bench.py quotes a counter run only if the unit the kernel lives in is unchanged.
 * "unknown" for a build outside _lib.build(). */
const char *wd_build_id(void);
const char *wd_strerror(int code);
const char *wd_last_error(const wd_ctx *ctx);

/* Creates a context on GPU `device_id` (-1 = current device).  Returns NULL on failure;
 * wd_create_status() then holds the WD_ERR_* code of the failed attempt. */
wd_ctx *wd_create(int device_id);
int wd_create_status(void);
void wd_destroy(wd_ctx *ctx);

/* Run on a caller-owned HIP stream (hipStream_t, e.g. a torch.cuda.Stream's cuda_stream);
 * NULL returns to the context's own (non-blocking) stream.  To run on the HIP null stream
 * itself (torch's default stream has handle 0) use wd_set_option("null_stream", 1). */
int wd_set_stream(wd_ctx *ctx, void *hip_stream);
int wd_synchronize(wd_ctx *ctx);

/* Tunables, by name (default): "early_exit" (1), "targets_per_block" (64), "queue_kernel" (1),
 * "queue_first" (0 = from k), "batch_first" (4), "batch_next" (4), "profile" (0),
 * "null_stream" (0), "dense_kernel" (-1 = automatic: lane-per-target kernel when T >= 65536),
 * "dense_tile_chunk" (16: tiles one wave takes a group of 64 targets through in the dense path),
 * "dense_queue_cap" (0 = 16, 32 for Levenshtein: survivor entries per 64 targets, at most 64), "dense_pack" (-1 = settle the
 * survivors on packed rows of the wells they involve whenever there are any, 0 = byte by byte on
 * the planes, 1 = rows always), "dense_windows" (1: groups of consecutive centres compare their
 * neighbours' signatures from LDS windows; 0 = every group gathers them through L1), "dense_sym" (1: when
 * every well is a centre and the neighbour relation is symmetric - b in ring r of a exactly when a in ring
 * r of b, every ring in ascending order: checked once per targets set - the dense path compares each pair
 * from its lower well only and records a duplicate for both targets; 0 = every pair from both ends),
 * "line_walk" (-1: sampled targets of more than 127 neighbour slots on average are scanned pair by pair in
 * the order of the neighbour wells, so that a cache line is fetched once per cycle however many targets
 * want it; 1 = wherever it applies (equality, Hamming,
 * Levenshtein <= 2 on plane-per-cycle input), 0 = never), "line_pairs" (0 = 12288: pairs per workgroup of
 * that walk),
 * "inflate_warm" (write-only: sets up the batch loaders' pinned ring, streams and events now instead of
 * inside the first batch; may be called from a thread of its own), "fast_inflate" (1:
 * the loaders try the library's own gunzip before zlib; the environment variable WD_FAST_INFLATE
 * sets the default), "well_stride" (1 = a plane per cycle; 4 = interleaved, see wd_interleave4),
 * "fast_exit" (0; 1 = the process exits right after wd_destroy: the call then waits for the device and
 * frees nothing - no unpinning of the ingest ring, no stream destruction, 25 ms less per run).
 * Read-only (wd_get_option; -1 before the first dense scan of the current targets):
 * "dense_uniform_groups" (64-target groups of consecutive centres that share their neighbour
 * offsets), "dense_window_groups" (those scanned through signature windows in LDS),
 * "dense_window_dwords" (LDS dwords per wave of the largest window), "dense_sym_on" (1: the tables
 * built last are those of the one-ended compare), "line_walk_blocks" (workgroups per tile of the line
 * walk's tables: -1 not built, 0 = the walk does not apply to these targets).
 * Unknown names return WD_ERR_ARG. */
int wd_set_option(wd_ctx *ctx, const char *name, int64_t value);
int wd_get_option(wd_ctx *ctx, const char *name, int64_t *value);

/* Device memory helpers, so a host program needs no other GPU runtime. */
int wd_malloc(wd_ctx *ctx, size_t bytes, void **out_dev);
int wd_free(wd_ctx *ctx, void *dev);
int wd_memcpy_h2d(wd_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int wd_memcpy_d2h(wd_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int wd_memset(wd_ctx *ctx, void *dst_dev, int value, size_t bytes);

/* ---- targets ----------------------------------------------------------------------- */
/*
 * Replaces: load_targets()/AllTargets/Target as consumed by the compare loop
 * (target.py:6-40, :56-61, :118-128; count_well_duplicates.py:202-204, :228-230, :248).
 *
 * T targets in file order; for target t the wells of ring l (1-based, Target.get_indices(l))
 * are nbr[lvl_off[t*(levels+1) + l-1] .. lvl_off[t*(levels+1) + l]).  Host pointers; the
 * arrays are copied to the GPU and stay resident (one s.locs, hence one targets file, per
 * flowcell).  levels <= WD_MAX_LEVELS.  Offsets must be non-decreasing and lie in
 * [0, P] with P = lvl_off[T*(levels+1) - 1] (else WD_ERR_ARG).
 */
int wd_set_targets(wd_ctx *ctx, int T, int levels, const int32_t *centre,
                   const int32_t *lvl_off, const int32_t *nbr);

/*
 * Replaces the index producer: get_indexes()/yield_coords() of prepare_cluster_indexes.py
 * (:38-78, :99-116) for a list of centre wells, or for every well (centres == NULL: the
 * all-centres mode of BASELINE config 5, out of reach of the 70 ms/target reference).
 *
 *   x, y       host arrays of n pixel coordinates as the reference decodes them from s.locs:
 *              int(v * 10 + 1000.5) (:110-112).
 *   centres    host array of n_centres well indices (targets, in output order) or NULL.
 *   max_dists  levels+1 increasing pixel radii; ring r (0-based) holds wells with
 *              max_dists[r] < dist <= max_dists[r+1] (:61-63; the reference's table is
 *              {1,22,42,62,82,102}).  Only records max(0, c-20000) .. c+20001 are examined
 *              (:52-67) and every ring is emitted in ascending index order, like the
 *              reference's scan.  A centre with an empty ring fails the call with
 *              WD_ERR_NO_WELLS (:70-76).
 * On success the result is installed as the context's targets (as wd_set_targets) and stays
 * on the GPU; wd_targets_info / wd_get_targets read it back, e.g. to write the targets file.
 */
int wd_targets_from_coords(wd_ctx *ctx, const int32_t *x, const int32_t *y, int64_t n,
                           const int32_t *centres, int64_t n_centres, int levels,
                           const int32_t *max_dists, int64_t *P_out);
int wd_targets_info(wd_ctx *ctx, int *T, int *levels, int64_t *P);
int wd_get_targets(wd_ctx *ctx, int32_t *centre, int32_t *lvl_off, int32_t *nbr);

/* ---- the scan ---------------------------------------------------------------------- */
/*
 * Replaces, for n_tiles tiles at once: the per-tile body of main()
 * (count_well_duplicates.py:212-265) - Tile.get_seqs() for every target well
 * (bcl_direct_reader.py:158-220: byte 0 -> 'N', else "ACGT"[byte & 3], :352-361; filter
 * flag = byte & 1, :246), the centre filter gate (:236-237), the per-level neighbour
 * compare (:244-262) - plus the integer part of output_writer (:63-106).
 *
 *   planes  host array of n_tiles*L DEVICE pointers: planes[i*L + c] -> the N raw BCL
 *           payload bytes (the file content after its 4-byte count header) of tile i,
 *           c-th scanned cycle.  With --cycles a-b,c-d the ranges are concatenated, as
 *           the reference joins the per-range strings (:238, :251).
 *   filter  host array of n_tiles DEVICE pointers to the N raw .filter payload bytes
 *           (after the 12-byte header).
 *   N       clusters per tile.  Any centre or neighbour index outside [0, N) fails the
 *           whole call with WD_ERR_INDEX before anything is launched (:186-192).
 *   mode,k  WD_MODE_*; a neighbour is a duplicate when dist(centre, neighbour) <= k.
 *   out_tile        n_tiles rows of 1 + 5*levels int64:
 *                     [0]                     valid targets (centre passed the filter)
 *                     [1 + 0*levels + l]      Wells: ring sizes summed over valid targets
 *                     [1 + 1*levels + l]      Dups
 *                     [1 + 2*levels + l]      Hit:   targets with >= 1 dup at level l
 *                     [1 + 3*levels + l]      first: targets whose innermost hit level is l
 *                     [1 + 4*levels + l]      last:  targets whose outermost hit level is l
 *                   AccO = prefix sums of first, AccI = suffix sums of last (:80-89).
 *   out_per_target  optional (NULL to skip): n_tiles*T*levels dup counts in target file
 *                   order, WD_INVALID_TARGET in every level of a target whose centre failed
 *                   the filter - enough to rebuild the reference's lane_dupl (:226, :265).
 *
 * wd_count_tiles: out_* are HOST pointers; synchronous; returns WD_ERR_EMPTY_LEVEL if a
 * valid target has an empty ring (:249).  Its planes / filters may also point to HOST memory
 * (each is then copied to the GPU first - convenient, and as slow as PCIe).
 * wd_scan_async: out_* are DEVICE pointers; work is queued on the context's stream and the
 * call returns at once; wd_scan_status() synchronises and returns the deferred status.
 */
int wd_count_tiles(wd_ctx *ctx, int n_tiles, int L, int mode, int k,
                   const uint8_t *const *planes, const uint8_t *const *filter, int64_t N,
                   int64_t *out_tile, uint32_t *out_per_target);

int wd_scan_async(wd_ctx *ctx, int n_tiles, int L, int mode, int k,
                  const uint8_t *const *planes, const uint8_t *const *filter, int64_t N,
                  int64_t *out_tile_dev, uint32_t *out_per_target_dev);
int wd_scan_status(wd_ctx *ctx);

/*
 * Ingest: files of a run directory straight into device memory, replacing the file side of
 * Tile.get_seqs / _get_filter_offsets / _get_seqs_from_bcl (bcl_direct_reader.py:200-216,
 * :232-240, :333-345).  wd_load_bcl_gz gunzips `<cycle dir>/<tile>.bcl.gz` (zlib) into a pinned
 * staging buffer, checks the uint32 cluster count (:338) and copies the n_clusters payload bytes
 * to dst_dev; wd_load_filter does the same for a .filter file (header 0, 3, n: :148-152).
 * Both are THREAD-SAFE on one context (each call leases its own staging buffer and copy
 * stream): call them from a pool of host threads so that gunzip, PCIe and the GPU overlap.
 * wd_gather_wells returns out[w*L + c] = planes[c][idx[w]] (host output; planes[c][4 * idx[w]]
 * while the "well_stride" option is 4): the bytes of the
 * wells the stderr duplicate log prints, without a host copy of the planes.
 */
int wd_load_bcl_gz(wd_ctx *ctx, const char *path, uint8_t *dst_dev, int64_t n_clusters);
/* The same into the interleaved layout (wd_interleave4): dst_dev = address of well 0's byte of this
 * cycle inside its group of four (group base + cycle % 4), well_stride = 4; 1 = wd_load_bcl_gz. */
int wd_load_bcl_gz_strided(wd_ctx *ctx, const char *path, uint8_t *dst_dev, int64_t n_clusters, int well_stride);
int wd_load_filter(wd_ctx *ctx, const char *path, uint8_t *dst_dev, int64_t n_clusters);
/* A batch of .bcl.gz files (typically all cycles of a few tiles) decoded ON THE GPU: the same
 * contract as n_files calls of wd_load_bcl_gz (bcl_direct_reader.py:200-216, :333-345), file i ->
 * dst_dev[i] (4-byte aligned, n_clusters bytes).  `threads` host threads only read the compressed
 * files into pinned memory; the compressed bytes cross PCIe, four or eight waves per file inflate them
 * (csrc/gpu_inflate.inc: every lane decodes a piece of a block's bit stream), a second
 * kernel takes the CRC-32 and the gzip trailer (CRC, length), the cluster count and the plane size
 * are checked.  The GPU decoder is an accelerator, not an authority: a file it declines or whose
 * checks fail (corrupt, truncated, several members, header options, a piece of the stream that
 * expands more than 256-fold) is loaded again by wd_load_bcl_gz, whose return code is the one
 * reported.  rc (nullable): n_files WD_* codes; the return value is the first non-zero one.
 * THREAD-SAFE, and meant to be called from several threads: the calls take turns at the pinned ring
 * and the copy stream in the order they were made, and up to three are in flight - one reading its
 * files, one being decoded, one waiting for its results - so a caller that keeps three batches of
 * about 512 files queued (what one launch of the decoder holds) sees the rate of the chunk copies.
 * Safe beside the other loaders.
 * wd_get_option "inflate_files_gpu" / "inflate_files_host" count how the files of all batches were
 * decoded; option "inflate_chunk_mb" (default 16) sizes the 4 pinned staging chunks, "inflate_waves"
 * (0 = by the launch's size, 1, 4, 8) how many waves decode one file together. */
int wd_load_bcl_gz_batch(wd_ctx *ctx, int n_files, const char *const *paths, uint8_t *const *dst_dev,
                         int64_t n_clusters, int threads, int *rc);
/* The same with the tiles' .filter files in the batch: is_filter[i] != 0 marks file i as a .filter
 * (header 0, 3, n_clusters checked as wd_load_filter does, :148-152, :236-240; its n_clusters bytes are
 * copied, not decoded).  is_filter = NULL: wd_load_bcl_gz_batch.  One call then brings everything a
 * batch of tiles needs into HBM over one ring of pinned memory and two streams.
 * well_stride = 4 lands every decoded plane in its byte lane of an interleaved group, as
 * wd_load_bcl_gz_strided does (dst_dev[i] = group base + cycle % 4); filters are never strided. */
int wd_load_tile_files_batch(wd_ctx *ctx, int n_files, const char *const *paths, uint8_t *const *dst_dev,
                             const uint8_t *is_filter, int64_t n_clusters, int well_stride, int threads, int *rc);
/* Resident layout option for the equality / Hamming scan of sampled targets.  A line of HBM holds
 * 128 wells of ONE cycle in the BCL files' plane-per-cycle layout, and the scan wants ~11
 * neighbouring wells of SEVERAL cycles: with the cycles interleaved by four ([group of 4 cycles]
 * [well][4 bytes]) one dword is a well's first round and the scan touches half the lines.
 * wd_interleave4 builds one group from up to four device planes (NULL = cycle beyond the read),
 * queued on the context's stream; dst_dev: 4 * n_clusters bytes, 4-byte aligned.  A scan of such
 * a batch passes plane pointers that describe it (cycle c of a tile at base + (c / 4) * group
 * stride + c % 4) after wd_set_option(ctx, "well_stride", 4); other kernels (Levenshtein, the
 * dense path) answer WD_ERR_UNSUPPORTED. */
int wd_interleave4(wd_ctx *ctx, const uint8_t *const src[4], int64_t n_clusters, uint8_t *dst_dev);

/* The host-side gunzip behind the two loaders, exposed for testing and reuse; no GPU involved.
 * All members of a gzip file src[0, src_len) -> dst (at most dst_cap bytes), *produced = bytes
 * written.  mode 0: zlib.  mode 1: the library's own RFC 1951/1952 decoder (CRC-32 and length of
 * every member checked); it declines - WD_ERR_UNSUPPORTED - whatever it does not like, and the
 * loaders then decode the file with zlib.  WD_ERR_CORRUPT / WD_ERR_TRUNCATED: bad or short
 * stream; WD_ERR_IO: more data than dst_cap. */
int wd_gunzip(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap, size_t *produced, int mode);

/* NovaSeq: one tile's block of a `L00<lane>_<surface>.cbcl` file -> an n_clusters-byte plane on
 * the device, replacing _get_seqs_from_cbcl (bcl_direct_reader.py:255-325): header and tile
 * table checks (:263-295, WD_ERR_FORMAT), gunzip of the tile's block (:300-301), nibble
 * expansion on the GPU (:316-321) including the excluded-wells indirection through the tile's
 * filter (:303-314; filter_dev must already hold the tile's filter bytes).  Thread-safe like
 * wd_load_bcl_gz. */
int wd_load_cbcl_tile(wd_ctx *ctx, const char *path, int tile_number, const uint8_t *filter_dev,
                      int64_t n_clusters, uint8_t *dst_dev);
/* The same into the interleaved layout: well_stride = 4, dst_dev = group base + cycle % 4 (as
 * wd_load_bcl_gz_strided); 1 = wd_load_cbcl_tile. */
int wd_load_cbcl_tile_strided(wd_ctx *ctx, const char *path, int tile_number, const uint8_t *filter_dev,
                              int64_t n_clusters, uint8_t *dst_dev, int well_stride);
/* A batch of tile blocks, inflated on the GPU: entry i = the block of tile tile_number[i] in the .cbcl
 * file paths[i] -> dst_dev[i], as n calls of wd_load_cbcl_tile would (bcl_direct_reader.py:255-325;
 * filter_dev[i] must already hold the tile's filter bytes).  The files' headers and tile tables are
 * read once per file on the host; reader threads bring the blocks into pinned memory; one launch of the
 * DEFLATE kernel (csrc/gpu_inflate.inc) decodes them into packed planes, which the expansion kernels
 * turn into byte planes.  CRC-32 and length of every block are checked; an entry the GPU decoder
 * declines or whose checks fail is loaded again by wd_load_cbcl_tile, whose return code is reported.
 * rc (nullable): n WD_* codes; the return value is the first non-zero one. */
int wd_load_cbcl_batch(wd_ctx *ctx, int n, const char *const *paths, const int *tile_number,
                       const uint8_t *const *filter_dev, uint8_t *const *dst_dev, int64_t n_clusters, int threads,
                       int *rc);
/* ... with every expanded plane landing in its byte lane of an interleaved group (well_stride = 4, dst_dev[i] =
 * group base + cycle % 4), so that a NovaSeq run can be kept resident in the layout the sampled scans read in
 * half the cache lines; well_stride = 1: wd_load_cbcl_batch. */
int wd_load_cbcl_batch_strided(wd_ctx *ctx, int n, const char *const *paths, const int *tile_number,
                               const uint8_t *const *filter_dev, uint8_t *const *dst_dev, int64_t n_clusters,
                               int well_stride, int threads, int *rc);
int wd_gather_wells(wd_ctx *ctx, const uint8_t *const *planes, int L, const int32_t *idx, int64_t n,
                    int64_t n_clusters, uint8_t *out_host);

/*
 * Duplicate log: the (centre, well, distance) records the reference prints to stderr for
 * every duplicate (count_well_duplicates.py:258-262).  Enable with a capacity before a
 * scan; afterwards fetch the records (unordered; sort by tile/target/slot on the host).
 * total_out receives the number of duplicates seen, which may exceed the capacity.
 */
typedef struct wd_hit {
    int32_t tile;      /* index into the call's tile list */
    int32_t target;    /* index into the targets file */
    int32_t slot;      /* position in nbr[] */
    int32_t dist;      /* distance as the reference would print it (0 in WD_MODE_EQ) */
} wd_hit;
/* wd_hitlog_fetch copies min(total, max_records, capacity) records: when *total_out exceeds the capacity the
 * log overflowed and the records beyond it are lost (wd_get_option "hitlog_capacity" reads the capacity). */
int wd_hitlog_enable(wd_ctx *ctx, int64_t capacity);   /* 0 disables */
int wd_hitlog_fetch(wd_ctx *ctx, wd_hit *out_host, int64_t max_records, int64_t *total_out);

/* Timing of the scan kernels with HIP events on the stream they run on ("profile" = n > 0: events
 * around every n-th scan call): total milliseconds and timed launches since the last reset. */
int wd_profile_get(wd_ctx *ctx, double *total_ms, int64_t *launches);
int wd_profile_reset(wd_ctx *ctx);
/* What this GPU's memory gives a kernel that only reads: `passes` passes over `bytes` of device memory
 * (16-byte aligned; 16 bytes per lane and load, non-temporal), timed with HIP events on the context's stream;
 * *ms_per_pass = the mean.  Measurement aid (bench.py quotes the scan kernels' rates beside it); the
 * reference has nothing like it. */
int wd_stream_read_probe(wd_ctx *ctx, const void *src_dev, size_t bytes, int passes, double *ms_per_pass);
/* Template name of the compare kernel the last scan launched, as the code object spells it
 * (e.g. "k_scan_q<true, 2, 0, 1>"; "" before the first scan): ties a counter profile of a kernel
 * to the kernel a measurement really ran.  The string lives in the context. */
const char *wd_last_kernel(const wd_ctx *ctx);

/* ---- multi-GPU --------------------------------------------------------------------- */
/*
 * Tiles are independent (count_well_duplicates.py:207-226: one lane_dupl entry per tile,
 * summed only in output_writer), so ranks take disjoint tile ranges and the only exchange
 * is one in-place int64 sum of the zero-initialised [all tiles, 1+5*levels] block.
 * RCCL is bound at run time (dlopen), so single-GPU use needs no RCCL at all.
 */
#define WD_UNIQUE_ID_BYTES 128
int wd_comm_unique_id(void *out128);                      /* rank 0, then share the bytes */
int wd_comm_init(wd_ctx *ctx, int rank, int world, const void *id128);
int wd_allreduce_counts(wd_ctx *ctx, int64_t *buf_dev, size_t n);   /* in place, sum */
int wd_comm_destroy(wd_ctx *ctx);

/* ---- synthetic data (bench / tests) ------------------------------------------------ */
/* Device-side twin of well_duplicates_amd/synth.py: byte-identical planes and filters. */
typedef struct wd_synth_spec {
    uint64_t seed;
    int64_t n_clusters;
    int64_t row;
    uint32_t nocall_per_64k, pass_per_64k, plant_per_64k;
    uint32_t filter_noise;
    uint32_t tile_dead;
    uint32_t plant_far;
    uint32_t qual_levels;     /* distinct quality values (0 = 39) */
} wd_synth_spec;
int wd_synth_plane(wd_ctx *ctx, uint8_t *dst_dev, const wd_synth_spec *spec, int lane, int tile,
                   int cycle);
int wd_synth_filter(wd_ctx *ctx, uint8_t *dst_dev, const wd_synth_spec *spec, int lane, int tile);

#ifdef __cplusplus
}
#endif
#endif /* WELLDUP_H */
