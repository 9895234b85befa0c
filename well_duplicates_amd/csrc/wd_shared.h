// wd_shared.h - what every translation unit of libwelldup.so sees: the C ABI, the constants and the
// argument blocks that cross from the launch code of one unit into another.  Internal (not installed).
#ifndef WD_SHARED_H
#define WD_SHARED_H
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
#include <utility>
#include <string>
#include <vector>

#include "welldup.h"

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWaves = kBlock / kWave;
constexpr int kMaxLevels = WD_MAX_LEVELS;
constexpr int kCounters = 1 + 5 * kMaxLevels;
constexpr uint32_t kStatusEmptyLevel = 1u;
constexpr int kXcds = 8;              // MI355X: 8 accelerator dies, workgroup i runs on XCD i % 8
constexpr int kMaxTpb = 64;           // targets per workgroup of the queue kernels, at most
constexpr int kPass = 127;            // neighbour slots per pass of the queue kernels (scan_queue.inc)
constexpr int kMaxPasses = 4;         // wd_scan_async falls back to k_scan for targets with more slots

// first-round depth of the banded-Levenshtein kernels per band half-width H: where ~2-4 % of random
// neighbours are still alive under LevState::alive's lag-free criterion (k = 2H or 2H+1)
constexpr int lev_first(int H) { return H == 1 ? 7 : H == 2 ? 10 : H == 3 ? 13 : H == 4 ? 16 : H <= 6 ? 20 : 24; }

// dense path (scan_dense.inc), as far as the dispatch in wd_scan_async needs to know it
// The chain of kernels a dense scan launches, as wd_last_kernel() names it: bumped whenever a kernel
// joins, leaves or changes its job, so that counter profiles of an older chain read as stale.
constexpr int kDenseChainVersion = 5;
constexpr int kSigCycles = 10;
constexpr int kRowGroups = 4;         // packed wells: 64 bytes (4 x uint4) per marked well, cycles 10 .. 169 (k_dense_pack)
constexpr int kDenseMaxK = 1 << 17;   // a slot index must fit the survivor entry's 17 bits

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
    const int32_t *perm;            // nullable: centre / lvl_off are a sorted view, perm[t] = the target's index in the file
    int T, levels, L, k, tpb, early, check_empty, log_hits;
};

// Kept out of the kernel-argument registers: only duplicates and malformed targets need them.
struct ScanRare {
    uint32_t *status;
    wd_hit *hits;
    unsigned long long *hit_count;
    long long hit_cap;
};

// gpu_inflate.inc (the ingest unit completes them)
struct InfJob;
struct InfResult;

#endif
