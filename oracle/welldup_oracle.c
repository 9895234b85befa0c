/*
 * welldup_oracle.c - CPU restatement of the well-duplicate scan path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under well_duplicates_amd/ may import, link or call
 * this file; it is the checker the HIP path is compared with (tests/, __graft_entry__.smoke,
 * bench.py's cpu_baseline leg).  It restates the reference algorithm in the plainest
 * possible C, one function per reference step, each citing the lines it follows in
 * /root/reference (EdinburghGenomics/well_duplicates):
 *
 *   count_well_duplicates.py:228-265   target -> level -> neighbour loop, dist <= e
 *   count_well_duplicates.py:63-106    per-tile Wells/Dups/Hit/AccO/AccI
 *   bcl_direct_reader.py:181,:352-361  BCL byte -> 'A','C','G','T' or 'N'
 *   bcl_direct_reader.py:195-197,:246  filter flag = byte & 1
 *   bcl_direct_reader.py:186-192       IndexError for indices outside the tile
 *
 * Pinning: the tally arithmetic is pinned by the reference's own known-answer tables
 * (test/test_count_well_duplicates.py:37-91) and the whole path by golden fixtures made
 * by running the unmodified reference in the build container (tools/make_golden.py,
 * tests/golden/).  The two distance functions restate the published definitions of the
 * third-party `Levenshtein` package the reference imports (count_well_duplicates.py:9,
 * :200; no version pinned by the reference): hamming = number of differing positions of
 * two equal-length strings, distance = unit-cost insert/delete/substitute edit distance.
 * The reference has no test that pins those two functions ("parity unpinned" for the
 * third-party arithmetic itself; at -e 0 both reduce to string equality).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define WDO_OK 0
#define WDO_ERR_INDEX (-2)   /* reference: IndexError */
#define WDO_ERR_EMPTY (-3)   /* reference: assert len(well_indices) > 0 */
#define WDO_ERR_ARG (-1)

enum { WDO_MODE_EQ = 0, WDO_MODE_HAMMING = 1, WDO_MODE_LEVENSHTEIN = 2 };

/* bcl_direct_reader.py:181 (initialise to 'N'), :352-361 (non-zero byte -> "ACGT"[b & 3]) */
static void wdo_sequence(const uint8_t *const *planes, int L, int64_t idx, char *out)
{
    static const char bases[4] = { 'A', 'C', 'G', 'T' };
    for (int c = 0; c < L; c++) {
        uint8_t b = planes[c][idx];
        out[c] = 'N';
        if (b)
            out[c] = bases[b & 3];
    }
}

/* Levenshtein.hamming: positions at which two equal-length strings differ. */
int wdo_hamming(const char *a, const char *b, int n)
{
    int d = 0;
    for (int i = 0; i < n; i++)
        d += a[i] != b[i];
    return d;
}

/* Levenshtein.distance: textbook full dynamic programme, two rows. */
int wdo_levenshtein(const char *a, int la, const char *b, int lb)
{
    int *prev = (int *)malloc(sizeof(int) * (size_t)(lb + 1));
    int *cur = (int *)malloc(sizeof(int) * (size_t)(lb + 1));
    for (int j = 0; j <= lb; j++)
        prev[j] = j;
    for (int i = 1; i <= la; i++) {
        cur[0] = i;
        for (int j = 1; j <= lb; j++) {
            int sub = prev[j - 1] + (a[i - 1] != b[j - 1]);
            int del = prev[j] + 1;
            int ins = cur[j - 1] + 1;
            int m = sub < del ? sub : del;
            cur[j] = m < ins ? m : ins;
        }
        int *t = prev; prev = cur; cur = t;
    }
    int d = prev[lb];
    free(prev);
    free(cur);
    return d;
}

static int wdo_distance(int mode, const char *a, const char *b, int L)
{
    switch (mode) {
    case WDO_MODE_EQ: return memcmp(a, b, (size_t)L) != 0;   /* 0 iff equal */
    case WDO_MODE_HAMMING: return wdo_hamming(a, b, L);
    default: return wdo_levenshtein(a, L, b, L);
    }
}

/*
 * One tile of count_well_duplicates.py:228-265.
 *
 * planes[c]  : L pointers to N raw BCL bytes (cycle ranges already concatenated, :238/:251)
 * filter     : N raw .filter bytes
 * centre[T], lvl_off[T*(levels+1)], nbr[P] : targets in file order, CSR of rings 1..levels
 * mode, k    : dup iff distance <= k (mode EQ: iff the strings are equal, k ignored)
 * out_dups   : T*levels; -1 in every level of a target whose centre failed the filter
 *              (:236-237 `continue`: such a target records nothing)
 * out_len    : T*levels ring lengths (len(well_indices), :265), also for invalid targets
 * out_dist   : optional P distances (or -1 when the centre is invalid), for the dup log
 * out_valid  : T flags, 1 = centre passed the filter
 */
int wdo_count_tile(const uint8_t *const *planes, int L, const uint8_t *filter, int64_t N,
                   int T, int levels, const int32_t *centre, const int32_t *lvl_off,
                   const int32_t *nbr, int mode, int k,
                   int32_t *out_dups, int32_t *out_len, int32_t *out_dist,
                   uint8_t *out_valid)
{
    if (L < 0 || T < 0 || levels < 0)
        return WDO_ERR_ARG;
    /* Tile.get_seqs fails fast on any requested index outside the tile (:186-192) */
    for (int t = 0; t < T; t++) {
        if (centre[t] < 0 || centre[t] >= N)
            return WDO_ERR_INDEX;
        for (int p = lvl_off[t * (levels + 1)]; p < lvl_off[t * (levels + 1) + levels]; p++)
            if (nbr[p] < 0 || nbr[p] >= N)
                return WDO_ERR_INDEX;
    }
    char *cseq = (char *)malloc((size_t)L + 1);
    char *wseq = (char *)malloc((size_t)L + 1);
    int rc = WDO_OK;
    for (int t = 0; t < T && rc == WDO_OK; t++) {
        const int32_t *off = lvl_off + (size_t)t * (levels + 1);
        for (int lev = 0; lev < levels; lev++)
            out_len[t * levels + lev] = off[lev + 1] - off[lev];
        out_valid[t] = filter[centre[t]] & 1;
        if (!out_valid[t]) {                                 /* :236-237 */
            for (int lev = 0; lev < levels; lev++)
                out_dups[t * levels + lev] = -1;
            if (out_dist)
                for (int p = off[0]; p < off[levels]; p++)
                    out_dist[p] = -1;
            continue;
        }
        wdo_sequence(planes, L, centre[t], cseq);            /* :238 */
        for (int lev = 0; lev < levels; lev++) {             /* :244 */
            if (off[lev + 1] - off[lev] <= 0) {              /* :249 */
                rc = WDO_ERR_EMPTY;
                break;
            }
            int dups = 0;
            for (int p = off[lev]; p < off[lev + 1]; p++) {  /* :250 */
                wdo_sequence(planes, L, nbr[p], wseq);       /* :251 */
                int dist = wdo_distance(mode, cseq, wseq, L);/* :252 */
                int is_dup = (mode == WDO_MODE_EQ) ? (dist == 0) : (dist <= k);  /* :258 */
                dups += is_dup;
                if (out_dist)
                    out_dist[p] = dist;
            }
            out_dups[t * levels + lev] = dups;               /* :265 */
        }
    }
    free(cseq);
    free(wseq);
    return rc;
}

/*
 * Per-tile tallies, count_well_duplicates.py:63-106, from the per-target records.
 * block: 1 + 5*levels int64 = targets, then wells[], dups[], hits[], acco[], acci[].
 * (AccO / AccI themselves, by the reference's two explicit loops - not histograms.)
 */
void wdo_tally_tile(int T, int levels, const uint8_t *valid, const int32_t *dups,
                    const int32_t *len, int64_t *block)
{
    int64_t *wells = block + 1, *d = wells + levels, *hits = d + levels;
    int64_t *acco = hits + levels, *acci = acco + levels;
    memset(block, 0, sizeof(int64_t) * (size_t)(1 + 5 * levels));
    for (int t = 0; t < T; t++) {
        const int32_t *td = dups + (size_t)t * levels;
        if (!valid[t])
            continue;                                        /* invalid centre: not in the list */
        block[0]++;                                          /* :66 */
        int seen = 0;
        for (int lev = 0; lev < levels; lev++) {             /* :80-84 */
            if (td[lev])
                seen = 1;
            acco[lev] += seen;
        }
        seen = 0;
        for (int lev = levels - 1; lev >= 0; lev--) {        /* :85-89 */
            if (td[lev])
                seen = 1;
            acci[lev] += seen;
        }
        for (int lev = 0; lev < levels; lev++) {             /* :93-95 */
            wells[lev] += len[(size_t)t * levels + lev];
            d[lev] += td[lev];
            hits[lev] += td[lev] != 0;
        }
    }
}

/*
 * Many tiles: count + tally each, tiles spread over `threads` OpenMP threads.  The
 * reference's only parallelism is one process per lane (Snakefile.count_dups:25,
 * :153-160); tiles are independent, so threads-over-tiles is the same decomposition.
 * planes: n_tiles*L pointers (tile-major), filters: n_tiles pointers.
 * blocks: n_tiles rows of wdo_tally_tile output.
 */
int wdo_count_tiles_mt(int n_tiles, const uint8_t *const *planes, int L,
                       const uint8_t *const *filters, int64_t N,
                       int T, int levels, const int32_t *centre, const int32_t *lvl_off,
                       const int32_t *nbr, int mode, int k, int64_t *blocks, int threads)
{
    int rc_all = WDO_OK;
    if (threads < 1)
        threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int i = 0; i < n_tiles; i++) {
        int32_t *dups = (int32_t *)malloc(sizeof(int32_t) * (size_t)T * (size_t)(levels ? levels : 1));
        int32_t *len = (int32_t *)malloc(sizeof(int32_t) * (size_t)T * (size_t)(levels ? levels : 1));
        uint8_t *valid = (uint8_t *)malloc((size_t)(T ? T : 1));
        int rc = wdo_count_tile(planes + (size_t)i * L, L, filters[i], N, T, levels, centre,
                                lvl_off, nbr, mode, k, dups, len, NULL, valid);
        if (rc == WDO_OK)
            wdo_tally_tile(T, levels, valid, dups, len, blocks + (size_t)i * (1 + 5 * levels));
        else {
#pragma omp critical
            rc_all = rc;
        }
        free(dups);
        free(len);
        free(valid);
    }
    return rc_all;
}
