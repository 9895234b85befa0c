"""BASELINE configs[3] at size: a NovaSeq-style tile (4 091 904 wells, cbcl_read.py:77-78), 10 000
sampled targets with 7 rings from the device generator (the reference's own generator stops at 5,
prepare_cluster_indexes.py:19), planes fed from .cbcl files with excluded wells through
wd_load_cbcl_tile (bcl_direct_reader.py:255-325).  One whole tile against the oracle, the .cbcl
expansion against the host mirror of the reference's reader, and size-independent properties
over every kernel family."""
import os

import numpy as np
import pytest

from helpers import blocks_to_reference
from oracle import oracle
from well_duplicates_amd import bcl, cluster_indexes, synth, workload
from well_duplicates_amd.scanner import INVALID_TARGET, Scanner, TileBatch

pytestmark = pytest.mark.gpu

ROWS, COLS = workload.NOVASEQ_ROWS, workload.NOVASEQ_COLS
N = ROWS * COLS
LEVELS, T, L = 7, 10000, 36
TILES = ["1101", "2678"]                                   # first of the top surface, last of the bottom one


@pytest.fixture(scope="module")
def sc():
    with Scanner(0) as s:
        yield s


@pytest.fixture(scope="module")
def run_dir(tmp_path_factory):
    spec = synth.SynthSpec(seed=3, n_clusters=N, row=COLS)
    d = str(tmp_path_factory.mktemp("novaseq_full"))
    synth.write_run_dir_cbcl(spec, d, [3], TILES, list(range(L)), excluded=True)
    return spec, d


def test_config4_full_size(sc, run_dir):
    spec, d = run_dir
    assert N == 4091904 and len(workload.tiles_for_stype(workload.NOVASEQ_STYPE)) == 936
    x, y = synth.honeycomb_pixels(ROWS, COLS)
    centres = cluster_indexes.sample_centres(N, T, 13)
    n_t, n_p = sc.targets_from_coords(x, y, centres, levels=LEVELS, max_dists=cluster_indexes.max_dists_for(LEVELS))
    centre, lvl_off, nbr = sc.get_targets()
    assert n_t == T and lvl_off.shape == (T, LEVELS + 1) and n_p == nbr.shape[0]
    sizes = lvl_off[:, 1:] - lvl_off[:, :-1]
    assert (sizes > 0).all() and 150 <= np.median(sizes.sum(axis=1)) <= 175     # 6 l wells per ring away from the edges
    # the generator's rings equal the vectorised restatement of prepare_cluster_indexes.py:38-78 on a sample
    some = np.sort(np.random.default_rng(1).choice(T, size=40, replace=False))
    ref = cluster_indexes.generate(x, y, [int(centre[t]) for t in some], LEVELS)
    c2, o2, n2 = workload.targets_to_csr(ref)
    for j, t in enumerate(some):
        assert c2[j] == centre[t]
        assert (n2[o2[j, 0]:o2[j, -1]] == nbr[lvl_off[t, 0]:lvl_off[t, -1]]).all()
        assert (o2[j] - o2[j, 0] == lvl_off[t] - lvl_off[t, 0]).all()

    rd = bcl.BCLReader(d)
    handles = [rd.get_tile(3, t) for t in TILES]
    tb = TileBatch(sc, len(TILES), L, N)
    try:
        for i, h in enumerate(handles):
            assert h.num_clusters == N
            sc.load_filter(h.filter_file, tb.filter_ptr(i), N)
        for i, h in enumerate(handles):
            for c in range(L):
                sc.load_cbcl_tile(h.cbcl_path(c), int(h.tile), tb.filter_ptr(i), N, tb.plane_ptr(i, c))
        # the expansion (nibbles, excluded-wells ranks) equals the host mirror of the reference's reader
        for i, c in ((0, 0), (1, L - 1)):
            got, want = tb.download_plane(i, c), handles[i].read_plane(c)
            assert (np.where(got == 0, 4, got & 3) == np.where(want == 0, 4, want & 3)).all()
        # one whole tile against the oracle, three metrics
        planes = [tb.download_plane(1, c) for c in range(L)]
        filt = tb.download_filter(1)
        results = {}
        for mode, k in ((0, 0), (1, 2), (2, 2)):
            blocks, pt = tb.count(mode, k, per_target=True)
            results[(mode, k)] = (blocks, pt)
            valid, dups, lens, _ = oracle.count_tile(planes, filt, centre, lvl_off, nbr, mode, k)
            got = pt[1].astype(np.int64)
            got[got == INVALID_TARGET] = -1
            assert (got == np.where(valid[:, None] == 1, dups, -1)).all(), (mode, k)
            assert (blocks_to_reference(blocks[1], LEVELS) == oracle.tally_tile(valid, dups, lens)).all()
            # size-independent: Wells = ring sizes of the valid targets, Dups <= Wells, Hit <= Targets,
            # first / last histograms each sum to the targets with a duplicate
            for b in blocks:
                tv = b[0]
                wells, dp, hit = b[1:1 + LEVELS], b[1 + LEVELS:1 + 2 * LEVELS], b[1 + 2 * LEVELS:1 + 3 * LEVELS]
                first, last = b[1 + 3 * LEVELS:1 + 4 * LEVELS], b[1 + 4 * LEVELS:]
                assert 0 < tv <= T and (dp <= wells).all() and (hit <= tv).all() and (hit <= dp).all()
                assert first.sum() == last.sum() <= tv and first.sum() >= hit.max()
            assert blocks[:, 1 + LEVELS:1 + 2 * LEVELS].sum() > 0
        # monotone in the metric: equality <= Hamming <= 2 <= Levenshtein <= 2, per target and ring
        eq, h2, l2 = (results[key][1].astype(np.int64) for key in ((0, 0), (1, 2), (2, 2)))
        assert (eq <= h2).all() and (h2 <= l2).all()
        # every kernel family gives the same answer (168 slots per target: two passes of the queue kernel)
        for opts in ({"queue_kernel": 0}, {"early_exit": 0}, {"targets_per_block": 16}, {"dense_kernel": 1}):
            for name, val in opts.items():
                sc.set_option(name, val)
            try:
                for key in ((0, 0), (2, 2)):
                    blocks, pt = tb.count(key[0], key[1], per_target=True)
                    assert (blocks == results[key][0]).all() and (pt == results[key][1]).all(), (opts, key)
            finally:
                sc.set_option("queue_kernel", 1)
                sc.set_option("early_exit", 1)
                sc.set_option("targets_per_block", 64)
                sc.set_option("dense_kernel", -1)
        # the resident layout the CLI keeps for this configuration (--layout auto): the .cbcl blocks expanded
        # straight into their byte lanes of the interleaved groups - tile 0 block by block
        # (wd_load_cbcl_tile_strided), tile 1 through the GPU decoder (wd_load_cbcl_batch_strided) - and
        # scanned by the line walk a dword at a time (k_scan_lines<.., 4>)
        il = TileBatch(sc, len(TILES), L, N, interleave=4)
        try:
            for i, h in enumerate(handles):
                sc.load_filter(h.filter_file, il.filter_ptr(i), N)
            for c in range(L):
                sc.load_cbcl_tile(handles[0].cbcl_path(c), int(handles[0].tile), il.filter_ptr(0), N, il.plane_ptr(0, c), 4)
            sc.load_cbcl_batch([(handles[1].cbcl_path(c), int(handles[1].tile), il.filter_ptr(1), il.plane_ptr(1, c))
                                for c in range(L)], N, threads=4, well_stride=4)
            for i, c in ((0, 0), (0, 5), (1, 2), (1, L - 1)):
                assert (il.download_plane(i, c) == tb.download_plane(i, c)).all(), (i, c)
            for key in ((0, 0), (1, 2), (2, 2)):
                blocks, pt = il.count(key[0], key[1], per_target=True)
                assert sc.last_kernel().startswith("k_scan_lines<true, 8, -1, 4>" if key[0] == 2 else "k_scan_lines<true, 4, 0, 4>")
                assert (blocks == results[key][0]).all() and (pt == results[key][1]).all(), key
        finally:
            il.free()
    finally:
        tb.free()
