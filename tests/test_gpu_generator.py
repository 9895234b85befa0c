"""Device neighbour-index generator (wd_targets_from_coords) against the numpy generator,
which tests/test_synth_and_generator.py pins to the reference's prepare_cluster_indexes.py
output (sha256) - sampled centres and the all-centres mode."""

import numpy as np
import pytest

from helpers import fixture_targets
from well_duplicates_amd import cluster_indexes, synth, workload
from well_duplicates_amd.scanner import Scanner

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    s = Scanner(0)
    yield s
    s.close()


def test_sampled_centres_equal_reference_targets_file(sc):
    """Same centres as `prepare_cluster_indexes.py -n 160 -s 13` on the 150 x 173 honeycomb:
    the device CSR equals the committed reference-generated targets file."""
    _, (centre, lvl_off, nbr) = fixture_targets("mid")
    x, y = synth.honeycomb_pixels(150, 173)
    T, P = sc.targets_from_coords(x, y, centre, levels=5)
    c2, o2, n2 = sc.get_targets()
    assert T == 160 and P == nbr.shape[0]
    assert (c2 == centre).all() and (o2 == lvl_off).all() and (n2 == nbr).all()


@pytest.mark.parametrize("rows,cols,levels", [(40, 60, 3), (25, 31, 7), (64, 1571, 5)])
def test_all_centres_equal_numpy_generator(sc, rows, cols, levels):
    x, y = synth.honeycomb_pixels(rows, cols)
    n = rows * cols
    step = 1 if n < 5000 else 97          # the full-width geometry: every 97th well on the host
    centres = np.arange(0, n, step, dtype=np.int32)
    want = workload.targets_to_csr(cluster_indexes.generate(x, y, centres.tolist(), levels))
    if step == 1:
        T, P = sc.targets_from_coords(x, y, None, levels=levels)
    else:
        T, P = sc.targets_from_coords(x, y, centres, levels=levels)
    got = sc.get_targets()
    assert T == centres.shape[0] and P == want[2].shape[0]
    for a, b in zip(got, want):
        assert (a == b).all()


def test_window_limit_and_errors(sc):
    # wells further than 20000 records away are never neighbours, however close in pixels
    n = 50000
    x = np.full(n, 1000, np.int32)
    y = (1000 + (np.arange(n) % 5) * 10).astype(np.int32)      # everything within 40 px of everything
    centres = np.array([0, 25000, n - 1], np.int32)
    want = workload.targets_to_csr(cluster_indexes.generate(x, y, centres.tolist(), 2, [1, 22, 42]))
    with pytest.raises(RuntimeError):       # > 2048 wells inside the outer ring: refused, not truncated
        sc.targets_from_coords(x, y, centres, levels=2, max_dists=[1, 22, 42])
    assert want[2].shape[0] > 2048
    # a lone well has no neighbours: RuntimeError as the reference (:70-76)
    with pytest.raises(RuntimeError):
        sc.targets_from_coords(np.array([1000], np.int32), np.array([1000], np.int32), None, levels=1)
    with pytest.raises(IndexError):
        sc.targets_from_coords(x, y, np.array([n], np.int32), levels=2)
    with pytest.raises(ValueError):
        sc.targets_from_coords(x, y, centres, levels=2, max_dists=[1, 50, 42])


def test_generated_targets_feed_the_scan(sc):
    """Targets made on the device are scanned without a host round trip."""
    from oracle import oracle
    from well_duplicates_amd.scanner import TileBatch, INVALID_TARGET
    rows, cols, levels, L = 30, 40, 3, 40
    n = rows * cols
    x, y = synth.honeycomb_pixels(rows, cols)
    sc.targets_from_coords(x, y, None, levels=levels)
    centre, lvl_off, nbr = sc.get_targets()
    spec = synth.SynthSpec(seed=5, n_clusters=n, row=cols, plant_per_64k=5000, plant_far=True)
    tb = TileBatch(sc, 1, L, n)
    tb.fill_synthetic(spec, [(1, 1101)], list(range(L)))
    blocks, pt = tb.count(0, 0, per_target=True)
    planes = [synth.plane_bytes(spec, 1, 1101, c) for c in range(L)]
    filt = synth.filter_bytes(spec, 1, 1101)
    valid, dups, lens, _ = oracle.count_tile(planes, filt, centre, lvl_off, nbr, 0, 0)
    got = pt[0].astype(np.int64)
    got[got == INVALID_TARGET] = -1
    assert (got == np.where(valid[:, None] == 1, dups, -1)).all()
    tb.free()


def test_full_size_all_centres_dense_path_equals_queue_kernel(sc):
    """BASELINE configs[4] at full tile size (2743 x 1571 wells, every well a centre, 3 levels):
    the dense path (signatures -> pairs -> verify, packed and unpacked) and the wave-per-target
    queue kernel - which the golden fixtures pin - give the same per-target counts and tallies."""
    from well_duplicates_amd.scanner import TileBatch
    rows, cols, levels, L = workload.HISEQ4000_ROWS, workload.HISEQ4000_COLS, 3, 24
    n = rows * cols
    x, y = synth.honeycomb_pixels(rows, cols)
    T, P = sc.targets_from_coords(x, y, None, levels=levels)
    assert T == n and 35 * n < P < 36 * n + 1
    spec = synth.SynthSpec(seed=21, n_clusters=n, row=cols, plant_per_64k=700, nocall_per_64k=300)
    tb = TileBatch(sc, 2, L, n)
    tb.fill_synthetic(spec, [(1, 1101), (1, 2228)], list(range(L)))
    try:
        sc.set_option("dense_kernel", 0)
        want_b, want_pt = tb.count(0, 0, per_target=True)
        for pack in (0, 1):
            sc.set_option("dense_kernel", 1)
            sc.set_option("dense_pack", pack)
            got_b, got_pt = tb.count(0, 0, per_target=True)
            assert (got_b == want_b).all(), pack
            assert (got_pt == want_pt).all(), pack
        assert want_b[:, 1 + levels:1 + 2 * levels].sum() > 10000          # duplicates were found
        for mode, k in ((1, 1), (2, 2)):                                    # automatic choices
            sc.set_option("dense_kernel", 0)
            want_b1, want_pt1 = tb.count(mode, k, per_target=True)
            sc.set_option("dense_kernel", -1)
            sc.set_option("dense_pack", -1)
            got_b1, got_pt1 = tb.count(mode, k, per_target=True)
            assert (got_b1 == want_b1).all() and (got_pt1 == want_pt1).all(), (mode, k)
    finally:
        sc.set_option("dense_kernel", -1)
        sc.set_option("dense_pack", -1)
        tb.free()
