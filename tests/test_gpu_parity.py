"""GPU parity tests: the HIP path, called through the C ABI (libwelldup.so via ctypes),
against the CPU oracle and the golden fixtures made from the unmodified reference.

Bar: bit-exact (integer work).  Everything here needs a real MI355X: `-m gpu`.
"""
import numpy as np
import pytest

from helpers import (FIXTURES, MODE_ID, blocks_to_reference, compact_tile, fixture_targets,
                     load_fixture, run_cycles)
from oracle import oracle
from well_duplicates_amd import cluster_indexes, synth
from well_duplicates_amd.scanner import INVALID_TARGET, Scanner, TileBatch

pytestmark = pytest.mark.gpu

DEFAULT_OPTIONS = (("early_exit", 1), ("batch_first", 4), ("batch_next", 4), ("targets_per_block", 32),
                   ("queue_kernel", 1), ("queue_first", 0), ("dense_kernel", -1), ("dense_pack", -1),
                   ("dense_queue_cap", 0), ("dense_sym", 1), ("line_walk", -1), ("line_pairs", 0))


@pytest.fixture(scope="module")
def sc():
    s = Scanner(0)
    yield s
    s.close()


def device_lane_dupl(pt_tile, lvl_off):
    """uint32 [T, levels] (INVALID rows skipped) + ring sizes -> reference lane_dupl entry."""
    lens = np.diff(lvl_off, axis=1)
    out = []
    for t in range(pt_tile.shape[0]):
        if pt_tile.shape[1] and pt_tile[t, 0] == INVALID_TARGET:
            assert (pt_tile[t] == INVALID_TARGET).all()
            continue
        out.append([[int(pt_tile[t, l]), int(lens[t, l])] for l in range(pt_tile.shape[1])])
    return out


# ---------------------------------------------------------------------------------------
def test_synth_device_matches_numpy(sc):
    for spec, lane, tile in (
            (synth.SynthSpec(seed=3, n_clusters=200003, row=517, plant_per_64k=9000,
                             nocall_per_64k=1500, filter_noise=True), 2, 1203),
            (synth.SynthSpec(seed=1, n_clusters=70001, row=1571), 1, 1101),
            (synth.SynthSpec(seed=8, n_clusters=50021, row=173, plant_per_64k=20000, plant_far=True), 4, 2101),
            (synth.SynthSpec(seed=9, n_clusters=30011, row=173, qual_levels=7), 2, 1105),
            (synth.SynthSpec(seed=5, n_clusters=4096, row=64, dead_tiles=(2205,)), 8, 2205)):
        cycles = [0, 1, 2, 50, 126, 127, 128, 149]
        tb = TileBatch(sc, 1, len(cycles), spec.n_clusters)
        tb.fill_synthetic(spec, [(lane, tile)], cycles)
        for c, cyc in enumerate(cycles):
            assert (tb.download_plane(0, c) == synth.plane_bytes(spec, lane, tile, cyc)).all(), cyc
        assert (tb.download_filter(0) == synth.filter_bytes(spec, lane, tile)).all()
        tb.free()


@pytest.mark.parametrize("walk", ["targets", "lines"])
@pytest.mark.parametrize("name", FIXTURES)
def test_golden_fixtures(sc, name, walk):
    """Every golden run of the reference, reproduced on the device: per-target dup counts
    (the reference's lane_dupl), the per-tile tally block, and the duplicate log - target by target
    (k_scan_q and the kernels behind it) and with the pairs walked in the order of their neighbour
    wells (option line_walk: k_scan_lines, where it applies - equality, Hamming, Levenshtein <= 2)."""
    sc.set_option("line_walk", 1 if walk == "lines" else 0)
    sc.set_option("line_pairs", 700 if walk == "lines" else 0)       # (several blocks even on a fixture's few thousand pairs)
    fx = load_fixture(name)
    spec = synth.spec_from_dict(fx["spec"])
    targets, (centre, lvl_off, nbr) = fixture_targets(name)
    levels = fx["levels"]
    sc.set_targets(centre, lvl_off, nbr)
    slots = [(lane, tile) for lane in fx["lanes"] for tile in fx["tiles"]]
    batches = {}
    for run in fx["runs"]:
        cycles = tuple(run_cycles(run))
        if cycles not in batches:
            tb = TileBatch(sc, len(slots), len(cycles), spec.n_clusters)
            tb.fill_synthetic(spec, [(l, int(t)) for l, t in slots], cycles)
            if fx.get("cbcl"):
                # excluded-wells CBCL: failed wells read as no-calls (bcl_direct_reader.py:303-314)
                for i, (l, t) in enumerate(slots):
                    f = synth.filter_bytes(spec, l, int(t))
                    planes = [np.where(f & 1, synth.plane_bytes(spec, l, int(t), c), 0).astype(np.uint8)
                              for c in cycles]
                    tb.upload_tile(i, planes, f)
            batches[cycles] = tb
        tb = batches[cycles]
        mode, k = MODE_ID[run["mode"]], run["k"]
        sc.hitlog_enable(100000)
        blocks, pt = tb.count(mode, k, per_target=True)
        hits, total = sc.hitlog_fetch(100000)
        sc.hitlog_enable(0)
        want_by_lane = {int(r["lane"]): r["lane_dupl"] for r in run["lanes"]}
        n_dups = 0
        for i, (lane, tile) in enumerate(slots):
            got = device_lane_dupl(pt[i], lvl_off)
            assert got == want_by_lane[lane][tile], (name, run["flags"], lane, tile)
            # tally block vs the oracle's reduction of the same per-target data
            valid = (pt[i][:, 0] != INVALID_TARGET).astype(np.uint8) if levels else None
            d = np.where(pt[i] == INVALID_TARGET, 0, pt[i]).astype(np.int32)
            ref_block = oracle.tally_tile(valid, d, np.diff(lvl_off, axis=1).astype(np.int32))
            assert (blocks_to_reference(blocks[i], levels) == ref_block).all()
            n_dups += int(d.sum())
        # duplicate log: one record per duplicate, with the reference's distances
        assert total == n_dups == len(run["dup_log"]) // 3 or "-q" in run["flags"]
        if "-q" not in run["flags"]:
            want = []
            log = run["dup_log"]
            for j in range(0, len(log), 3):
                c = int(log[j].split(":")[0].split()[-1])
                w = int(log[j + 1].split(":")[0].split()[-1])
                dist = int(log[j + 2].split(":")[1])
                want.append((c, w, dist))
            got = sorted((int(centre[h["target"]]), int(nbr[h["slot"]]), int(h["dist"])) for h in hits)
            assert got == sorted(want)
        if walk == "lines" and (mode != 2 or k == 2) and sc.get_option("line_walk_blocks") > 0:
            assert sc.last_kernel().startswith("k_scan_lines"), sc.last_kernel()      # the path under test ran
    for tb in batches.values():
        tb.free()
    sc.set_option("line_walk", -1)
    sc.set_option("line_pairs", 0)


# ---------------------------------------------------------------------------------------
def _random_case(rng, n_clusters, T, levels, ring=8):
    """Random ragged targets on a small tile (neighbours near the centre so plants hit)."""
    centres = rng.choice(n_clusters, size=T, replace=False)
    lvl_off = np.zeros((T, levels + 1), dtype=np.int32)
    nbr = []
    pos = 0
    for t, c in enumerate(centres):
        lvl_off[t, 0] = pos
        for l in range(levels):
            n = int(rng.integers(1, ring + 1))
            cand = c + rng.integers(-3, 4, size=n) + rng.integers(-2, 3, size=n) * 97
            cand = np.clip(cand, 0, n_clusters - 1)
            nbr.extend(cand.tolist())
            pos += n
            lvl_off[t, l + 1] = pos
    return centres.astype(np.int32), lvl_off, np.asarray(nbr, dtype=np.int32)


@pytest.mark.parametrize("L", [1, 2, 5, 9, 50, 64, 65, 100, 150])
def test_all_modes_vs_oracle(sc, L):
    """Equality, Hamming <= k and Levenshtein <= k for many k (incl. k < 0, k >= L), heavy
    planting so that near-duplicates of every variant occur, ragged rings, repeated wells."""
    rng = np.random.default_rng(100 + L)
    spec = synth.SynthSpec(seed=20 + L, n_clusters=20011, row=97, plant_per_64k=30000,
                           nocall_per_64k=3000, pass_per_64k=52000)
    T, levels = 300, 4
    centre, lvl_off, nbr = _random_case(rng, spec.n_clusters, T, levels)
    sc.set_targets(centre, lvl_off, nbr)
    cycles = list(range(3, 3 + L))
    tb = TileBatch(sc, 2, L, spec.n_clusters)
    tb.fill_synthetic(spec, [(1, 1101), (3, 2210)], cycles)
    host = []
    for i, (lane, tile) in enumerate([(1, 1101), (3, 2210)]):
        host.append(compact_tile(spec, lane, tile, cycles, centre, nbr))
    ks = {0: [0], 1: [-1, 0, 1, 2, 3, 7, L - 1, L, L + 5],
          2: [-1, 0, 1, 2, 3, 4, 5, 6, 7, 9, 12, 13, 16, 17, 18, 19, 25, L - 2, L - 1, L, L + 3]}
    for mode, klist in ks.items():
        for k in klist:
            blocks, pt = tb.count(mode, k, per_target=True)
            for i in range(2):
                planes, filt, c2, n2, _ = host[i]
                valid, dups, lens, _ = oracle.count_tile(planes, filt, c2, lvl_off, n2, mode, k)
                want = np.where(valid[:, None] == 1, dups, -1)
                got = pt[i].astype(np.int64)
                got[got == INVALID_TARGET] = -1
                assert (got == want).all(), (L, mode, k, i)
                assert (blocks_to_reference(blocks[i], levels) == oracle.tally_tile(valid, dups, lens)).all()
    tb.free()


def test_kernel_variants_agree(sc):
    """early_exit on/off, every batch shape, several targets_per_block, strided and
    pointer-table plane layouts: identical counters."""
    rng = np.random.default_rng(5)
    spec = synth.SynthSpec(seed=77, n_clusters=30011, row=97, plant_per_64k=20000, nocall_per_64k=2000)
    T, levels, L = 500, 5, 50
    centre, lvl_off, nbr = _random_case(rng, spec.n_clusters, T, levels, ring=40)
    sc.set_targets(centre, lvl_off, nbr)
    tb = TileBatch(sc, 3, L, spec.n_clusters)
    tb.fill_synthetic(spec, [(1, 1101), (1, 1102), (2, 1101)], list(range(L)))
    base = {}
    for mode, k in ((0, 0), (1, 2), (2, 2), (2, 5)):
        base[(mode, k)] = tb.count(mode, k, per_target=True)
    try:
        # queue kernel (default for equality / Hamming): every first-round depth, tpb
        for qf in (0, 1, 2, 3, 4, 5, 6, 7, 8):
            for tpb in (1, 5, 64):
                sc.set_option("queue_first", qf)
                sc.set_option("targets_per_block", tpb)
                for (mode, k), (bl, pt) in base.items():
                    bl2, pt2 = tb.count(mode, k, per_target=True)
                    assert (bl2 == bl).all() and (pt2 == pt).all(), ("queue", qf, tpb, mode, k)
        sc.set_option("queue_kernel", 0)
        for early in (0, 1):
            for b1, b2 in ((2, 4), (3, 4), (4, 4), (4, 8), (8, 8)):
                for tpb in (1, 4, 7, 64):
                    sc.set_option("early_exit", early)
                    sc.set_option("batch_first", b1)
                    sc.set_option("batch_next", b2)
                    sc.set_option("targets_per_block", tpb)
                    for (mode, k), (bl, pt) in base.items():
                        bl2, pt2 = tb.count(mode, k, per_target=True)
                        assert (bl2 == bl).all() and (pt2 == pt).all(), (early, b1, b2, tpb, mode, k)
        # pointer-table layout: planes in scrambled order with odd alignments
        sc.set_option("early_exit", 1)
        sc.set_option("queue_kernel", 1)
        n = spec.n_clusters
        slab = sc.malloc(3 * L * (n + 13) + 64)
        ptrs = [[0] * L for _ in range(3)]
        order = rng.permutation(3 * L)
        for j, o in enumerate(order):
            i, c = divmod(int(o), L)
            ptrs[i][c] = slab + 1 + j * (n + 13)
            sc.h2d(ptrs[i][c], tb.download_plane(i, c))
        for (mode, k), (bl, pt) in base.items():
            bl2, pt2 = sc.count_tiles(ptrs, tb.filter_ptrs(), n, mode, k, per_target=True)
            assert (bl2 == bl).all() and (pt2 == pt).all()
        sc.free(slab)
    finally:
        for name, v in DEFAULT_OPTIONS:
            sc.set_option(name, v)
    tb.free()


def test_sorted_view_of_the_targets(sc):
    """The queue kernels walk the targets sorted by centre (by column strip, then by well), workgroups
    dealt out so that each XCD takes a contiguous stretch of (tile, chunk) pairs - with a grid that is
    not a multiple of 8 here.  Whatever the order: the same tally blocks, the same per-target counts at
    the targets' positions IN THE FILE, the same hit records (target = index in the file, slot =
    position in the file's neighbour list).  File order, plain well order, strips of several widths;
    targets given as CSR and targets generated on the device for sampled centres."""
    from well_duplicates_amd import cluster_indexes
    rows, cols, levels, L = 150, 700, 4, 30
    n = rows * cols
    x, y = synth.honeycomb_pixels(rows, cols)
    spec = synth.SynthSpec(seed=91, n_clusters=n, row=cols, plant_per_64k=12000, nocall_per_64k=1500)
    tiles = [(1, 1101), (1, 1102), (3, 2103)]
    tb = TileBatch(sc, len(tiles), L, n)
    tb.fill_synthetic(spec, tiles, list(range(L)))
    centres = np.random.default_rng(17).permutation(n)[:1501].astype(np.int32)      # file order = random
    try:
        for how in ("generated", "csr"):
            results = {}
            for sort_targets, strip in ((0, 256), (1, 0), (1, 64), (1, 256), (1, 350)):
                sc.set_option("sort_strip", strip)             # (read when the targets are installed)
                if how == "generated":
                    sc.targets_from_coords(x, y, centres, levels=levels)
                    csr = sc.get_targets()
                else:
                    sc.set_targets(*csr)
                assert (sc.get_targets()[0] == centres).all()              # the file's order is what the caller sees
                sc.set_option("sort_targets", sort_targets)
                for tpb in (64, 7):                                        # 24 x 3 = 72 and 215 x 3 = 645 workgroups
                    sc.set_option("targets_per_block", tpb)
                    for mode, k in ((0, 0), (1, 2), (2, 2), (2, 3)):
                        sc.hitlog_enable(200000)
                        blocks, pt = tb.count(mode, k, per_target=True)
                        hits, total = sc.hitlog_fetch(200000)
                        assert 0 < total <= 200000
                        hits = np.sort(hits, order=["tile", "target", "slot"])
                        key = (mode, k)
                        if key not in results:
                            results[key] = (blocks, pt, hits)
                            # the file-order run against the oracle: one tile, per target
                            if mode == 2 and k == 2:
                                planes = [synth.plane_bytes(spec, 1, 1102, c) for c in range(L)]
                                valid, dups, lens, _ = oracle.count_tile(planes, synth.filter_bytes(spec, 1, 1102), *csr, mode, k)
                                got = pt[1].astype(np.int64)
                                got[got == INVALID_TARGET] = -1
                                assert (got == np.where(valid[:, None] == 1, dups, -1)).all()
                        else:
                            b0, p0, h0 = results[key]
                            assert (blocks == b0).all() and (pt == p0).all(), (how, sort_targets, strip, tpb, mode, k)
                            assert (hits == h0).all(), (how, sort_targets, strip, tpb, mode, k)
    finally:
        sc.hitlog_enable(0)
        sc.set_option("sort_targets", 1)
        sc.set_option("sort_strip", 256)
        for name, v in DEFAULT_OPTIONS:
            sc.set_option(name, v)
        tb.free()


@pytest.mark.parametrize("kind", ["all_nocall", "all_copies"])
def test_low_diversity_stress(sc, kind):
    """Worst case for the early exit: (nearly) every neighbour equals its centre, so nothing
    dies early, the survivor queues overflow and drain constantly, and targets with up to
    4 passes (and > 508 slots: fallback kernel) occur.  Still bit-exact."""
    rng = np.random.default_rng(11)
    if kind == "all_nocall":
        spec = synth.SynthSpec(seed=3, n_clusters=9001, row=97, nocall_per_64k=65536)
    else:
        spec = synth.SynthSpec(seed=3, n_clusters=9001, row=1, plant_per_64k=60000, nocall_per_64k=500)
    L = 23
    for ring, levels, T in ((60, 5, 150), (126, 4, 80), (200, 3, 40)):
        centre, lvl_off, nbr = _random_case(rng, spec.n_clusters, T, levels, ring=ring)
        sc.set_targets(centre, lvl_off, nbr)
        tb = TileBatch(sc, 2, L, spec.n_clusters)
        tb.fill_synthetic(spec, [(1, 1101), (2, 1102)], list(range(L)))
        slots_max = int((lvl_off[:, -1] - lvl_off[:, 0]).max())
        il = TileBatch(sc, 2, L, spec.n_clusters, interleave=4)
        il.fill_synthetic(spec, [(1, 1101), (2, 1102)], list(range(L)))
        for mode, k in ((0, 0), (1, 1), (1, 3), (2, 2)):
            res = {}
            for q in (0, 1):
                sc.set_option("queue_kernel", q)
                res[q] = tb.count(mode, k, per_target=True)
            sc.set_option("queue_kernel", 1)
            assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all()
            # the line walk on the same input, both layouts: steps whose pairs nearly all survive the first
            # round are finished in place (kLwInPlace), the rest through the queue and its drains
            try:
                sc.set_option("line_walk", 1)
                for pairs in (0, 300):
                    sc.set_option("line_pairs", pairs)
                    for batch in (tb, il):
                        if batch is il and mode == 2 and k > 3:
                            continue
                        got = batch.count(mode, k, per_target=True)
                        assert sc.last_kernel().startswith("k_scan_lines"), sc.last_kernel()
                        assert (got[0] == res[1][0]).all() and (got[1] == res[1][1]).all(), (kind, ring, mode, k, pairs)
                sc.set_option("line_walk", 0)
                if slots_max <= 508:
                    got = il.count(mode, k, per_target=True)           # k_scan_q on the interleaved layout
                    assert (got[0] == res[1][0]).all() and (got[1] == res[1][1]).all(), (kind, ring, mode, k)
            finally:
                sc.set_option("line_walk", -1)
                sc.set_option("line_pairs", 0)
            for i, (lane, tile) in enumerate([(1, 1101), (2, 1102)]):
                planes, filt, c2, n2, _ = compact_tile(spec, lane, tile, list(range(L)), centre, nbr)
                valid, dups, lens, _ = oracle.count_tile(planes, filt, c2, lvl_off, n2, mode, k)
                want = np.where(valid[:, None] == 1, dups, -1)
                got = res[1][1][i].astype(np.int64)
                got[got == INVALID_TARGET] = -1
                assert (got == want).all(), (kind, ring, mode, k)
                assert (blocks_to_reference(res[1][0][i], levels) == oracle.tally_tile(valid, dups, lens)).all()
            if kind == "all_nocall" and mode == 0:
                assert res[1][0][:, 1 + levels:1 + 2 * levels].sum() == res[1][0][:, 1:1 + levels].sum()  # all dups
        tb.free()
        il.free()


def _mutate(rng, seq, n_edits):
    """n_edits random edits (sub / ins / del), then trimmed or padded back to len(seq)."""
    s = list(seq)
    for _ in range(n_edits):
        op = rng.integers(0, 3)
        pos = int(rng.integers(0, len(s) + 1))
        if op == 0 and s:
            s[min(pos, len(s) - 1)] = int(rng.integers(0, 5))
        elif op == 1:
            s.insert(pos, int(rng.integers(0, 5)))
        elif s:
            del s[min(pos, len(s) - 1)]
    s = s[:len(seq)]
    while len(s) < len(seq):
        s.append(int(rng.integers(0, 5)))
    return s


@pytest.mark.parametrize("L", [7, 20, 50, 64, 101])
def test_levenshtein_crafted_pairs(sc, L):
    """Neighbours are random edit scripts (substitutions, insertions, deletions, no-calls) of
    their centre, 0..7 edits: every band width of the register DP against the oracle's full
    DP, with uploaded (not generated) planes and thresholds around every H boundary."""
    rng = np.random.default_rng(1000 + L)
    T, per = 120, 12
    n = T * (per + 1)
    codes = np.zeros((n, L), dtype=np.int64)          # 0..3 bases, 4 = no-call
    centre = np.arange(T, dtype=np.int32) * (per + 1)
    nbr = []
    for t in range(T):
        c = centre[t]
        codes[c] = rng.integers(0, 5, L) if t % 3 else rng.integers(0, 2, L)   # low diversity too
        for j in range(per):
            codes[c + 1 + j] = _mutate(rng, codes[c].tolist(), int(rng.integers(0, 8)))
            nbr.append(c + 1 + j)
    # BCL bytes: no-call = 0, else random quality bits over the base
    q = rng.integers(1, 41, size=codes.shape)
    planes_b = np.where(codes == 4, 0, (q << 2) | (codes & 3)).astype(np.uint8)
    lvl_off = np.zeros((T, 4), dtype=np.int32)
    for t in range(T):
        lvl_off[t] = t * per + np.array([0, 3, 7, per])
    nbr = np.asarray(nbr, dtype=np.int32)
    filt = np.ones(n, dtype=np.uint8)
    sc.set_targets(centre, lvl_off, nbr)
    tb = TileBatch(sc, 1, L, n)
    planes = [np.ascontiguousarray(planes_b[:, c]) for c in range(L)]
    tb.upload_tile(0, planes, filt)
    for mode, ks in ((2, list(range(0, 22)) + [L - 3]), (1, [0, 1, 2, 5, 9])):
        for k in ks:
            _, pt = tb.count(mode, k, per_target=True)
            valid, dups, lens, dist = oracle.count_tile(planes, filt, centre, lvl_off, nbr, mode, k,
                                                        want_dist=True)
            assert (pt[0].astype(np.int64) == dups).all(), (L, mode, k)
    tb.free()


def test_errors_and_edges(sc):
    spec = synth.SynthSpec(seed=2, n_clusters=5000, row=50)
    tb = TileBatch(sc, 1, 4, spec.n_clusters)
    tb.fill_synthetic(spec, [(1, 1101)], [0, 1, 2, 3])
    centre = np.array([10, 20], np.int32)
    # index == N: IndexError (bcl_direct_reader.py:186-192), nothing launched
    sc.set_targets(centre, np.array([[0, 1], [1, 2]], np.int32), np.array([11, 5000], np.int32))
    with pytest.raises(IndexError):
        tb.count(0, 0)
    sc.set_targets(centre, np.array([[0, 1], [1, 2]], np.int32), np.array([11, -1], np.int32))
    with pytest.raises(IndexError):
        tb.count(0, 0)
    # a neighbour slot no target refers to is not range-checked
    sc.set_targets(centre, np.array([[0, 1], [1, 2]], np.int32), np.array([11, 21, 999999], np.int32))
    tb.count(0, 0)
    # empty ring under a valid centre: AssertionError (count_well_duplicates.py:249)
    filt = synth.filter_bytes(spec, 1, 1101)
    good = int(np.flatnonzero(filt & 1)[0])
    bad = int(np.flatnonzero((filt & 1) == 0)[0])
    sc.set_targets(np.array([good], np.int32), np.array([[0, 1, 1]], np.int32), np.array([3], np.int32))
    with pytest.raises(AssertionError):
        tb.count(0, 0)
    # ... but not under an invalid centre (:236-237 skips it first)
    sc.set_targets(np.array([bad], np.int32), np.array([[0, 1, 1]], np.int32), np.array([3], np.int32))
    blocks, pt = tb.count(0, 0, per_target=True)
    assert blocks[0, 0] == 0 and (pt == INVALID_TARGET).all()
    # argument errors
    with pytest.raises(ValueError):
        sc.set_targets(np.array([1], np.int32), np.array([[0, 2, 1]], np.int32), np.array([3, 4], np.int32))
    with pytest.raises(ValueError):
        sc.set_option("no_such_option", 1)
    sc.set_targets(centre, np.array([[0, 1], [1, 2]], np.int32), np.array([11, 21], np.int32))
    with pytest.raises(ValueError):
        tb.count(5, 0)
    # no targets / no tiles / no cycles
    sc.set_targets(np.zeros(0, np.int32), np.zeros((0, 3), np.int32), np.zeros(0, np.int32))
    blocks, _ = tb.count(0, 0)
    assert blocks.shape == (1, 11) and (blocks == 0).all()
    sc.set_targets(centre, np.array([[0, 1], [1, 2]], np.int32), np.array([11, 21], np.int32))
    bl, _ = sc.count_tiles([], [], spec.n_clusters, 0, 0)
    assert bl.shape == (0, 6)
    tb0 = TileBatch(sc, 1, 0, spec.n_clusters)      # L = 0: every pair of empty strings matches
    tb0.fill_synthetic(spec, [(1, 1101)], [])
    bl, pt = tb0.count(1, 0, per_target=True)
    v = filt[centre] & 1
    assert bl[0, 0] == v.sum() and bl[0, 2] == v.sum()      # Dups == Wells == 1 per valid target
    bl, _ = tb0.count(1, -1)
    assert bl[0, 2] == 0
    tb0.free()
    tb.free()


def test_line_walk_limits(sc):
    """Where the line walk (option line_walk, csrc/scan_lines.inc) does not apply it stands aside and the
    queue kernel answers: an empty ring (the reference's assertion must still fire), a target of more than
    4095 slots (the entry's slot field), no cycles; and it does take what it is forced on - a target of 3000
    slots in ten blocks, more targets in a block than its table holds (256), reads shorter than its first
    round."""
    rng = np.random.default_rng(11)
    spec = synth.SynthSpec(seed=6, n_clusters=20011, row=101, plant_per_64k=20000, nocall_per_64k=1000)
    L = 12
    tb = TileBatch(sc, 2, L, spec.n_clusters)
    tiles = [(1, 1101), (2, 1101)]
    tb.fill_synthetic(spec, tiles, list(range(L)))
    filt = synth.filter_bytes(spec, 1, 1101)
    good = int(np.flatnonzero(filt & 1)[0])
    try:
        sc.set_option("line_walk", 1)
        # an empty ring: :249's assertion, from the queue kernel
        sc.set_targets(np.array([good], np.int32), np.array([[0, 1, 1]], np.int32), np.array([3], np.int32))
        with pytest.raises(AssertionError):
            tb.count(0, 0)
        assert sc.get_option("line_walk_blocks") == 0
        # 5000 slots in one target: beyond the entry's slot field
        big = rng.integers(0, spec.n_clusters, 5000).astype(np.int32)
        sc.set_targets(np.array([good], np.int32), np.array([[0, 5000]], np.int32), big)
        want = tb.count(0, 0, per_target=True)
        assert sc.get_option("line_walk_blocks") <= 0 and not sc.last_kernel().startswith("k_scan_lines")
        del want
        # 3000 slots in one target and 700 small targets around it, blocks of 300 pairs; then blocks of 5000
        # pairs, which would hold more than 256 targets each and are cut short
        T = 700
        centre = np.concatenate([[good], rng.choice(np.setdiff1d(np.arange(spec.n_clusters), [good]), T - 1, replace=False)])
        sizes = np.concatenate([[[1500, 1500]], rng.integers(1, 6, size=(T - 1, 2))])
        lvl_off = np.zeros((T, 3), np.int32)
        lvl_off[:, 1:] = np.cumsum(sizes, axis=1)
        lvl_off += np.concatenate([[0], np.cumsum(sizes.sum(axis=1))[:-1]]).astype(np.int32)[:, None]
        nbr = np.clip(np.repeat(centre, sizes.sum(axis=1)) + rng.integers(-150, 151, int(sizes.sum())), 0,
                      spec.n_clusters - 1).astype(np.int32)
        centre = centre.astype(np.int32)
        sc.set_targets(centre, lvl_off, nbr)
        host = [([synth.plane_bytes(spec, lane, tile, c) for c in range(L)], synth.filter_bytes(spec, lane, tile))
                for lane, tile in tiles]
        for mode, k in ((0, 0), (1, 1), (1, 3), (2, 2)):
            for pairs in (300, 5000):
                sc.set_option("line_pairs", pairs)
                blocks, pt = tb.count(mode, k, per_target=True)
                assert sc.last_kernel().startswith("k_scan_lines"), sc.last_kernel()
                for i in range(2):
                    valid, dups, lens, _ = oracle.count_tile(host[i][0], host[i][1], centre, lvl_off, nbr, mode, k)
                    got = pt[i].astype(np.int64)
                    got[got == INVALID_TARGET] = -1
                    assert (got == np.where(valid[:, None] == 1, dups, -1)).all(), (mode, k, pairs)
                    assert (blocks_to_reference(blocks[i], 2) == oracle.tally_tile(valid, dups, lens)).all()
        # reads shorter than the first round (2 cycles against 5 for Levenshtein <= 2, 3 for Hamming <= 1)
        tb2 = TileBatch(sc, 1, 2, spec.n_clusters)
        tb2.fill_synthetic(spec, [(1, 1101)], [0, 1])
        planes2 = [synth.plane_bytes(spec, 1, 1101, c) for c in range(2)]
        for mode, k in ((0, 0), (1, 1), (2, 2)):
            blocks, pt = tb2.count(mode, k, per_target=True)
            assert sc.last_kernel().startswith("k_scan_lines")
            valid, dups, lens, _ = oracle.count_tile(planes2, filt, centre, lvl_off, nbr, mode, k)
            got = pt[0].astype(np.int64)
            got[got == INVALID_TARGET] = -1
            assert (got == np.where(valid[:, None] == 1, dups, -1)).all(), (mode, k)
        tb2.free()
    finally:
        sc.set_option("line_walk", -1)
        sc.set_option("line_pairs", 0)
        tb.free()


@pytest.mark.parametrize("sym", [1, 0])
def test_dense_all_centres_small(sc, sym):
    """BASELINE config 5 in small: every well of a tile is a centre, 3 levels, 150 bp.  sym = 1: the
    neighbour relation is symmetric, so every pair is compared from its lower well only and a duplicate
    recorded at both ends (the default); sym = 0: every pair from both ends, as for sampled targets."""
    from well_duplicates_amd import workload
    sc.set_option("dense_sym", sym)
    rows, cols, levels, L = 40, 60, 3, 150
    n = rows * cols
    x, y = synth.honeycomb_pixels(rows, cols)
    centre, lvl_off, nbr = workload.targets_to_csr(cluster_indexes.generate(x, y, range(n), levels))
    assert centre.shape[0] == n and 30 <= (lvl_off[:, -1] - lvl_off[:, 0]).max() <= 36
    spec = synth.SynthSpec(seed=12, n_clusters=n, row=cols, plant_per_64k=3000, plant_far=True)
    sc.set_targets(centre, lvl_off, nbr)
    tiles = [(1, 1101), (2, 1101)]
    tb = TileBatch(sc, 2, L, n)
    tb.fill_synthetic(spec, tiles, list(range(L)))
    host = [([synth.plane_bytes(spec, lane, tile, c) for c in range(L)],
             synth.filter_bytes(spec, lane, tile)) for lane, tile in tiles]
    for mode, k, dense, pack in ((0, 0, 1, 0), (0, 0, 1, 1), (0, 0, 0, -1), (1, 1, 1, 1), (1, 1, 1, 0),
                                 (1, 2, 1, -1), (1, 2, 1, 1), (1, 3, 1, -1), (2, 2, 1, -1)):
        sc.set_option("dense_kernel", dense)      # lane-per-target path vs the queue kernel
        sc.set_option("dense_pack", pack)         # survivors checked on packed rows / on the planes
        blocks, pt = tb.count(mode, k, per_target=True)
        if dense:
            assert sc.get_option("dense_sym_on") == sym
        sc.set_option("dense_kernel", -1)
        sc.set_option("dense_pack", -1)
        for i, (lane, tile) in enumerate(tiles):
            planes, filt = host[i]
            valid, dups, lens, _ = oracle.count_tile(planes, filt, centre, lvl_off, nbr, mode, k)
            want = np.where(valid[:, None] == 1, dups, -1)
            got = pt[i].astype(np.int64)
            got[got == INVALID_TARGET] = -1
            assert (got == want).all(), (mode, k)
            assert (blocks_to_reference(blocks[i], levels) == oracle.tally_tile(valid, dups, lens)).all()
        assert blocks[:, 1 + levels:1 + 2 * levels].sum() > 0
    # a survivor queue far too small: what does not fit is finished inside the pairs kernel
    want = {}
    for mode, k in ((0, 0), (1, 1), (2, 2)):
        sc.set_option("dense_kernel", 1)
        want[(mode, k)] = tb.count(mode, k, per_target=True)
        sc.set_option("dense_queue_cap", 37)
        blocks, pt = tb.count(mode, k, per_target=True)
        sc.set_option("dense_queue_cap", 0)
        sc.set_option("dense_kernel", -1)
        assert (blocks == want[(mode, k)][0]).all() and (pt == want[(mode, k)][1]).all()
    tb.free()
    sc.set_option("dense_sym", 1)


@pytest.mark.parametrize("sym", [1, 0])
@pytest.mark.parametrize("L", [150, 24, 17])
def test_dense_window_groups(sc, L, sym):
    """A grid wide enough (200 wells per row) that most 64-target groups lie inside one row: the
    dense path scans them through LDS windows of their neighbours' signatures (k_dense_windows,
    k_dense_pairs) and settles the survivors on packed rows of the marked wells only.  Every mode
    and pack setting against the oracle, duplicates planted near and far, per-target counts,
    tallies and the hit log."""
    from well_duplicates_amd import workload
    rows, cols, levels = 30, 200, 3
    n = rows * cols
    x, y = synth.honeycomb_pixels(rows, cols)
    centre, lvl_off, nbr = workload.targets_to_csr(cluster_indexes.generate(x, y, range(n), levels))
    spec = synth.SynthSpec(seed=21 + L, n_clusters=n, row=cols, plant_per_64k=3000, plant_far=True,
                           nocall_per_64k=1500)
    sc.set_option("dense_sym", sym)          # pairs from their lower well only (the default) / from both ends
    sc.set_targets(centre, lvl_off, nbr)
    tiles = [(1, 1101), (2, 1205), (3, 2101)]
    tb = TileBatch(sc, len(tiles), L, n)
    tb.fill_synthetic(spec, tiles, list(range(L)))
    host = [([synth.plane_bytes(spec, lane, tile, c) for c in range(L)],
             synth.filter_bytes(spec, lane, tile)) for lane, tile in tiles]
    try:
        for mode, k, pack in ((0, 0, -1), (0, 0, 0), (1, 1, -1), (1, 2, 1), (1, 2, 0), (2, 2, -1), (2, 2, 0)):
            sc.set_option("dense_kernel", 1)
            sc.set_option("dense_pack", pack)
            sc.hitlog_enable(400000)
            blocks, pt = tb.count(mode, k, per_target=True)
            hits, total = sc.hitlog_fetch(400000)
            sc.hitlog_enable(0)
            groups = (n + 63) // 64
            assert sc.get_option("dense_window_groups") > groups // 4        # the path under test ran
            assert sc.get_option("dense_sym_on") == sym
            want_hits = []
            for i in range(len(tiles)):
                planes, filt = host[i]
                valid, dups, lens, dist = oracle.count_tile(planes, filt, centre, lvl_off, nbr, mode, k, want_dist=True)
                got = pt[i].astype(np.int64)
                got[got == INVALID_TARGET] = -1
                assert (got == np.where(valid[:, None] == 1, dups, -1)).all(), (L, mode, k, pack)
                assert (blocks_to_reference(blocks[i], levels) == oracle.tally_tile(valid, dups, lens)).all()
                kk = 0 if mode == 0 else k
                slots = np.arange(nbr.shape[0])
                tgt = np.searchsorted(lvl_off[:, 0], slots, side="right") - 1
                ok = (valid[tgt] == 1) & (slots < lvl_off[tgt, -1]) & (dist <= kk)
                want_hits += [(i, int(t), int(p), int(dist[p])) for t, p in zip(tgt[ok], slots[ok])]
            assert total == len(want_hits) > 0
            assert sorted((int(h["tile"]), int(h["target"]), int(h["slot"]), int(h["dist"])) for h in hits) \
                == sorted(want_hits)
        # a survivor queue far too small: the overflow paths (finish in place / leave the block to
        # k_dense_verify with every well of the block marked)
        for mode, k in ((0, 0), (1, 2), (2, 2)):
            want = tb.count(mode, k, per_target=True)
            sc.set_option("dense_queue_cap", 3)
            got = tb.count(mode, k, per_target=True)
            sc.set_option("dense_queue_cap", 0)
            assert (got[0] == want[0]).all() and (got[1] == want[1]).all(), (L, mode, k)
    finally:
        sc.set_option("dense_kernel", -1)
        sc.set_option("dense_pack", -1)
        sc.set_option("dense_queue_cap", 0)
        sc.set_option("dense_sym", 1)
        tb.free()


@pytest.mark.parametrize("sym", [1, 0])
def test_dense_tiles_per_wave(sc, sym):
    """`dense_tile_chunk` - the tiles a wave of the compare stage takes one group of targets through, 16 by
    default - against a scan of 19 tiles: a full chunk and a ragged one (16 + 3), and the same tiles in
    chunks of 1, 5 and 8.  Every mode; tiles 0, 15, 16 and 18 (the chunks' ends) against the oracle, all
    19 rows and per-target counts equal whatever the chunk."""
    from well_duplicates_amd import workload
    rows, cols, levels, L = 12, 200, 3, 40
    n = rows * cols
    x, y = synth.honeycomb_pixels(rows, cols)
    centre, lvl_off, nbr = workload.targets_to_csr(cluster_indexes.generate(x, y, range(n), levels))
    spec = synth.SynthSpec(seed=77, n_clusters=n, row=cols, plant_per_64k=4000, plant_far=True, nocall_per_64k=900,
                           pass_per_64k=50000)
    sc.set_option("dense_sym", sym)
    sc.set_targets(centre, lvl_off, nbr)
    tiles = [(1 + i % 4, 1101 + i) for i in range(19)]
    tb = TileBatch(sc, len(tiles), L, n)
    tb.fill_synthetic(spec, tiles, list(range(L)))
    try:
        sc.set_option("dense_kernel", 1)
        assert sc.get_option("dense_tile_chunk") == 16
        for mode, k in ((0, 0), (1, 2), (2, 2)):
            blocks, pt = tb.count(mode, k, per_target=True)
            assert sc.get_option("dense_window_groups") > 0 and sc.get_option("dense_sym_on") == sym
            for i in (0, 15, 16, 18):
                lane, tile = tiles[i]
                planes = [synth.plane_bytes(spec, lane, tile, c) for c in range(L)]
                valid, dups, lens, _ = oracle.count_tile(planes, synth.filter_bytes(spec, lane, tile), centre, lvl_off, nbr, mode, k)
                got = pt[i].astype(np.int64)
                got[got == INVALID_TARGET] = -1
                assert (got == np.where(valid[:, None] == 1, dups, -1)).all(), (mode, k, i)
                assert (blocks_to_reference(blocks[i], levels) == oracle.tally_tile(valid, dups, lens)).all(), (mode, k, i)
            assert int(blocks[:, 1 + levels:1 + 2 * levels].sum()) > 0
            for chunk in (1, 5, 8):
                sc.set_option("dense_tile_chunk", chunk)
                b2, p2 = tb.count(mode, k, per_target=True)
                sc.set_option("dense_tile_chunk", 16)
                assert (b2 == blocks).all() and (p2 == pt).all(), (mode, k, chunk)
    finally:
        sc.set_option("dense_kernel", -1)
        sc.set_option("dense_tile_chunk", 16)
        sc.set_option("dense_sym", 1)
        tb.free()


@pytest.mark.parametrize("L", [1, 3, 5, 6, 9])
def test_dense_short_reads(sc, L):
    """Reads no longer than the 5-cycle signature never reach the verify kernel; low diversity
    (1-3 bases) makes nearly every neighbour a duplicate."""
    rng = np.random.default_rng(L)
    spec = synth.SynthSpec(seed=50 + L, n_clusters=5003, row=71, plant_per_64k=20000, nocall_per_64k=3000)
    T, levels = 900, 3
    centre, lvl_off, nbr = _random_case(rng, spec.n_clusters, T, levels, ring=9)
    sc.set_targets(centre, lvl_off, nbr)
    tb = TileBatch(sc, 1, L, spec.n_clusters)
    tb.fill_synthetic(spec, [(1, 1101)], list(range(L)))
    planes, filt, c2, n2, _ = compact_tile(spec, 1, 1101, list(range(L)), centre, nbr)
    try:
        for mode, k in ((0, 0), (1, 1), (1, 2), (2, 2)):
            sc.set_option("dense_kernel", 1)
            bl, pt = tb.count(mode, k, per_target=True)
            valid, dups, lens, _ = oracle.count_tile(planes, filt, c2, lvl_off, n2, mode, k)
            got = pt[0].astype(np.int64)
            got[got == INVALID_TARGET] = -1
            assert (got == np.where(valid[:, None] == 1, dups, -1)).all(), (L, mode, k)
            assert (blocks_to_reference(bl[0], levels) == oracle.tally_tile(valid, dups, lens)).all()
    finally:
        sc.set_option("dense_kernel", -1)
        tb.free()


def test_dense_far_neighbours(sc):
    """Neighbours further than +-32767 wells from their centre: the dense path's index table falls
    back from int16 offsets to int32 indices."""
    rng = np.random.default_rng(5)
    n, T, levels, L = 200003, 700, 3, 24
    spec = synth.SynthSpec(seed=77, n_clusters=n, row=449, plant_per_64k=0, nocall_per_64k=2000)
    centre = rng.choice(n, size=T, replace=False).astype(np.int32)
    sizes = rng.integers(1, 7, size=(T, levels))
    lvl_off = np.zeros((T, levels + 1), np.int32)
    lvl_off[:, 1:] = np.cumsum(sizes, axis=1)
    lvl_off += np.concatenate([[0], np.cumsum(sizes.sum(axis=1))[:-1]]).astype(np.int32)[:, None]
    nbr = rng.integers(0, n, size=int(sizes.sum())).astype(np.int32)
    assert np.abs(nbr.astype(np.int64) - np.repeat(centre, sizes.sum(axis=1))).max() > 40000
    sc.set_targets(centre, lvl_off, nbr)
    tb = TileBatch(sc, 1, L, n)
    tb.fill_synthetic(spec, [(1, 1101)], list(range(L)))
    # make some far pairs identical: copy the centre's bytes over its first neighbour
    planes = [tb.download_plane(0, c) for c in range(L)]
    first = nbr[lvl_off[::3, 0]]
    for c in range(L):
        planes[c][first] = planes[c][centre[::3]]
        sc.h2d(tb.plane_ptr(0, c), planes[c])
    filt = tb.download_filter(0)
    try:
        for mode, k in ((0, 0), (1, 1)):
            sc.set_option("dense_kernel", 1)
            bl, pt = tb.count(mode, k, per_target=True)
            valid, dups, lens, _ = oracle.count_tile(planes, filt, centre, lvl_off, nbr, mode, k)
            got = pt[0].astype(np.int64)
            got[got == INVALID_TARGET] = -1
            assert (got == np.where(valid[:, None] == 1, dups, -1)).all()
            assert (blocks_to_reference(bl[0], levels) == oracle.tally_tile(valid, dups, lens)).all()
            assert dups[valid == 1].sum() >= (valid[::3] == 1).sum() > 0
    finally:
        sc.set_option("dense_kernel", -1)
        tb.free()


def test_less_travelled_paths(sc):
    """Pointer-table plane layout x {dense kernel, generic Levenshtein}, their hit logs, empty
    rings under the dense kernel, and scans without the per-target output."""
    rng = np.random.default_rng(77)
    spec = synth.SynthSpec(seed=41, n_clusters=8009, row=97, plant_per_64k=25000, plant_far=True,
                           nocall_per_64k=2500)
    T, levels, L = 400, 3, 30
    centre, lvl_off, nbr = _random_case(rng, spec.n_clusters, T, levels, ring=12)
    sc.set_targets(centre, lvl_off, nbr)
    tb = TileBatch(sc, 2, L, spec.n_clusters)
    tiles = [(1, 1101), (1, 1102)]
    tb.fill_synthetic(spec, tiles, list(range(L)))
    n = spec.n_clusters
    slab = sc.malloc(2 * L * (n + 7) + 64)
    ptrs = [[slab + 3 + (i * L + c) * (n + 7) for c in range(L)] for i in range(2)]
    for i in range(2):
        for c in range(L):
            sc.h2d(ptrs[i][c], tb.download_plane(i, c))
    host = [compact_tile(spec, lane, tile, list(range(L)), centre, nbr) for lane, tile in tiles]
    try:
        for mode, k, dense, pack in ((0, 0, 1, 0), (0, 0, 1, 1), (1, 1, 1, 1), (1, 1, 1, 0), (2, 2, 1, -1), (2, 20, -1, -1),
                                     (2, 25, -1, -1)):
            sc.set_option("dense_kernel", dense)
            sc.set_option("dense_pack", pack)
            sc.hitlog_enable(200000)
            bl, pt = sc.count_tiles(ptrs, tb.filter_ptrs(), n, mode, k, per_target=True)
            hits, total = sc.hitlog_fetch(200000)
            sc.hitlog_enable(0)
            bl2, none = sc.count_tiles(ptrs, tb.filter_ptrs(), n, mode, k)     # no per-target output
            assert none is None and (bl2 == bl).all()
            bl3, pt3 = tb.count(mode, k, per_target=True)                       # strided layout
            assert (bl3 == bl).all() and (pt3 == pt).all()
            want_hits = []
            for i in range(2):
                planes, filt, c2, n2, _ = host[i]
                valid, dups, lens, dist = oracle.count_tile(planes, filt, c2, lvl_off, n2, mode, k,
                                                            want_dist=True)
                got = pt[i].astype(np.int64)
                got[got == INVALID_TARGET] = -1
                assert (got == np.where(valid[:, None] == 1, dups, -1)).all(), (mode, k, dense)
                assert (blocks_to_reference(bl[i], levels) == oracle.tally_tile(valid, dups, lens)).all()
                kk = 0 if mode == 0 else k
                for t in range(T):
                    if valid[t]:
                        for p in range(lvl_off[t, 0], lvl_off[t, levels]):
                            if dist[p] <= kk:
                                want_hits.append((i, t, p, int(dist[p])))
            assert total == len(want_hits)
            assert sorted((int(h["tile"]), int(h["target"]), int(h["slot"]), int(h["dist"])) for h in hits) \
                == sorted(want_hits)
        # an empty ring under a valid centre, dense kernel: AssertionError as the reference
        filt = synth.filter_bytes(spec, 1, 1101)
        good = int(np.flatnonzero(filt & 1)[0])
        sc.set_targets(np.array([good, good + 1], np.int32), np.array([[0, 1, 1], [1, 2, 3]], np.int32),
                       np.array([3, 4, 5], np.int32))
        sc.set_option("dense_kernel", 1)
        with pytest.raises(AssertionError):
            tb.count(0, 0)
    finally:
        sc.set_option("dense_kernel", -1)
        sc.set_option("dense_pack", -1)
        sc.free(slab)
        tb.free()


def test_known_answer_pairs_through_the_device(sc):
    """tests/test_oracle_golden.py's hand-checked DNA pairs (the definitions python-Levenshtein
    publishes, pinned there by the package's own doc examples), each as a two-well tile: well 0 is
    the centre, well 1 its only neighbour.  For every threshold the device counts the pair exactly
    when the tabulated distance is within it, and the hit log carries the tabulated distance."""
    from test_oracle_golden import DNA_KNOWN_ANSWERS
    code = {"N": 0, "A": 4 | 0, "C": 4 | 1, "G": 4 | 2, "T": 4 | 3}     # BCL byte: quality bits | base, 0 = no-call
    L = len(DNA_KNOWN_ANSWERS[0][0])
    n_pairs = len(DNA_KNOWN_ANSWERS)
    # one tile holds all pairs: wells 2i (centre) and 2i + 1 (neighbour)
    n = 2 * n_pairs
    planes = np.zeros((L, n), dtype=np.uint8)
    for i, (a, b, _, _) in enumerate(DNA_KNOWN_ANSWERS):
        for c in range(L):
            planes[c, 2 * i] = code[a[c]]
            planes[c, 2 * i + 1] = code[b[c]]
    centre = np.arange(0, n, 2, dtype=np.int32)
    lvl_off = np.stack([np.arange(n_pairs), np.arange(n_pairs) + 1], axis=1).astype(np.int32)
    nbr = np.arange(1, n, 2, dtype=np.int32)
    sc.set_targets(centre, lvl_off, nbr)
    tb = TileBatch(sc, 1, L, n)
    try:
        tb.upload_tile(0, planes, np.ones(n, dtype=np.uint8))
        for mode, col in ((2, 2), (1, 3)):
            for k in range(0, L + 2):
                sc.hitlog_enable(64)
                bl, pt = tb.count(mode, k, per_target=True)
                hits, total = sc.hitlog_fetch(64)
                sc.hitlog_enable(0)
                want = [1 if row[col] <= k else 0 for row in DNA_KNOWN_ANSWERS]
                assert pt[0, :, 0].tolist() == want, (mode, k)
                assert bl[0, 2] == sum(want) == total
                got = {int(h["target"]): int(h["dist"]) for h in hits}
                assert got == {i: row[col] for i, row in enumerate(DNA_KNOWN_ANSWERS) if row[col] <= k}, (mode, k)
        bl, pt = tb.count(0, 0, per_target=True)     # equality = distance 0 under either metric
        assert pt[0, :, 0].tolist() == [1 if row[2] == 0 else 0 for row in DNA_KNOWN_ANSWERS]
    finally:
        tb.free()


@pytest.mark.parametrize("L,k", [(12, 12), (12, 40), (30, 30)])
def test_hit_log_holds_edit_distance_when_threshold_reaches_read_length(sc, L, k):
    """Levenshtein with -e >= read length: every pair counts (the tallies are a Hamming problem),
    but the reference logs each pair's true edit distance (count_well_duplicates.py:252, :262),
    which can be smaller than the number of mismatching positions."""
    rng = np.random.default_rng(L + k)
    spec = synth.SynthSpec(seed=90 + L, n_clusters=4001, row=61, plant_per_64k=20000, nocall_per_64k=3000)
    T, levels = 120, 3
    centre, lvl_off, nbr = _random_case(rng, spec.n_clusters, T, levels, ring=7)
    sc.set_targets(centre, lvl_off, nbr)
    tb = TileBatch(sc, 1, L, spec.n_clusters)
    tb.fill_synthetic(spec, [(1, 1101)], list(range(L)))
    planes, filt, c2, n2, _ = compact_tile(spec, 1, 1101, list(range(L)), centre, nbr)
    try:
        want_bl, want_pt = tb.count(2, k, per_target=True)            # no hit log: the Hamming rewrite
        sc.hitlog_enable(100000)
        bl, pt = tb.count(2, k, per_target=True)
        hits, total = sc.hitlog_fetch(100000)
        sc.hitlog_enable(0)
        assert (bl == want_bl).all() and (pt == want_pt).all()
        valid, dups, lens, dist = oracle.count_tile(planes, filt, c2, lvl_off, n2, 2, k, want_dist=True)
        assert (blocks_to_reference(bl[0], levels) == oracle.tally_tile(valid, dups, lens)).all()
        want = sorted((t, p, int(dist[p])) for t in range(T) if valid[t]
                      for p in range(lvl_off[t, 0], lvl_off[t, levels]))
        assert total == len(want) == int(dups[valid == 1].sum())
        assert sorted((int(h["target"]), int(h["slot"]), int(h["dist"])) for h in hits) == want
        ham = oracle.count_tile(planes, filt, c2, lvl_off, n2, 1, k, want_dist=True)[3]
        assert (dist[[p for _, p, _ in want]] < ham[[p for _, p, _ in want]]).any()      # the case matters
    finally:
        tb.free()


@pytest.mark.parametrize("L", [1, 3, 4, 5, 8, 23, 50])
def test_interleaved_layout_equals_plane_layout(sc, L):
    """The resident layout option (cycles interleaved by four, include/welldup.h wd_interleave4):
    the same tiles stored both ways give the same per-target counts, tallies and hit log;
    kernels that only read planes refuse the layout."""
    rng = np.random.default_rng(400 + L)
    spec = synth.SynthSpec(seed=70 + L, n_clusters=9001, row=97, plant_per_64k=22000, nocall_per_64k=2500)
    T, levels = 500, 4
    centre, lvl_off, nbr = _random_case(rng, spec.n_clusters, T, levels, ring=40)
    sc.set_targets(centre, lvl_off, nbr)
    tiles = [(1, 1101), (3, 2210)]
    plane = TileBatch(sc, len(tiles), L, spec.n_clusters)
    inter = TileBatch(sc, len(tiles), L, spec.n_clusters, interleave=4)
    try:
        plane.fill_synthetic(spec, tiles, list(range(L)))
        inter.fill_synthetic(spec, tiles, list(range(L)))
        for c in range(L):
            assert (inter.download_plane(1, c) == plane.download_plane(1, c)).all()
        host = [compact_tile(spec, lane, tile, list(range(L)), centre, nbr) for lane, tile in tiles]
        for mode, k in ((0, 0), (1, 1), (1, 2), (1, 3), (1, L), (2, 1), (2, 2), (2, 3)):
            sc.set_option("line_walk", 0)
            sc.hitlog_enable(100000)
            want = plane.count(mode, k, per_target=True)
            want_hits, want_total = sc.hitlog_fetch(100000)
            for i, (planes, filt, c2, n2, _) in enumerate(host):          # the plane layout against the oracle
                valid, dups, lens, _ = oracle.count_tile(planes, filt, c2, lvl_off, n2, mode, k)
                ref = want[1][i].astype(np.int64)
                ref[ref == INVALID_TARGET] = -1
                assert (ref == np.where(valid[:, None] == 1, dups, -1)).all(), (L, mode, k, i)
            # the interleaved batch: target by target (k_scan_q<.., 4>) and pair by pair in the order of the
            # neighbour wells (k_scan_lines<.., 4>: one dword per pair and round, two for Levenshtein <= 2)
            for walk in (0, 1):
                sc.set_option("line_walk", walk)
                sc.set_option("line_pairs", 900 if walk else 0)
                got = inter.count(mode, k, per_target=True)
                hits, total = sc.hitlog_fetch(100000)
                assert (got[0] == want[0]).all() and (got[1] == want[1]).all(), (L, mode, k, walk)
                key = lambda h: (int(h["tile"]), int(h["target"]), int(h["slot"]), int(h["dist"]))
                assert total == want_total and sorted(map(key, hits)) == sorted(map(key, want_hits)), (L, mode, k, walk)
                ran = sc.last_kernel()
                lev = mode == 2 and k >= 2 and L >= 2       # (k >= L is a Hamming problem, but the hit log wants true distances)
                if walk and (not lev or k == 2):
                    assert ran.startswith("k_scan_lines<true, 8, -1, 4>" if lev else "k_scan_lines<true, 4, 0, 4>"), (mode, k, ran)
                else:
                    assert ran.startswith("k_scan_q<true") and ran.split(">")[0].endswith(", 4"), (mode, k, walk, ran)
            sc.hitlog_enable(0)
        sc.set_option("line_walk", -1)
        sc.set_option("line_pairs", 0)
        if L >= 5:
            with pytest.raises(RuntimeError):                 # wider bands read planes only
                inter.count(2, 4)
        sc.set_option("dense_kernel", 1)
        with pytest.raises(RuntimeError):
            inter.count(0, 0)
        sc.set_option("dense_kernel", -1)
        with pytest.raises(ValueError):                       # only 1 and 4 exist
            sc.set_option("well_stride", 3)
        if L > 1:
            # pointers of a plane-per-cycle batch do not describe an interleaved one
            sc.set_option("well_stride", 4)
            try:
                with pytest.raises(ValueError):
                    sc.count_tiles(None, plane.filter_ptrs(), spec.n_clusters, 0, 0, tables=plane.tables, L=L)
            finally:
                sc.set_option("well_stride", 1)
    finally:
        sc.hitlog_enable(0)
        sc.set_option("dense_kernel", -1)
        sc.set_option("line_walk", -1)
        sc.set_option("line_pairs", 0)
        plane.free()
        inter.free()


def test_count_tiles_accepts_host_memory(sc):
    """wd_count_tiles with planes / filters in host memory (numpy buffers), alone and mixed with
    device-resident ones, gives what the device-resident batch gives."""
    rng = np.random.default_rng(8)
    spec = synth.SynthSpec(seed=61, n_clusters=6007, row=83, plant_per_64k=20000, nocall_per_64k=2000)
    T, levels, L = 300, 4, 20
    centre, lvl_off, nbr = _random_case(rng, spec.n_clusters, T, levels, ring=10)
    sc.set_targets(centre, lvl_off, nbr)
    tiles = [(1, 1101), (1, 1102), (2, 1101)]
    tb = TileBatch(sc, len(tiles), L, spec.n_clusters)
    tb.fill_synthetic(spec, tiles, list(range(L)))
    try:
        want_b, want_pt = tb.count(1, 1, per_target=True)
        host_planes = [[np.ascontiguousarray(tb.download_plane(i, c)) for c in range(L)] for i in range(len(tiles))]
        host_filt = [np.ascontiguousarray(tb.download_filter(i)) for i in range(len(tiles))]
        ptrs = [[a.ctypes.data for a in row] for row in host_planes]
        fptrs = [a.ctypes.data for a in host_filt]
        got_b, got_pt = sc.count_tiles(ptrs, fptrs, spec.n_clusters, 1, 1, per_target=True)
        assert (got_b == want_b).all() and (got_pt == want_pt).all()
        # tile 1 stays on the device, odd cycles of tile 0 too
        ptrs[1] = [tb.plane_ptr(1, c) for c in range(L)]
        fptrs[1] = tb.filter_ptr(1)
        ptrs[0] = [tb.plane_ptr(0, c) if c & 1 else ptrs[0][c] for c in range(L)]
        got_b, got_pt = sc.count_tiles(ptrs, fptrs, spec.n_clusters, 1, 1, per_target=True)
        assert (got_b == want_b).all() and (got_pt == want_pt).all()
    finally:
        tb.free()


def test_stream_read_probe(sc):
    """wd_stream_read_probe: a plausible rate for a kernel that only reads (between a PCIe link's and the HBM's
    nominal 8 TB/s), whatever the buffer's length modulo the unrolled step; bad arguments refused."""
    for nbytes in (1 << 28, (1 << 26) + 16 * 12345, 4096, 16):
        buf = sc.malloc(nbytes)
        try:
            sc.memset(buf, 7, nbytes)
            gbs = sc.stream_read_gbs(buf, nbytes, passes=3)
            assert gbs > 0
            if nbytes >= (1 << 28):
                assert 500.0 < gbs < 8000.0, gbs
            with pytest.raises(ValueError):
                sc.stream_read_gbs(buf + 4, nbytes - 16 if nbytes > 16 else 16)          # not 16-byte aligned
            with pytest.raises(ValueError):
                sc.stream_read_gbs(buf, nbytes, passes=0)
        finally:
            sc.free(buf)


def test_fresh_context_requires_targets():
    s = Scanner(0)
    with pytest.raises(RuntimeError):
        s.count_tiles([], [], 10, 0, 0)
    s.close()


# ---------------------------------------------------------------------------------------
def test_production_shape_properties(sc):
    """BASELINE config 1/2 shape (HiSeq-4000 tile geometry, 2500 targets, 5 levels, 50 bp)
    on a few tiles: size-independent properties + one tile against the oracle."""
    rows, cols = 2743, 1571
    n = rows * cols
    spec = synth.SynthSpec(seed=2, n_clusters=n, row=cols)
    x, y = synth.honeycomb_pixels(rows, cols)
    centres = cluster_indexes.sample_centres(n, 2500, 13)
    tg = cluster_indexes.generate(x, y, centres, 5)
    centre = np.array([c for c, _ in tg], np.int32)
    lens = np.array([[len(r) for r in rings] for _, rings in tg], np.int32)
    lvl_off = np.zeros((2500, 6), np.int32)
    lvl_off[:, 1:] = np.cumsum(lens, axis=1)
    lvl_off += np.concatenate([[0], np.cumsum(lens.sum(axis=1))[:-1]]).astype(np.int32)[:, None]
    nbr = np.concatenate([np.concatenate(rings) for _, rings in tg]).astype(np.int32)
    sc.set_targets(centre, lvl_off, nbr)
    L, tiles = 50, [(1, 1101), (1, 1102), (1, 2228), (5, 1205)]
    tb = TileBatch(sc, len(tiles), L, n)
    tb.fill_synthetic(spec, tiles, list(range(L)))
    res = {}
    for mode, k in ((0, 0), (1, 2), (2, 2)):
        bl, pt = tb.count(mode, k, per_target=True)
        res[(mode, k)] = bl
        sc.set_option("early_exit", 0)
        bl_full, pt_full = tb.count(mode, k, per_target=True)      # full gather, same answer
        sc.set_option("early_exit", 1)
        assert (bl == bl_full).all() and (pt == pt_full).all()
        bl_again, _ = tb.count(mode, k)                               # idempotent
        assert (bl == bl_again).all()
        for i, (lane, tile) in enumerate(tiles):
            valid = synth.filter_bytes(spec, lane, tile, centre) & 1
            assert bl[i, 0] == valid.sum()
            assert (bl[i, 1:6] == (lens * valid[:, None]).sum(axis=0)).all()      # Wells
            d = np.where(pt[i] == INVALID_TARGET, 0, pt[i]).astype(np.int64)
            assert ((pt[i][:, 0] == INVALID_TARGET) == (valid == 0)).all()
            assert (bl[i, 6:11] == d.sum(axis=0)).all()                            # Dups
            assert (bl[i, 11:16] == (d > 0).sum(axis=0)).all()                     # Hit
            assert bl[i, 16:21].sum() == bl[i, 21:26].sum() == (d.sum(axis=1) > 0).sum()
    # monotone in the threshold: eq <= hamming 2 <= levenshtein 2, level by level
    assert (res[(0, 0)][:, 6:11] <= res[(1, 2)][:, 6:11]).all()
    assert (res[(1, 2)][:, 6:11] <= res[(2, 2)][:, 6:11]).all()
    assert res[(0, 0)][:, 6:11].sum() > 0
    # one whole tile against the oracle (planes generated by numpy, not downloaded)
    planes, filt, c2, n2, _ = compact_tile(spec, 1, 1102, list(range(L)), centre, nbr)
    for (mode, k), bl in res.items():
        valid, dups, ln, _ = oracle.count_tile(planes, filt, c2, lvl_off, n2, mode, k)
        assert (blocks_to_reference(bl[1], 5) == oracle.tally_tile(valid, dups, ln)).all()
    tb.free()
