"""Synthetic-data spec and the neighbour-index generator (CPU)."""
import hashlib
import io
import json
import os

import numpy as np
import pytest

from helpers import GOLD
from well_duplicates_amd import cluster_indexes, synth


def test_known_answer_bytes():
    """Pins the hash: any change to synth.py (or its HIP twin) must keep these bytes."""
    sp = synth.SynthSpec(seed=3)
    idx = np.array([5, 100, 99999, 4309252])
    assert synth.plane_bytes(sp, 1, 1101, 7, idx).tolist() == [8, 97, 80, 106]
    assert synth.plane_bytes(sp, 1, 1101, 0, np.arange(16)).tolist() == \
        [14, 39, 135, 116, 159, 76, 83, 11, 105, 132, 140, 17, 77, 116, 158, 36]


def test_point_and_plane_generation_agree():
    sp = synth.SynthSpec(seed=9, n_clusters=50000, row=200, plant_per_64k=9000, nocall_per_64k=2000)
    rng = np.random.default_rng(0)
    idx = rng.integers(0, sp.n_clusters, 5000)
    for cyc in (0, 1, 17, 127, 128):
        whole = synth.plane_bytes(sp, 2, 1203, cyc)
        assert whole.shape == (sp.n_clusters,)
        assert (synth.plane_bytes(sp, 2, 1203, cyc, idx) == whole[idx]).all()
    f = synth.filter_bytes(sp, 2, 1203)
    assert (synth.filter_bytes(sp, 2, 1203, idx) == f[idx]).all()
    assert set(np.unique(f)) <= {0, 1}
    assert 0.6 < f.mean() < 0.8


def test_planted_variants_do_what_they_say():
    sp = synth.SynthSpec(seed=4, n_clusters=30000, row=100, plant_per_64k=20000, nocall_per_64k=0)
    L = 130
    planes = np.stack([synth.plane_bytes(sp, 1, 1101, c) for c in range(L)])
    codes = planes & 3
    src, var, s1, s2 = synth.plant_info(sp, 1, 1101, np.arange(sp.n_clusters))
    unplanted = src == np.arange(sp.n_clusters)
    # exact copies of an un-planted source are equal over all cycles
    ex = np.flatnonzero((~unplanted) & (var <= synth.VAR_EXACT_MAX) & unplanted[src])
    assert len(ex) > 100
    assert (codes[:, ex] == codes[:, src[ex]]).all()
    # one-substitution copies differ in at most one cycle, at cycle sub1
    one = np.flatnonzero((var == synth.VAR_SUB1) & unplanted[src])
    diff = codes[:, one] != codes[:, src[one]]
    assert diff.sum(axis=0).max() <= 1
    assert (np.flatnonzero(diff.any(axis=1))[:, None] == s1[one][None, :]).any(axis=1).all()
    # shifted copies: well[c] == source[c + 1]
    sh = np.flatnonzero((var == synth.VAR_SHIFT) & unplanted[src])
    assert len(sh) > 50
    assert (codes[:-1, sh] == codes[1:, src[sh]]).all()


def test_dead_tile_and_noise():
    sp = synth.SynthSpec(seed=1, n_clusters=1000, row=50, dead_tiles=(1102,), filter_noise=True)
    assert (synth.filter_bytes(sp, 1, 1102) & 1).sum() == 0
    assert (synth.filter_bytes(sp, 1, 1101) & 1).sum() > 0
    assert (synth.filter_bytes(sp, 1, 1101) & 2).sum() > 0
    assert synth.spec_from_dict(synth.spec_to_dict(sp)) == sp


def test_slocs_roundtrip(tmp_path):
    x, y = synth.honeycomb_pixels(40, 60)
    p = tmp_path / "s.locs"
    p.write_bytes(synth.slocs_bytes(x, y))
    x2, y2 = cluster_indexes.read_slocs(str(p))
    assert (x2 == x).all() and (y2 == y).all()


def _gen_text(rows, cols, n, seed, levels=5):
    x, y = synth.honeycomb_pixels(rows, cols)
    centres = cluster_indexes.sample_centres(rows * cols, n, seed)
    buf = io.StringIO()
    cluster_indexes.write_targets(cluster_indexes.generate(x, y, centres, levels), buf)
    return buf.getvalue()


def test_generator_matches_reference_output():
    """sha256 of the reference prepare_cluster_indexes.py output (tests/golden/generator.json)
    on three honeycomb geometries, incl. edge targets and a full-width (1571-well) row."""
    gens = json.load(open(os.path.join(GOLD, "generator.json")))
    for g in gens:
        text = _gen_text(g["rows"], g["cols"], g["n"], g["seed"])
        assert text.splitlines()[:12] == g["head"]
        assert hashlib.sha256(text.encode()).hexdigest() == g["sha256"]


def test_generator_reproduces_committed_targets_file():
    want = open(os.path.join(GOLD, "mid.targets.list")).read()
    assert _gen_text(150, 173, 160, 13) == want


def test_generator_ring_shapes_and_errors():
    x, y = synth.honeycomb_pixels(60, 60)
    c = 30 * 60 + 30
    rings = cluster_indexes.rings_for(c, x, y, levels=7)
    assert [len(r) for r in rings][:2] == [6, 12]      # outer rings are ragged
    assert all(len(r) > 0 for r in rings)
    allw = np.concatenate(rings)
    assert len(set(allw.tolist())) == len(allw) and c not in allw
    # a lone well has no neighbours: RuntimeError as the reference (:70-76)
    with pytest.raises(RuntimeError):
        cluster_indexes.rings_for(0, x[:1], y[:1], levels=1)
    assert cluster_indexes.max_dists_for(7) == [1, 22, 42, 62, 82, 102, 122, 142]


def test_cli_main(tmp_path, capsys):
    x, y = synth.honeycomb_pixels(150, 173)
    p = tmp_path / "s.locs"
    p.write_bytes(synth.slocs_bytes(x, y))
    assert cluster_indexes.main(["-f", str(p), "-n", "160", "-s", "13"]) == 0
    out = capsys.readouterr().out
    assert out == open(os.path.join(GOLD, "mid.targets.list")).read()
