"""End to end on the GPU: a real run directory (.bcl.gz / .filter files written from the
synthetic spec) through this package's count_well_duplicates CLI must print what the
unmodified reference printed for the same run (tests/golden/*.json): stdout byte for byte,
and the stderr duplicate log line for line."""
import io
import os
from contextlib import redirect_stderr, redirect_stdout

import pytest

from helpers import GOLD, load_fixture, run_cycles
from well_duplicates_amd import count_well_duplicates as cwd
from well_duplicates_amd import synth

pytestmark = pytest.mark.gpu


def _run_dir(tmp_path_factory, name):
    fx = load_fixture(name)
    spec = synth.spec_from_dict(fx["spec"])
    d = tmp_path_factory.mktemp(name)
    cycles = sorted({c for run in fx["runs"] for c in run_cycles(run)})
    if fx.get("cbcl") is None:
        synth.write_run_dir(spec, str(d), fx["lanes"], fx["tiles"], cycles)
    else:
        synth.write_run_dir_cbcl(spec, str(d), fx["lanes"], fx["tiles"], cycles, excluded=fx["cbcl"])
    return fx, str(d)


def _cli(fx, run_dir, run, extra=()):
    argv = ["-f", os.path.join(GOLD, fx["targets_file"]), "-n", str(fx["n_targets"]),
            "-l", str(fx["levels"]), "-s", fx.get("stype", "hiseq_4000"), "-r", run_dir,
            "-t", ",".join(fx["tiles"]), "-i", ",".join(str(l) for l in fx["lanes"])]
    argv += run["flags"] + list(extra)
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        assert cwd.main(argv) == 0
    return out.getvalue(), err.getvalue()


@pytest.mark.parametrize("name", ["mid", "mid_subset", "dead_tile", "seven_levels", "far", "novaseq",
                                  "novaseq_all_wells"])
def test_cli_matches_reference(tmp_path_factory, name):
    fx, run_dir = _run_dir(tmp_path_factory, name)
    for run in fx["runs"]:
        out, err = _cli(fx, run_dir, run)
        assert out == run["stdout"], (name, run["flags"])
        keep = ("center seq at", "well seq at", "edit distance:")
        log = [ln for ln in err.splitlines() if ln.startswith(keep)]
        if "-q" in run["flags"]:
            assert err == ""
        else:
            assert log == run["dup_log"], (name, run["flags"])
            assert err.splitlines()[0] == "Reading tile %s in lane %s" % (fx["tiles"][0], fx["lanes"][0])


def test_cli_interleaved_layout(tmp_path_factory):
    """--layout interleaved: the loaders write every plane into its byte lane of a group of four
    cycles, the queue kernels (equality, Hamming, Levenshtein <= 3) scan that layout, the duplicate
    log is gathered from it - same stdout and stderr as the reference."""
    fx, run_dir = _run_dir(tmp_path_factory, "mid")
    for run in fx["runs"]:
        if run["mode"] == "levenshtein" and run["k"] > 3:
            continue
        out, err = _cli(fx, run_dir, run, ["--layout", "interleaved", "--tile-batch", "1"])
        assert out == run["stdout"], run["flags"]
        keep = ("center seq at", "well seq at", "edit distance:")
        if "-q" not in run["flags"]:
            assert [ln for ln in err.splitlines() if ln.startswith(keep)] == run["dup_log"], run["flags"]
    with pytest.raises(SystemExit):                       # the dense path reads planes
        cwd.main(["--all-wells", "-s", "hiseq_4000", "-r", run_dir, "--layout", "interleaved"])


def test_cli_tile_batching_and_errors(tmp_path_factory):
    fx, run_dir = _run_dir(tmp_path_factory, "dead_tile")
    run = fx["runs"][0]
    base, _ = _cli(fx, run_dir, run)
    one_by_one, _ = _cli(fx, run_dir, run, ["--tile-batch", "1", "--threads", "1"])
    assert one_by_one == base == run["stdout"]
    with pytest.raises(AssertionError):                   # -t pattern matching no tile
        cwd.main(["-f", os.path.join(GOLD, fx["targets_file"]), "-s", "hiseq_4000", "-r", run_dir,
                  "-t", "9999", "-i", "1", "-q"])
    with pytest.raises(RuntimeError):                     # tile without files (bcl :131-132)
        cwd.main(["-f", os.path.join(GOLD, fx["targets_file"]), "-s", "hiseq_4000", "-r", run_dir,
                  "-t", "1103", "-i", "1", "-q", "--cycles", "0-50", "-l", "5"])


def test_cli_pipeline_of_batches(tmp_path):
    """Seven tiles in batches of 1, 2 and 3 (the last one short): three batches are on their way at
    any time and a finished batch's buffers are taken over by the next - report and duplicate log
    equal the one-batch, one-at-a-time, host-inflated run (whose path the goldens above pin)."""
    fx = load_fixture("mid")
    spec = synth.spec_from_dict(fx["spec"])
    run = fx["runs"][2]                                    # the reference's default metric, with the duplicate log
    tiles = [str(1101 + i) for i in range(7)]
    synth.write_run_dir(spec, str(tmp_path), [1], tiles, sorted(run_cycles(run)))
    fx7 = dict(fx, tiles=tiles, lanes=[1])
    base = _cli(fx7, str(tmp_path), run, ["--tile-batch", "7", "--serial-ingest", "--host-inflate"])
    assert base[1].count("Reading tile") == 7 and "edit distance:" in base[1]
    for extra in (["--tile-batch", "1"], ["--tile-batch", "2"], ["--tile-batch", "3"], [],
                  ["--tile-batch", "2", "--serial-ingest"], ["--tile-batch", "2", "--host-inflate"]):
        assert _cli(fx7, str(tmp_path), run, extra) == base, extra


def _torchrun(argv, nproc):
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                           "--master-addr", "127.0.0.1", "--master-port", str(port),
                           "-m", "well_duplicates_amd.count_well_duplicates"] + argv,
                          cwd=repo, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)


@pytest.mark.parametrize("name,run_idx,nproc", [("mid", 2, 2), ("mid", 0, 3), ("far", 2, 2), ("mid", 2, 4)])
def test_cli_ranks_share_one_gpu(tmp_path_factory, tmp_path, name, run_idx, nproc):
    """torchrun with 2, 3 and 4 ranks (all on GPU 0, gloo for the collective; four ranks, the launcher and
    the test runner are the six processes this pool lets share one card): the flat (lane, tile)
    list - 2 lanes x 2 tiles for `mid`, so ranks own parts of different lanes - is sharded, ONE
    all-reduce merges the rows, and rank 0 prints the lanes in order: report and duplicate log
    equal the reference's single-process output."""
    fx, run_dir = _run_dir(tmp_path_factory, name)
    run = fx["runs"][run_idx]
    argv = ["-f", os.path.join(GOLD, fx["targets_file"]), "-n", str(fx["n_targets"]),
            "-l", str(fx["levels"]), "-s", fx.get("stype", "hiseq_4000"), "-r", run_dir,
            "-t", ",".join(fx["tiles"]), "-i", ",".join(str(l) for l in fx["lanes"])]
    report_file = str(tmp_path / "report.txt")
    argv += run["flags"] + ["--device", "0", "--dist-backend", "gloo", "-o", report_file]
    res = _torchrun(argv, nproc)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    # (gloo prints connection banners on stdout, hence the report goes to a file here)
    assert open(report_file).read() == run["stdout"]
    keep = ("center seq at", "well seq at", "edit distance:")
    log = [ln for ln in res.stderr.decode().splitlines() if ln.startswith(keep)]
    assert log == run["dup_log"]


def test_cli_ranks_that_own_nothing(tmp_path_factory, tmp_path):
    """Two (lane, tile) items over four ranks: ranks 1 and 3 own nothing (shard_bounds gives them empty
    blocks), as ranks of an eight-GPU node do on a run of a few tiles - they create their context, set the
    targets, skip the scan and take part in the flag, the merge and the gather.  Report and log equal the
    single-process run of the same tiles."""
    fx, run_dir = _run_dir(tmp_path_factory, "mid")
    run = fx["runs"][2]
    one_tile = dict(fx, tiles=fx["tiles"][:1])
    want_out, want_err = _cli(one_tile, run_dir, run)
    argv = ["-f", os.path.join(GOLD, fx["targets_file"]), "-n", str(fx["n_targets"]),
            "-l", str(fx["levels"]), "-s", fx.get("stype", "hiseq_4000"), "-r", run_dir,
            "-t", fx["tiles"][0], "-i", ",".join(str(l) for l in fx["lanes"])]
    report_file = str(tmp_path / "report.txt")
    res = _torchrun(argv + run["flags"] + ["--device", "0", "--dist-backend", "gloo", "-o", report_file], 4)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert open(report_file).read() == want_out
    keep = ("center seq at", "well seq at", "edit distance:")
    assert [ln for ln in res.stderr.decode().splitlines() if ln.startswith(keep)] == \
        [ln for ln in want_err.splitlines() if ln.startswith(keep)]


def test_cli_failing_rank_takes_the_others_down(tmp_path_factory, tmp_path):
    """One rank's input is broken (a cycle file of ITS tile is missing): every rank leaves with
    an error instead of waiting in the collective, and the report is not written."""
    fx, run_dir = _run_dir(tmp_path_factory, "mid")
    run = fx["runs"][0]
    victim = os.path.join(run_dir, "Data", "Intensities", "BaseCalls", "L002", "C7.1")
    gone = [f for f in os.listdir(victim) if "1102" in f]
    assert gone
    for f in gone:
        os.remove(os.path.join(victim, f))
    argv = ["-f", os.path.join(GOLD, fx["targets_file"]), "-n", str(fx["n_targets"]),
            "-l", str(fx["levels"]), "-s", fx.get("stype", "hiseq_4000"), "-r", run_dir,
            "-t", ",".join(fx["tiles"]), "-i", ",".join(str(l) for l in fx["lanes"])]
    report_file = str(tmp_path / "report.txt")
    argv += run["flags"] + ["--device", "0", "--dist-backend", "gloo", "-o", report_file, "-q"]
    res = _torchrun(argv, 2)
    assert res.returncode != 0
    err = res.stderr.decode()
    assert "FileNotFoundError" in err and "another rank failed" in err
    assert not os.path.exists(report_file) or open(report_file).read() == ""


def test_cli_rank_that_fails_in_setup_takes_the_others_down(tmp_path_factory, tmp_path):
    """Rank 1 cannot even create its context (without --device a rank takes GPU LOCAL_RANK, and this
    box has one GPU): it never reaches the lane loop, yet rank 0 must not wait for it in the merge -
    setup and scan feed the same failure flag."""
    import time
    fx, run_dir = _run_dir(tmp_path_factory, "mid")
    run = fx["runs"][0]
    argv = ["-f", os.path.join(GOLD, fx["targets_file"]), "-n", str(fx["n_targets"]),
            "-l", str(fx["levels"]), "-s", fx.get("stype", "hiseq_4000"), "-r", run_dir,
            "-t", ",".join(fx["tiles"]), "-i", ",".join(str(l) for l in fx["lanes"])]
    report_file = str(tmp_path / "report.txt")
    argv += run["flags"] + ["--dist-backend", "gloo", "-o", report_file, "-q"]
    t0 = time.time()
    res = _torchrun(argv, 2)
    assert res.returncode != 0
    assert time.time() - t0 < 120                      # (a rank left waiting would sit out the group's timeout)
    err = res.stderr.decode()
    assert "another rank failed" in err
    assert not os.path.exists(report_file) or open(report_file).read() == ""


def test_cli_rank_without_a_run_directory_takes_the_others_down(tmp_path_factory, tmp_path):
    """Rank 1 is pointed at a run directory that does not exist (WD_TEST_RUN_SUFFIX: a per-rank suffix of
    -r, test only): BCLReader raises on that rank alone, BEFORE anything of the scan - the process group is
    up by then, so the error goes through the ranks' failure flag and both ranks end within seconds, not
    after the rendezvous timeout (bcl_direct_reader.py:59-70)."""
    import time
    fx, run_dir = _run_dir(tmp_path_factory, "mid")
    run = fx["runs"][0]
    argv = ["-f", os.path.join(GOLD, fx["targets_file"]), "-n", str(fx["n_targets"]),
            "-l", str(fx["levels"]), "-s", fx.get("stype", "hiseq_4000"), "-r", run_dir,
            "-t", ",".join(fx["tiles"]), "-i", ",".join(str(l) for l in fx["lanes"])]
    report_file = str(tmp_path / "report.txt")
    argv += run["flags"] + ["--device", "0", "--dist-backend", "gloo", "-o", report_file, "-q"]
    os.environ["WD_TEST_RUN_SUFFIX"] = "1:/no/such/run"
    try:
        t0 = time.time()
        res = _torchrun(argv, 2)
        took = time.time() - t0
    finally:
        del os.environ["WD_TEST_RUN_SUFFIX"]
    assert res.returncode != 0
    assert took < 120, took
    err = res.stderr.decode()
    assert "another rank failed" in err and ("FileNotFoundError" in err or "NotADirectoryError" in err), err[-1500:]
    assert not os.path.exists(report_file) or open(report_file).read() == ""


def test_cli_a_later_lane_that_fails_does_not_cost_an_earlier_lane_its_report(tmp_path_factory):
    """Lanes 1,2,3 asked for on a run that has lanes 1 and 2 (the default -i is 1..8 on every flowcell):
    the pipeline prefetches lane 3's first batch while lane 2 is still scanned, and get_tile raises for
    the missing directory at once - but the reference writes a lane's report before it touches the next
    (count_well_duplicates.py:207-226, :269), so lanes 1 and 2 are reported in full and only then does
    the error of lane 3 surface."""
    fx, run_dir = _run_dir(tmp_path_factory, "mid")
    run = fx["runs"][0]
    assert [str(l) for l in fx["lanes"]] == ["1", "2"]
    argv = ["-f", os.path.join(GOLD, fx["targets_file"]), "-n", str(fx["n_targets"]),
            "-l", str(fx["levels"]), "-s", fx.get("stype", "hiseq_4000"), "-r", run_dir,
            "-t", ",".join(fx["tiles"]), "-i", "1,2,3"] + run["flags"]
    for extra in ([], ["--tile-batch", "1"], ["--serial-ingest"]):
        out, err = io.StringIO(), io.StringIO()
        with redirect_stdout(out), redirect_stderr(err):
            with pytest.raises((FileNotFoundError, RuntimeError)):
                cwd.main(argv + extra)
        assert out.getvalue() == run["stdout"], extra          # both good lanes, byte for byte


def test_cli_missing_cycle_file_is_a_clean_error(tmp_path_factory):
    """A run folder that lacks one cycle's file of one tile: FileNotFoundError as in the reference
    (bcl_direct_reader.py:207-216) - raised only after every loader thread has finished, so
    nothing writes into freed buffers - and the process can go on using the GPU."""
    fx, run_dir = _run_dir(tmp_path_factory, "dead_tile")
    run = fx["runs"][0]
    victim = os.path.join(run_dir, "Data", "Intensities", "BaseCalls", "L001", "C3.1")
    for f in [f for f in os.listdir(victim) if "1102" in f]:
        os.remove(os.path.join(victim, f))
    with pytest.raises(FileNotFoundError):
        _cli(fx, run_dir, run, ["--threads", "8", "--tile-batch", "1"])
    with pytest.raises(FileNotFoundError):
        _cli(fx, run_dir, run, ["--threads", "8"])
    fx2, run_dir2 = _run_dir(tmp_path_factory, "mid_subset")      # the GPU is still fine
    out, _ = _cli(fx2, run_dir2, fx2["runs"][0])
    assert out == fx2["runs"][0]["stdout"]


def test_cli_collective_through_the_library(tmp_path_factory):
    """--dist-backend wd: the counter block is summed by wd_allreduce_counts, the library's own
    RCCL binding (communicator of this one rank; the unique id travels through the process
    group, here a single-process gloo group)."""
    import torch.distributed as tdist
    fx, run_dir = _run_dir(tmp_path_factory, "mid")
    run = fx["runs"][2]
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    from well_duplicates_amd import dist as wdist
    from well_duplicates_amd.scanner import Scanner
    import numpy as np
    tdist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        with Scanner(0) as sc:
            rows = np.arange(4 * 16, dtype=np.int64).reshape(4, 16) * 1234567
            full = wdist.merge_blocks(rows, 4, 0, 1, backend="wd", scanner=sc)
            assert (full == rows).all()
    finally:
        tdist.destroy_process_group()
    out, _ = _cli(fx, run_dir, run)
    assert out == run["stdout"]


def test_cli_all_wells_equals_targets_file_with_every_well(tmp_path):
    """--all-wells (rings generated on the GPU from the run's s.locs, dense scan path) prints the
    report that a targets file listing every well gives through the ordinary path - for
    equality, Hamming and the default Levenshtein <= 2."""
    from well_duplicates_amd import cluster_indexes
    rows, cols, levels, L = 36, 70, 3, 40
    n = rows * cols
    x, y = synth.honeycomb_pixels(rows, cols)
    spec = synth.SynthSpec(seed=33, n_clusters=n, row=cols, plant_per_64k=4000, nocall_per_64k=500)
    run_dir = str(tmp_path / "run")
    synth.write_run_dir(spec, run_dir, [1], ["1101", "1102"], list(range(L)), slocs=synth.slocs_bytes(x, y))
    tfile = str(tmp_path / "all.list")
    with open(tfile, "w") as fh:
        cluster_indexes.write_targets(cluster_indexes.generate(x, y, list(range(n)), levels), fh)
    base = ["-s", "hiseq_x", "-r", run_dir, "-t", "1101,1102", "-i", "1", "-l", str(levels),
            "--cycles", "0-%d" % L, "-q"]
    for metric in (["-e", "0"], ["-e", "1", "--hamming"], ["-e", "2"]):
        outs = []
        for how in (["-f", tfile, "-n", str(n)], ["--all-wells"]):
            out = io.StringIO()
            with redirect_stdout(out):
                assert cwd.main(base + metric + how) == 0
            outs.append(out.getvalue())
        assert outs[0] == outs[1], metric
        assert "Targets: " in outs[0] and "/%d" % n in outs[0]
