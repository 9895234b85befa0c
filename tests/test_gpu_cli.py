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


def test_cli_two_ranks_one_gpu(tmp_path_factory, tmp_path):
    """torchrun with 2 ranks (both on GPU 0, gloo for the collective): tiles are sharded, the
    merged report and duplicate log equal the reference's single-process output."""
    import subprocess
    import sys
    fx, run_dir = _run_dir(tmp_path_factory, "far")
    run = fx["runs"][2]                                   # Levenshtein <= 2, 3 tiles
    argv = ["-f", os.path.join(GOLD, fx["targets_file"]), "-n", str(fx["n_targets"]),
            "-l", str(fx["levels"]), "-s", fx.get("stype", "hiseq_4000"), "-r", run_dir,
            "-t", ",".join(fx["tiles"]), "-i", ",".join(str(l) for l in fx["lanes"])]
    report_file = str(tmp_path / "report.txt")
    argv += run["flags"] + ["--device", "0", "--dist-backend", "gloo", "-o", report_file]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          "-m", "well_duplicates_amd.count_well_duplicates"] + argv,
                         cwd=repo, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    # (gloo prints connection banners on stdout, hence the report goes to a file here)
    assert open(report_file).read() == run["stdout"]
    keep = ("center seq at", "well seq at", "edit distance:")
    log = [ln for ln in res.stderr.decode().splitlines() if ln.startswith(keep)]
    assert log == run["dup_log"]


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
