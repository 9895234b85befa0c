"""bench.py's own plumbing: the launcher it becomes for `--gpus N` outside torchrun (the shape of
the driver's command for the scaling runs) and the multi-rank path of the timed region - the
per-step slabs, the merge inside the timed region, the two alternating job blocks - rehearsed with
two ranks sharing one GPU over gloo and compared with one-rank runs of the same lanes.

(The reference's only parallelism is one process per lane, Snakefile.count_dups:25, :153-160; the
bench gives every rank one lane.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import REPO

BENCH = os.path.join(REPO, "bench.py")


def _run(args, timeout=900, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e)


def _last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text
    return json.loads(lines[-1])


def test_launcher_command_line():
    """`bench.py --gpus N` without torchrun becomes a launcher: one child, N ranks on this node,
    rendezvous on 127.0.0.1 - and it needs no GPU to say so (this test runs in the CPU container)."""
    p = _run(["--gpus", "8", "--steps", "7", "--warmup", "2", "--master-port", "29555", "--dry-run"])
    assert p.returncode == 0, p.stderr
    cmd = _last_json(p.stdout)["launcher"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29555"
    script = cmd.index(BENCH)
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "7", "--warmup", "2", "--master-port", "29555"]


def test_no_launcher_for_one_gpu_or_under_torchrun():
    p = _run(["--gpus", "1", "--dry-run"])
    assert p.returncode == 0 and _last_json(p.stdout)["launcher"] is None
    p = _run(["--gpus", "4", "--dry-run"], env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 0 and _last_json(p.stdout)["launcher"] is None


def test_launcher_passes_the_childs_exit_code(tmp_path):
    """The launcher leaves with its child's exit code: a bad flag in the ranks' argv fails the
    torchrun child (argparse, before any GPU call), and `bench.py --gpus 2` reports that."""
    p = _run(["--gpus", "2", "--mode", "eq", "--steps", "1", "--option", "malformed"], timeout=300)
    assert p.returncode != 0


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_match_one_rank_runs(tmp_path):
    """`python bench.py --gpus 2 --backend gloo` starts its own two ranks (both on GPU 0), times the
    steps with the merge inside, and its merged block holds lane 1's rows from rank 0 and lane 2's
    from rank 1: the same rows two one-rank runs of those lanes produce."""
    common = ["--tiles", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--profile-steps", "0"]
    two = str(tmp_path / "two.npy")
    p = _run(["--gpus", "2", "--backend", "gloo"] + common + ["--dump-block", two])
    assert p.returncode == 0, p.stderr[-4000:]
    line = _last_json(p.stdout)
    assert line["n_gpus"] == 2 and line["rccl_world"] == 2
    pg = line["config"]["process_group"]
    assert pg["allreduce_of_ones"] == 2 and pg["backend"] == "gloo" and pg["launched_by"] == "bench.py"
    assert sorted(r["rank"] for r in pg["ranks"]) == [0, 1]
    assert line["config"]["merge"].startswith("one int64 all-reduce per 3 step")
    merged = np.load(two)
    assert merged.shape == (8, 26)
    singles = []
    for lane in (1, 2):
        path = str(tmp_path / ("one_%d.npy" % lane))
        q = _run(["--gpus", "1", "--first-lane", str(lane)] + common + ["--dump-block", path])
        assert q.returncode == 0, q.stderr[-4000:]
        one = _last_json(q.stdout)
        assert one["n_gpus"] == 1 and one["rccl_world"] is None
        singles.append(np.load(path))
    assert (merged[:4] == singles[0]).all() and (merged[4:] == singles[1]).all()
    assert line["config"]["compares_per_step"] == int(merged[:, 1:6].sum())
    assert (merged[:4] != merged[4:]).any()              # two different lanes, not one lane twice
    # the merge-after-every-step form alternates the two job blocks: same rows
    three = str(tmp_path / "three.npy")
    r = _run(["--gpus", "2", "--backend", "gloo", "--merge-every", "1"] + common + ["--dump-block", three])
    assert r.returncode == 0, r.stderr[-4000:]
    assert (np.load(three) == merged).all()


@pytest.mark.gpu
def test_configs2_at_its_own_shape_on_one_gpu(tmp_path):
    """BASELINE configs[2] - 8 lanes x 112 tiles (HiSeq 4000), 2500 targets x 5 levels, 50 bp: 193 GB of planes,
    which one MI355X holds - at its OWN shape on the one GPU a lease has:
      (a) one process, eight lanes resident, one scan of 896 tiles per step;
      (b) four ranks x two lanes over gloo: the job blocks, the [896, 26] merge and the memory of a multi-rank
          run at that shape, sixteen tiles sampled over every rank's rows checked against the oracle;
    Both produce the same rows, lane by lane.  Four ranks, not eight: this pool's process guard ends a run in
    which more than six processes have the card open, and the test runner and the launcher count (a run with
    six ranks was ended that way).  The eight-rank process layout itself - a job per lane, as the reference
    runs it: Snakefile.count_dups:25, :153-160 - is rehearsed on the CPU
    (tests/test_dist.py::test_world_of_eight_at_the_configs2_shape)."""
    common = ["--stype", "hiseq_4000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--profile-steps", "0"]
    one = str(tmp_path / "one.npy")
    p = _run(["--gpus", "1", "--lanes-per-rank", "8"] + common + ["--dump-block", one], timeout=1200)
    assert p.returncode == 0, p.stderr[-4000:]
    a = _last_json(p.stdout)
    assert a["config"]["workload"].startswith("8 lane(s) x 112 tiles (hiseq_4000)")
    assert a["config"]["resident_bytes_per_rank"] > 190e9
    whole = np.load(one)
    assert whole.shape == (896, 26) and (whole[:, 0] > 0).all()
    four = str(tmp_path / "four.npy")
    q = _run(["--gpus", "4", "--lanes-per-rank", "2", "--backend", "gloo", "--check-tiles", "16"] + common +
             ["--dump-block", four], timeout=1200)
    assert q.returncode == 0, q.stderr[-4000:]
    b = _last_json(q.stdout)
    assert b["n_gpus"] == 4 and b["rccl_world"] == 4
    assert b["config"]["process_group"]["allreduce_of_ones"] == 4
    assert "BASELINE configs[2]" in b["config"]["workload"]
    chk = b["config"]["checked_against_oracle"]
    assert len(chk["rows"]) == 16 and chk["ranks_covered"] == [0, 1, 2, 3]
    assert (np.load(four) == whole).all()
    assert b["config"]["compares_per_step"] == int(whole[:, 1:6].sum()) == a["config"]["compares_per_step"]
    for lane in range(8):                                     # eight different lanes, not one lane eight times
        for other in range(lane):
            assert (whole[lane * 112:(lane + 1) * 112] != whole[other * 112:(other + 1) * 112]).any()
