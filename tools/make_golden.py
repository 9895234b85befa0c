#!/usr/bin/env python3
"""Generate tests/golden/ by running the UNMODIFIED reference in the build container.

Runs only where /root/reference exists (never on the GPU box).  For each case it
  1. materialises a synthetic run directory from a SynthSpec (well_duplicates_amd.synth),
  2. produces a targets file with the reference's own prepare_cluster_indexes.py (or takes
     one of the reference's test fixtures, or - for level counts the reference generator
     cannot emit - this package's generator),
  3. runs the reference's count_well_duplicates.main() in-process, capturing stdout, the
     stderr duplicate log and the `lane_dupl` structure handed to output_writer,
  4. stores spec + arguments + outputs as a small JSON fixture (inputs are regenerated
     from the spec by the tests; only data is committed, never reference source).

The reference imports the third-party `Levenshtein` package, which is not installed here
(ordinary ModuleNotFoundError).  A stand-in module with the package's two published
definitions (hamming = mismatching positions of equal-length strings, distance = unit-cost
edit distance) is written to a temp dir and put on sys.path; DESIGN.md records that the
third-party arithmetic itself is therefore "parity unpinned" (at -e 0 it is plain string
equality and needs no third-party code).

Usage: python tools/make_golden.py [--only NAME] [--keep]
"""
from __future__ import annotations

import argparse
import contextlib
import hashlib
import io
import json
import os
import shutil
import subprocess
import sys
import tempfile


REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from well_duplicates_amd import synth, cluster_indexes  # noqa: E402

LEV_STANDIN = '''\
"""Stand-in for the third-party python-Levenshtein package (golden generation only)."""

def hamming(a, b):
    if len(a) != len(b):
        raise ValueError("hamming: strings of unequal length")
    return sum(1 for x, y in zip(a, b) if x != y)

def distance(a, b):
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        cur = [i] + [0] * len(b)
        for j in range(1, len(b) + 1):
            cur[j] = min(prev[j - 1] + (a[i - 1] != b[j - 1]), prev[j] + 1, cur[j - 1] + 1)
        prev = cur
    return prev[len(b)]
'''


def import_reference(tmp):
    with open(os.path.join(tmp, "Levenshtein.py"), "w") as fh:
        fh.write(LEV_STANDIN)
    sys.path.insert(0, tmp)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import count_well_duplicates as cwd  # noqa
    return cwd


def run_reference_count(cwd, argv):
    """Call the reference main() with argv; returns (stdout, stderr, [lane_dupl per lane], exc)."""
    captured = []
    orig_writer = cwd.output_writer

    def spy(lane, sample_size, lane_dupl, levels=0, verbose=False):
        captured.append({"lane": str(lane), "sample_size": sample_size,
                         "lane_dupl": {t: [[list(x) for x in targ] for targ in v]
                                       for t, v in lane_dupl.items()}})
        return orig_writer(lane, sample_size, lane_dupl, levels=levels, verbose=verbose)

    cwd.output_writer = spy
    # main() rebinds the module-global `log` when -q is given; restore it afterwards
    orig_log = cwd.log
    out, err = io.StringIO(), io.StringIO()
    exc = None
    old_argv = sys.argv
    sys.argv = ["count_well_duplicates.py"] + list(argv)
    try:
        with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
            try:
                cwd.main()
            except Exception as e:  # recorded: some cases pin the reference's error class
                exc = type(e).__name__
    finally:
        sys.argv = old_argv
        cwd.output_writer = orig_writer
        cwd.log = orig_log
    return out.getvalue(), err.getvalue(), captured, exc


def dup_log_lines(stderr_text):
    """The three-line duplicate records (count_well_duplicates.py:260-262) only."""
    keep = ("center seq at", "well seq at", "edit distance:")
    return [ln for ln in stderr_text.splitlines() if ln.startswith(keep)]


def ref_targets(slocs_path, n, seed):
    """Targets text from the reference's own generator (it runs main() at import: subprocess)."""
    res = subprocess.run([sys.executable, os.path.join(REF, "prepare_cluster_indexes.py"),
                          "-f", slocs_path, "-n", str(n), "-s", str(seed)],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    return res.stdout.decode()


def count_dups(runs):
    return sum(t[0] for r in runs for tiles in r["lane_dupl"].values() for targ in tiles for t in targ)


def make_case(cwd, name, spec, geometry, targets_text, lanes, tiles, cycle_ranges, variants,
              n_targets, levels, keep=False, targets_origin="reference", reuse_targets=None,
              stype="hiseq_4000", cbcl=None):
    """Write the run dir once, run every flag variant, store one JSON."""
    tmp = tempfile.mkdtemp(prefix="wd_golden_")
    try:
        run_dir = os.path.join(tmp, "run")
        slocs = None
        if geometry:
            x, y = synth.honeycomb_pixels(geometry["rows"], geometry["cols"])
            slocs = synth.slocs_bytes(x, y)
        all_cycles = sorted({c for a, b in cycle_ranges for c in range(a, b)})
        if cbcl is None:
            synth.write_run_dir(spec, run_dir, lanes, tiles, all_cycles, slocs)
        else:
            synth.write_run_dir_cbcl(spec, run_dir, lanes, tiles, all_cycles, excluded=cbcl, slocs=slocs)
        tpath = os.path.join(tmp, "targets.list")
        with open(tpath, "w") as fh:
            fh.write(targets_text)
        tname = reuse_targets or (name + ".targets.list")
        if not reuse_targets:
            with open(os.path.join(GOLD, tname), "w") as fh:
                fh.write(targets_text)
        runs = []
        for var in variants:
            argv = ["-f", tpath, "-n", str(n_targets), "-l", str(levels), "-s", stype,
                    "-r", run_dir, "-t", ",".join(tiles), "-i", ",".join(str(l) for l in lanes)]
            argv += var["flags"]
            out, err, captured, exc = run_reference_count(cwd, argv)
            runs.append({
                "flags": var["flags"], "mode": var["mode"], "k": var["k"],
                "cycles": var.get("cycles", cycle_ranges[:1]),
                "stdout": out, "dup_log": dup_log_lines(err), "exception": exc,
                "lanes": captured,
            })
            nd = sum(count_dups([c]) for c in captured)
            print("  %-28s flags=%s dups=%d exc=%s" % (name, " ".join(var["flags"]), nd, exc))
        fixture = {
            "name": name,
            "spec": synth.spec_to_dict(spec),
            "geometry": geometry,
            "targets_file": tname,
            "targets_origin": targets_origin,
            "targets_sha256": hashlib.sha256(targets_text.encode()).hexdigest(),
            "lanes": [int(l) for l in lanes], "tiles": list(tiles),
            "n_targets": n_targets, "levels": levels, "stype": stype,
            "cbcl": cbcl,
            "runs": runs,
        }
        with open(os.path.join(GOLD, name + ".json"), "w") as fh:
            json.dump(fixture, fh, separators=(",", ":"))
            fh.write("\n")
    finally:
        if keep:
            print("  kept", tmp)
        else:
            shutil.rmtree(tmp, ignore_errors=True)


MODES = {"eq": 0, "hamming": 1, "levenshtein": 2}


def variant(mode, k, cycles):
    flags = ["--cycles", ",".join("%d-%d" % (a, b) for a, b in cycles), "-e", str(k)]
    if mode == "hamming":
        flags.append("--hamming")
    # -e 0 is string equality under either metric; the device runs it in mode "eq"
    dev_mode = "eq" if k == 0 else mode
    return {"flags": flags, "mode": dev_mode, "k": k, "cycles": [list(c) for c in cycles]}


def report_cases(cwd):
    """Known-answer tables of test/test_count_well_duplicates.py:37-91 through the reference
    output_writer as it is today (4 trailer lines included)."""
    lane_dupl = {"1208": [[(0, 6), (0, 12), (0, 18)], [(2, 6), (1, 10), (0, 12)],
                          [(3, 6), (1, 10), (1, 12)], [(0, 6), (1, 12), (0, 18)]]}
    bad = {"1209": []}
    bad.update(lane_dupl)
    # a case the reference's table lacks (its own FIXME at :76-77): a dup at the last level only
    outer = {"1101": [[(0, 6), (0, 12), (2, 18)], [(0, 6), (1, 12), (0, 18)], [(0, 6), (0, 12), (0, 17)]],
             "1102": [[(1, 6), (0, 11), (1, 18)]]}
    cases = [("full", lane_dupl, dict(verbose=1)), ("badtile_full", bad, dict(verbose=1)),
             ("badtile_brief", bad, dict(verbose=0)), ("levels2", lane_dupl, dict(verbose=1, levels=2)),
             ("empty", {"1222": []}, dict(verbose=1)), ("outer_only", outer, dict(verbose=1)),
             ("outer_levels2", outer, dict(verbose=1, levels=2))]
    out = []
    for nm, ld, kw in cases:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            cwd.output_writer(1, 4, ld, **kw)
        out.append({"name": nm, "lane": 1, "sample_size": 4,
                    "lane_dupl": {t: [[list(x) for x in targ] for targ in v] for t, v in ld.items()},
                    "kwargs": kw, "stdout": buf.getvalue()})
    with open(os.path.join(GOLD, "report_tables.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("  report_tables: %d cases" % len(out))


def find_seed(make_spec, check, start=1, tries=200):
    for s in range(start, start + tries):
        if check(make_spec(s)):
            return s
    raise RuntimeError("no seed found")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only")
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="wd_ref_")
    cwd = import_reference(tmp)
    want = lambda n: not args.only or args.only == n or (args.only == "novaseq" and n.startswith("novaseq"))

    if want("report_tables"):
        report_cases(cwd)

    # reference test data files (data, not source): targets parser fixtures
    for f in ("small.list", "bad1.list", "bad2.list"):
        shutil.copyfile(os.path.join(REF, "test", f), os.path.join(GOLD, f))

    three = lambda cyc: [variant("hamming", 0, cyc), variant("hamming", 2, cyc),
                         variant("levenshtein", 2, cyc), variant("levenshtein", 1, cyc),
                         variant("levenshtein", 0, cyc), variant("hamming", 1, cyc)]

    # --- cfg 1: the reference's small.list, one full-size tile, 50 bp, 2 levels ----------
    if want("small_list"):
        spec = synth.SynthSpec(seed=11, plant_per_64k=26000, pass_per_64k=60000)
        text = open(os.path.join(REF, "test", "small.list")).read()
        make_case(cwd, "small_list", spec, None, text, [1], ["1101"], [(0, 50)],
                  three([(0, 50)]), n_targets=2500, levels=2, keep=args.keep,
                  targets_origin="reference test/small.list")

    # --- mid: reference generator on a small honeycomb, 2 tiles, 5 levels ----------------
    geo = {"rows": 150, "cols": 173}
    n_mid = geo["rows"] * geo["cols"]
    if want("mid"):
        spec = synth.SynthSpec(seed=5, n_clusters=n_mid, row=geo["cols"], plant_per_64k=9000,
                               nocall_per_64k=1300, filter_noise=True)
        tmpd = tempfile.mkdtemp(prefix="wd_slocs_")
        x, y = synth.honeycomb_pixels(geo["rows"], geo["cols"])
        sl = os.path.join(tmpd, "s.locs")
        open(sl, "wb").write(synth.slocs_bytes(x, y))
        text = ref_targets(sl, 160, 13)
        shutil.rmtree(tmpd)
        cyc = [(0, 50)]
        multi = [(3, 20), (40, 61)]
        variants = three(cyc) + [variant("hamming", 0, multi), variant("levenshtein", 2, multi),
                                 variant("levenshtein", 3, cyc), variant("hamming", 5, cyc),
                                 variant("levenshtein", 6, cyc)]
        make_case(cwd, "mid", spec, geo, text, [1, 2], ["1101", "1102"], cyc + multi, variants,
                  n_targets=160, levels=5, keep=args.keep)

    # --- -n / -l subsets of the same targets file, summary-only output --------------------
    if want("mid_subset"):
        spec = synth.SynthSpec(seed=5, n_clusters=n_mid, row=geo["cols"], plant_per_64k=9000,
                               nocall_per_64k=1300, filter_noise=True)
        text = open(os.path.join(GOLD, "mid.targets.list")).read()
        cyc = [(0, 50)]
        v1 = variant("hamming", 0, cyc); v1["flags"] += ["-S"]
        v2 = variant("levenshtein", 2, cyc); v2["flags"] += ["-q"]
        make_case(cwd, "mid_subset", spec, geo, text, [1], ["1101", "1102"], cyc, [v1, v2],
                  n_targets=50, levels=3, keep=args.keep, targets_origin="mid.targets.list",
                  reuse_targets="mid.targets.list")

    # --- a tile whose every centre fails the filter, next to a live one -------------------
    if want("dead_tile"):
        spec = synth.SynthSpec(seed=5, n_clusters=n_mid, row=geo["cols"], plant_per_64k=9000,
                               dead_tiles=(1102,))
        text = open(os.path.join(GOLD, "mid.targets.list")).read()
        cyc = [(0, 50)]
        make_case(cwd, "dead_tile", spec, geo, text, [1], ["1101", "1102"], cyc,
                  [variant("hamming", 0, cyc), variant("levenshtein", 2, cyc)],
                  n_targets=160, levels=5, keep=args.keep, targets_origin="mid.targets.list",
                  reuse_targets="mid.targets.list")

    # --- duplicates at every level 1..5 (copies from up to 5 rows away): AccO/AccI columns ---
    if want("far"):
        spec = synth.SynthSpec(seed=21, n_clusters=n_mid, row=geo["cols"], plant_per_64k=20000,
                               plant_far=True, nocall_per_64k=900)
        text = open(os.path.join(GOLD, "mid.targets.list")).read()
        cyc = [(5, 45)]
        make_case(cwd, "far", spec, geo, text, [4], ["2101", "2102", "2103"], cyc,
                  [variant("hamming", 0, cyc), variant("hamming", 1, cyc),
                   variant("levenshtein", 2, cyc), variant("levenshtein", 3, cyc)],
                  n_targets=160, levels=5, keep=args.keep, targets_origin="mid.targets.list",
                  reuse_targets="mid.targets.list")

    # --- 7 levels: targets from this package's generator (the reference's stops at 5) -----
    if want("seven_levels"):
        spec = synth.SynthSpec(seed=8, n_clusters=n_mid, row=geo["cols"], plant_per_64k=9000)
        x, y = synth.honeycomb_pixels(geo["rows"], geo["cols"])
        import random
        random.seed(13)
        # centres away from the edge so that all 7 rings exist
        cand = [r * geo["cols"] + c for r in range(10, geo["rows"] - 10)
                for c in range(10, geo["cols"] - 10)]
        centres = random.sample(cand, 60)
        buf = io.StringIO()
        cluster_indexes.write_targets(cluster_indexes.generate(x, y, centres, levels=7), buf)
        cyc = [(10, 110)]
        make_case(cwd, "seven_levels", spec, geo, buf.getvalue(), [1], ["1101"], cyc,
                  [variant("hamming", 0, cyc), variant("hamming", 3, cyc),
                   variant("levenshtein", 2, cyc)],
                  n_targets=60, levels=7, keep=args.keep,
                  targets_origin="well_duplicates_amd.cluster_indexes levels=7")

    # --- BASELINE config 4 in small: NovaSeq layout (.cbcl, excluded wells), 7 levels -------
    if want("novaseq"):
        spec = synth.SynthSpec(seed=31, n_clusters=n_mid, row=geo["cols"], plant_per_64k=16000,
                               plant_far=True, nocall_per_64k=1500)
        text = open(os.path.join(GOLD, "seven_levels.targets.list")).read()
        cyc = [(0, 30), (60, 80)]
        for excl, nm in ((True, "novaseq"), (False, "novaseq_all_wells")):
            make_case(cwd, nm, spec, geo, text, [2], ["1101", "1102", "2678"], cyc,
                      [variant("hamming", 0, cyc), variant("levenshtein", 2, cyc)],
                      n_targets=60, levels=7, keep=args.keep,
                      targets_origin="seven_levels.targets.list",
                      reuse_targets="seven_levels.targets.list", stype="2678", cbcl=excl)

    # --- generator parity: reference prepare_cluster_indexes.py output, two geometries ----
    if want("generator"):
        gens = []
        for rows, cols, n, seed in ((150, 173, 160, 13), (64, 1571, 40, 7), (300, 90, 50, 2)):
            tmpd = tempfile.mkdtemp(prefix="wd_slocs_")
            x, y = synth.honeycomb_pixels(rows, cols)
            sl = os.path.join(tmpd, "s.locs")
            open(sl, "wb").write(synth.slocs_bytes(x, y))
            text = ref_targets(sl, n, seed)
            shutil.rmtree(tmpd)
            gens.append({"rows": rows, "cols": cols, "n": n, "seed": seed,
                         "sha256": hashlib.sha256(text.encode()).hexdigest(),
                         "head": text.splitlines()[:12]})
            print("  generator %dx%d n=%d sha=%s" % (rows, cols, n, gens[-1]["sha256"][:12]))
        with open(os.path.join(GOLD, "generator.json"), "w") as fh:
            json.dump(gens, fh, indent=1)

    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
