#!/usr/bin/env python3
"""Counter evidence for one tools/mode_probe.py case: one rocprofv3 run per counter pass
(MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2,
so they cannot share a pass; 8 SQ slots), plus a --kernel-trace --stats pass for the durations,
summarised per kernel into one JSON file.

  python3 tools/pmc_collect.py --case lev2 --out gpurun_out/pmc --tag r02
  python3 tools/pmc_collect.py --case dense_lev2 --probe-args "--tiles 8" --passes kt,fetch,write

The summary holds, per kernel: launches, mean duration (kernel-trace pass) and per-dispatch
counter means (the first dispatch of every kernel is dropped: first-use allocations and cold
tables), and the derived HBM bytes per dispatch: 2 x FETCH_SIZE KiB (gfx950 tallies a 128-byte
request as 64 B; profiles/r01_b_* calibrates that for this access pattern) + WRITE_SIZE KiB.
"""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys

PASSES = {
    "kt": None,
    "fetch": ["FETCH_SIZE", "TCC_EA0_RDREQ_sum"],
    "write": ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
    "sq": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES",
           "SQ_BUSY_CYCLES", "SQ_WAIT_ANY"],
    "ta": ["TA_BUSY_avr", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "GRBM_GUI_ACTIVE"],
    "lds": ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
            "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
}


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name).strip()


def summarise_counters(path, out):
    per = {}
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            key = (short(row["Kernel_Name"]), row["Counter_Name"])
            per.setdefault(key, {}).setdefault(int(row["Dispatch_Id"]), 0.0)
            per[key][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    for (kern, ctr), by_dispatch in per.items():
        vals = [by_dispatch[d] for d in sorted(by_dispatch)]
        if len(vals) > 1:
            vals = vals[1:]
        out.setdefault(kern, {})[ctr] = {"dispatches": len(vals), "mean": sum(vals) / len(vals)}


def summarise_trace(path, out):
    per = {}
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            per.setdefault(short(row["Kernel_Name"]), []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for kern, durs in per.items():
        if len(durs) > 1:
            durs = durs[1:]
        out.setdefault(kern, {})["duration_us"] = {"dispatches": len(durs), "mean": sum(durs) / len(durs) / 1e3,
                                                   "min": min(durs) / 1e3, "max": max(durs) / 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True)
    ap.add_argument("--probe-args", default="")
    ap.add_argument("--passes", default="kt,fetch,write,sq")
    ap.add_argument("--out", default="gpurun_out/pmc")
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--keep", default="k_", help="only kernels whose name contains this")
    a = ap.parse_args()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = "%s_%s%s" % (a.tag, a.case, "_" + re.sub(r"[^0-9A-Za-z]+", "", a.probe_args) if a.probe_args else "")
    summary = {"case": a.case, "probe_args": a.probe_args, "kernels": {}, "probe": {}}
    env = dict(os.environ, TMPDIR="/tmp")
    for p in a.passes.split(","):
        d = os.path.join(a.out, name, p)
        os.makedirs(d, exist_ok=True)
        cmd = ["rocprofv3"]
        cmd += ["--kernel-trace", "--stats"] if PASSES[p] is None else ["--pmc"] + PASSES[p]
        cmd += ["-d", d, "-o", "run", "--output-format", "csv", "--",
                "python3", os.path.join(repo, "tools", "mode_probe.py"), "--case", a.case] + a.probe_args.split()
        print("+", " ".join(cmd), flush=True)
        res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        with open(os.path.join(d, "log.txt"), "w") as fh:
            fh.write(res.stdout + "\n--- stderr ---\n" + res.stderr[-20000:])
        if res.returncode != 0:
            print("pass %s failed (%d): %s" % (p, res.returncode, res.stderr[-2000:]), flush=True)
            summary.setdefault("failed_passes", []).append(p)
            continue
        for line in res.stdout.splitlines():
            if line.startswith("{") and '"case"' in line:
                summary["probe"][p] = json.loads(line)
                # every pass must have profiled the same code: the library's own hashes (wd_build_id)
                for key in ("build_id", "unit", "unit_id", "kernel"):
                    v = summary["probe"][p].get(key)
                    if summary.setdefault(key, v) != v:
                        raise SystemExit("pass %s ran %s = %s, an earlier pass %s" % (p, key, v, summary[key]))
        if PASSES[p] is None:
            for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                summarise_trace(f, summary["kernels"])
        else:
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                summarise_counters(f, summary["kernels"])
    summary["kernels"] = {k: v for k, v in summary["kernels"].items() if a.keep in k and "synth" not in k}
    for kern, c in summary["kernels"].items():
        if "FETCH_SIZE" in c:
            hbm = 2.0 * c["FETCH_SIZE"]["mean"] * 1024.0
            if "WRITE_SIZE" in c:
                hbm += c["WRITE_SIZE"]["mean"] * 1024.0
            c["hbm_bytes_per_dispatch"] = hbm
            if "duration_us" in c:
                c["hbm_tb_per_s"] = hbm / (c["duration_us"]["mean"] * 1e-6) / 1e12
    path = os.path.join(a.out, name + ".json")
    with open(path, "w") as fh:
        json.dump(summary, fh, indent=1, sort_keys=True)
    print("wrote", path)
    for kern, c in sorted(summary["kernels"].items()):
        print("%-48s %s" % (kern[:48], {k: (round(v["mean"], 1) if isinstance(v, dict) else round(v, 3)) for k, v in c.items()}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
