#!/usr/bin/env python3
"""Copies the judged summaries of the last `scripts/gpu_profile.sh` run from gpurun_out/ into profiles/<round>/ and
refreshes profiles/pmc_traffic.json.  Usage: python scripts/collect_profiles.py r03 [gpurun_out/r03p]
(round 3 on: the directory written by scripts/gpu_r03_profiles.sh)"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.chdir(ROOT)
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src_dir = sys.argv[2] if len(sys.argv) > 2 else None
dst = os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)


def latest(pattern):
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1]


if src_dir:
    P = {"s6": f"{src_dir}/prof_1e6", "s7": f"{src_dir}/prof_1e7", "bench": f"{src_dir}/bench_default.json",
         "l6": f"{src_dir}/prof_1e6.log", "l7": f"{src_dir}/prof_1e7.log", "pmc": f"{src_dir}/pmc_"}
    shutil.copy(f"{src_dir}/bench_20steps.json", f"{dst}/bench_20steps.json")
    shutil.copy(f"{src_dir}/bench_default.json", f"{dst}/bench_default.json")
else:
    P = {"s6": "gpurun_out/prof3", "s7": "gpurun_out/prof3_1e7", "bench": "gpurun_out/bench3.json", "l6": "gpurun_out/prof3.log",
         "l7": "gpurun_out/prof3_1e7.log", "pmc": "gpurun_out/pmc_"}
shutil.copy(latest(P["s6"] + "/*/*kernel_stats.csv"), f"{dst}/bench_1e6_kernel_stats.csv")
shutil.copy(latest(P["s7"] + "/*/*kernel_stats.csv"), f"{dst}/bench_1e7_kernel_stats.csv")
shutil.copy(P["bench"], f"{dst}/bench_final.json")
for src, name in ((P["l6"], "bench_1e6_under_rocprof.json"), (P["l7"], "bench_1e7_under_rocprof.json")):
    with open(f"{dst}/{name}", "w") as f:
        f.write(open(src).read().strip().splitlines()[-1] + "\n")
out = {}
for tag in ("fetch_1e6", "write_1e6", "fetch_1e7", "write_1e7"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(latest(f"{P['pmc']}{tag}/*/*counter_collection.csv"))):
        acc[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
    out[tag] = {f"{k[0]} | {k[1]}": {"mean": sum(v) / len(v), "n": len(v), "min": min(v), "max": max(v)} for k, v in acc.items()}
json.dump(out, open(f"{dst}/pmc_counter_means.json", "w"), indent=1)
subprocess.run([sys.executable, "profiles/summarize_pmc.py", "1000001", P["pmc"] + "fetch_1e6", P["pmc"] + "write_1e6",
                "bench.py --steps 30 --warmup 5 (config3/5, 7-frame ring)"], check=True, stdout=subprocess.DEVNULL)
subprocess.run([sys.executable, "profiles/summarize_pmc.py", "10000001", P["pmc"] + "fetch_1e7", P["pmc"] + "write_1e7",
                "bench.py --n-molecular 10000000 --frames 2 --steps 10 --warmup 2"], check=True, stdout=subprocess.DEVNULL)
b = json.load(open(f"{dst}/bench_final.json"))
print("value", round(b["value"]), "evals/s;", round(1e3 * b["ms_per_step"], 2), "us; frac", round(b["roofline"]["frac"], 3),
      "traffic/algorithmic", round(b["roofline"]["traffic"] / b["roofline"]["algorithmic_bytes_per_launch"], 4))
print({k: (round(v["avg_ms"] * 1e3, 2), round(v["GBps"] or 0)) for k, v in b["roofline"]["kernels"].items()})
print(b["roofline"].get("kernel_time_us_percentiles"))
for k, v in b.get("extras", {}).items():
    print(k, {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk != "note"})
for f in (f"{dst}/bench_1e6_kernel_stats.csv", f"{dst}/bench_1e7_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(os.path.basename(f), r["Name"][:48], r["Calls"], r["AverageNs"])
