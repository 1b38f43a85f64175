#!/usr/bin/env python3
"""Turns rocprofv3 --pmc counter CSVs into profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

    python profiles/summarize_pmc.py <N_particles> <fetch_dir> <write_dir> [<label>]

FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2).
Corrections, as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes:
  - both counters are in KiB: x 1024;
  - on gfx950 FETCH_SIZE reports exactly HALF the bytes of a wide coalesced streaming read (16 B/lane): x 2;
  - WRITE_SIZE is exact for 16-B-per-lane streaming stores.
Values are per launch (mean over the dispatches of each kernel, warm-up dispatches included: every launch moves
the same bytes).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def short(name):
    for key in ("cavity_persistent_kernel", "dipole_partials_kernel", "force_map_aos_fused_kernel", "force_map_aos_kernel",
                "finalize_kernel", "force_map_strided_kernel"):
        if key in name:
            return key
    return None


def collect(directory, counter):
    acc = defaultdict(list)
    paths = sorted(glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for path in paths[-1:]:  # the most recent run only (gpurun_out/ accumulates earlier ones)
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                k = short(row.get("Kernel_Name", ""))
                if k:
                    acc[k].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    n = int(sys.argv[1])
    fetch_dir, write_dir = sys.argv[2], sys.argv[3]
    label = sys.argv[4] if len(sys.argv) > 4 else ""
    fetch, nf = collect(fetch_dir, "FETCH_SIZE")
    write, nw = collect(write_dir, "WRITE_SIZE")
    # the single-launch kernel is priced at the evaluation's 92 N although it moves 84 N (charges stay in LDS): SURVEY.md 8(d)
    algorithmic = {"cavity_persistent_kernel": 92 * n, "dipole_partials_kernel": 52 * n, "force_map_aos_fused_kernel": 40 * n,
                   "force_map_aos_kernel": 40 * n}
    out = {"_note": "HBM bytes per launch from rocprofv3 --pmc; FETCH_SIZE x1024 x2 (gfx950 correction), WRITE_SIZE x1024",
           "_label": label}
    for k in sorted(set(fetch) | set(write)):
        rd = fetch.get(k, 0.0) * 1024.0 * 2.0
        wr = write.get(k, 0.0) * 1024.0
        ent = {"fetch_size_raw_KiB": fetch.get(k), "write_size_raw_KiB": write.get(k), "read_bytes": rd, "write_bytes": wr,
               "hbm_bytes_per_launch": rd + wr, "dispatches_fetch_pass": nf.get(k, 0), "dispatches_write_pass": nw.get(k, 0)}
        if k in algorithmic:
            ent["algorithmic_bytes"] = algorithmic[k]
            ent["traffic_over_algorithmic"] = (rd + wr) / algorithmic[k]
        out[k] = ent
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            allv = json.load(f)
    except (OSError, ValueError):
        allv = {}
    allv[str(n)] = out
    with open(path, "w") as f:
        json.dump(allv, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
