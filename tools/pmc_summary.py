#!/usr/bin/env python3
"""Fold rocprofv3 --pmc CSVs (FETCH_SIZE and WRITE_SIZE, one pass each) into per-kernel HBM traffic.

    python tools/pmc_summary.py --fetch DIR_FETCH --write DIR_WRITE --workload c3 --out profiles/pmc_dominant_kernel.json

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming
read, which is what every load of these kernels is, so it is doubled; WRITE_SIZE is exact for
16-B-per-lane stores.  Traffic is reported per launch of each lcrec kernel.
"""
import argparse
import collections
import csv
import glob
import json
import os

NAMES = {"linear_fwd_pp2_kernel": "linear_fwd_pp_256x128", "linear_fwd_pp_kernel": "linear_fwd_pp_256x128", "linear_fwd_kernel<2, 2, 2, 2>": "linear_fwd_128x128",
         "linear_fwd_kernel<4, 1, 1, 2>": "linear_fwd_128x64", "linear_fwd_kernel<4, 1, 1, 1>": "linear_fwd_128x32",
         "rq_assign_kernel": "rq_assign"}


def short(name):
    for k, v in NAMES.items():
        if k in name:
            return v
    return None


def fold(directory, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            if k:
                acc[k].append(float(r["Counter_Value"]))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    fetch, write = fold(a.fetch, "FETCH_SIZE"), fold(a.write, "WRITE_SIZE")
    table = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch[k]) / max(1, len(fetch[k]))
        w = sum(write[k]) / max(1, len(write[k]))
        table[k] = {"launches_sampled": len(fetch[k]), "fetch_size_kib_raw": f, "write_size_kib": w,
                    "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                    "note": "FETCH_SIZE doubled (gfx950 wide-read correction); averages over all launches of the kernel"}
        print(f"{k:24s} launches {len(fetch[k]):4d}  fetch(raw) {f / 1024:9.1f} MiB  write {w / 1024:9.1f} MiB  "
              f"-> {table[k]['hbm_bytes_per_launch'] / 1e6:10.1f} MB per launch")
    if a.out:
        data = {}
        if os.path.exists(a.out):
            data = json.load(open(a.out))
        data[a.workload] = table
        with open(a.out, "w") as fh:
            json.dump(data, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
