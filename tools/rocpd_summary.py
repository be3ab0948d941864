#!/usr/bin/env python3
"""Summaries from rocprofv3's default (rocpd / SQLite) output -- ROCm 7.2 writes *_results.db.

    python tools/rocpd_summary.py stats  DB  > profiles/rNN_bench_c3_kernel_stats.csv
    python tools/rocpd_summary.py pmc --fetch DB --write DB --workload c3 --out profiles/pmc_dominant_kernel.json

`stats` is the per-kernel table of `rocprofv3 --kernel-trace --stats` (calls, total, average, min,
max in ns).  `pmc` folds one FETCH_SIZE pass and one WRITE_SIZE pass into HBM bytes per launch with
the corrections of /opt/skills/guides/MI355X_MICROARCH.md (counters are KiB; on gfx950 FETCH_SIZE
counts half of the bytes of wide coalesced reads -- every load of these kernels -- so it is doubled).
"""
import argparse
import collections
import json
import sqlite3
import sys

NAMES = [("bn_relu_backward_kernel", "bn_relu_backward"), ("bn_relu_forward_kernel", "bn_relu_forward"),
         ("sk_persistent_kernel", "sinkhorn"), ("adamw_step_kernel", "adamw_step"),
         ("linear_fwd_pp3_kernel", "linear_fwd_pp_256x128"), ("linear_fwd_pp2_kernel", "linear_fwd_pp_256x128"), ("linear_fwd_pp_kernel", "linear_fwd_pp_256x128"),
         ("linear_fwd_kernel<2, 2, 2, 2", "linear_fwd_128x128"), ("linear_fwd_kernel<2, 2, 1, 1", "linear_fwd_64x64"),
         ("linear_fwd_kernel<4, 1, 1, 2", "linear_fwd_128x64"), ("linear_fwd_kernel<4, 1, 1, 1", "linear_fwd_128x32"),
         ("rq_assign_kernel", "rq_assign")]


def short(name):
    for k, v in NAMES:
        if k in name:
            return v
    return None


def stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, calls, tot, avg, lo, hi in rows:
        name = name if len(name) <= 160 else name[:157] + "..."
        print(f'"{name}",{calls},{tot},{avg:.1f},{100.0 * tot / total:.4f},{lo},{hi}')


def bygrid(db):
    """Per (kernel, grid) table: the same kernel at different problem sizes (the layers of the MLP) apart."""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
    if not gx or not wx:
        sys.exit(f"kernels view has no grid columns: {cols}")
    rows = c.execute(f"select name, {gx}, {wx}, count(*), avg(end-start), min(end-start), max(end-start) from kernels "
                     f"group by name, {gx}, {wx} order by name, {gx}").fetchall()
    print('"Name","GridX","WorkgroupX","Calls","AverageNs","MinNs","MaxNs"')
    for name, g, w, calls, avg, lo, hi in rows:
        print(f'"{name[:100]}",{g},{w},{calls},{avg:.1f},{lo},{hi}')


def counter(db, which):
    acc = collections.defaultdict(list)
    c = sqlite3.connect(db)
    for name, value in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (which,)):
        k = short(name)
        if k:
            acc[k].append(float(value))
    return acc


def pmc(a):
    fetch, write = counter(a.fetch, "FETCH_SIZE"), counter(a.write, "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        # bench.py's model set-up also launches some of these kernels on its 16 k-row probe: those launches are not the
        # workload (round 1's rq_assign figure averaged 3 full-size launches with 4 probe-sized ones and came out below
        # the algorithmic bytes).  The probe is 4 k rows against >= 131 k per workload launch: anything below 2 % of the
        # kernel's largest launch is set-up.
        fcut, wcut = 0.02 * max(fetch[k] or [0.0]), 0.02 * max(write[k] or [0.0])
        fk = [v for v in fetch[k] if v >= fcut]
        wk = [v for v in write[k] if v >= wcut]
        f = sum(fk) / max(1, len(fk))
        w = sum(wk) / max(1, len(wk))
        out[k] = {"fetch_size_kib_raw": f, "write_size_kib": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                  "launches_sampled": len(fk), "launches_dropped_as_setup": len(fetch[k]) - len(fk),
                  "note": "FETCH_SIZE doubled (gfx950 wide-read correction); average over the launches of the timed workload "
                          "(launches below 2 % of the kernel's largest are bench.py's model set-up on its 4 k-row probe)"}
        print(f"{k:24s} launches {len(fetch[k]):4d}  fetch(raw) {f / 1024:9.1f} MiB  write {w / 1024:9.1f} MiB  -> "
              f"{out[k]['hbm_bytes_per_launch'] / 1e6:10.1f} MB per launch")
    if a.out:
        try:
            doc = json.load(open(a.out))
        except (OSError, ValueError):
            doc = {}
        doc[a.workload] = out
        json.dump(doc, open(a.out, "w"), indent=1, sort_keys=True)


MFMA_KERNELS = [("linear_fwd_pp3_kernel", "linear_fwd_pp3 (256x128 ping-pong)"), ("rq_assign_kernel", "rq_assign"),
                ("linear_fwd_kernel<2, 2, 1, 1", "linear_fwd 64x64 (training step)"),
                ("linear_s16_kernel", "linear_s16 32x64 on 16x16x4 (training step, under-filled launches)")]
CUS, SIMDS = 256, 4


def mfma(dbs):
    """Per-launch averages of the SQ counters of tools/pmc_mfma.sh for the three MFMA kernels, beside the launch duration of
    the same (profiled) runs.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMD pipes; `util` divides it by
    1024 x the launch's cycles -- taken from GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs) -- so 1.0 = every
    MFMA pipe busy every cycle of the launch."""
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for db in dbs:
        c = sqlite3.connect(db)
        for name, cname, value in c.execute("select kernel_name, counter_name, value from counters_collection"):
            for key, label in MFMA_KERNELS:
                if key in name:
                    vals[label][cname].append(float(value))
        for name, d in c.execute("select name, end - start from kernels"):
            for key, label in MFMA_KERNELS:
                if key in name:
                    dur[label].append(float(d))
    for _key, label in MFMA_KERNELS:
        if label not in vals:
            continue
        v = vals[label]
        # the large launches of the target only (set-up launches and the 1-tile probes fall below half of the largest)
        avg = {}
        for cname, xs in v.items():
            cut = 0.5 * max(xs)
            keep = [x for x in xs if x >= cut] or xs
            avg[cname] = sum(keep) / len(keep)
        ds = dur[label]
        dcut = 0.5 * max(ds)
        dk = [x for x in ds if x >= dcut]
        d_ns = sum(dk) / len(dk)
        print(f"{label}: {len(dk)} launches per pass-set, average duration {d_ns / 1e3:.1f} us (profiled runs)")
        for cname in sorted(avg):
            print(f"    {cname:32s} {avg[cname]:16.0f} per launch")
        busy, gui = avg.get("SQ_VALU_MFMA_BUSY_CYCLES"), avg.get("GRBM_GUI_ACTIVE")
        if busy and gui:
            cycles = gui / 8.0
            print(f"    launch cycles (GRBM_GUI_ACTIVE / 8 XCDs) {cycles:12.0f}  -> clock {cycles / d_ns:.2f} GHz")
            print(f"    MFMA pipe utilisation = MFMA_BUSY / ({CUS * SIMDS} pipes x cycles) = {busy / (CUS * SIMDS * cycles):.3f}")
        if busy:
            # GRBM_GUI_ACTIVE reads high on dispatches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS give-back): the same
            # ratio against the launch's wall time at the 2.4 GHz peak clock (the clock the 157.3 TFLOP/s roof assumes)
            print(f"    MFMA_BUSY / (1024 pipes x duration x 2.4 GHz) = {busy / (CUS * SIMDS * d_ns * 2.4):.3f}")
        n_mfma = avg.get("SQ_INSTS_MFMA")
        if n_mfma and busy:
            # v_mfma_f32_32x32x2_f32: 32*32*2 multiply-adds = 4096 flop, 64 pipe cycles each; v_mfma_f32_16x16x4_f32: 2048 flop, 32
            # cycles (busy / instructions tells which)
            per = 2048 if busy / n_mfma < 48 else 4096
            print(f"    MFMA instructions {n_mfma:.0f} x {per} flop / duration = {n_mfma * per / d_ns / 1e3:.1f} TFLOP/s "
                  f"(roof 157.3); busy cycles per instruction {busy / n_mfma:.1f}")
        wc = avg.get("SQ_WAVE_CYCLES")
        if wc:
            for cname in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
                if cname in avg:
                    print(f"    {cname} / SQ_WAVE_CYCLES = {avg[cname] / wc:.3f}")


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("stats")
    s.add_argument("db")
    g = sub.add_parser("bygrid")
    g.add_argument("db")
    p = sub.add_parser("pmc")
    p.add_argument("--fetch", required=True)
    p.add_argument("--write", required=True)
    p.add_argument("--workload", default="c3")
    p.add_argument("--out", default=None)
    m = sub.add_parser("mfma")
    m.add_argument("dbs", nargs="+")
    a = ap.parse_args()
    if a.cmd == "mfma":
        mfma(a.dbs)
    elif a.cmd == "stats":
        stats(a.db)
    elif a.cmd == "bygrid":
        bygrid(a.db)
    else:
        pmc(a)


if __name__ == "__main__":
    sys.exit(main())
