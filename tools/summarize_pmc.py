#!/usr/bin/env python3
"""Turns the counter_collection CSVs of tools/collect_profiles.sh into profiles/<tag>_bank_pmc_summary.json
(per-launch means for the bank kernel + derived HBM traffic, VALU instructions per partial-frame group, clock)."""
import collections
import csv
import glob
import json
import sys

src, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r02")
outdir = sys.argv[3] if len(sys.argv) > 3 else "profiles"
commit = sys.argv[4] if len(sys.argv) > 4 else "unknown"
V, P, T = 64, 4096, 4800
out = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(f"{src}/{name}/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "bank_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
stats = {}
for f in glob.glob(f"{src}/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "bank_kernel" in r["Name"]:
            stats = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
pf = V * P * T
avg_s = stats.get("average_ns", 0) * 1e-9
summ = {
    "commit": commit,
    "command": "tools/collect_profiles.sh (rocprofv3 --pmc <counters> -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --repeats 1; one pass per TCC counter)",
    "kernel": "fr::bank_kernel<1, 1, 4>", "per": "launch (one 4800-frame fill_buffer over 64 voices x 4096 partials)",
    "kernel_trace_stats": stats, "counters": out,
    "derived": {
        "hbm_read_bytes": out["FETCH_SIZE"]["mean"] * 1024, "hbm_write_bytes": out["WRITE_SIZE"]["mean"] * 1024,
        "hbm_traffic_bytes": (out["FETCH_SIZE"]["mean"] + out["WRITE_SIZE"]["mean"]) * 1024,
        "algorithmic_bytes": V * P * 8 + 4 * T + 4 * V * T,
        "fetch_note": "FETCH_SIZE/WRITE_SIZE are KiB. The parameter stream is read through the scalar cache (64-B s_load_dwordx16 "
                      "requests), not 16-B/lane vector loads, so the gfx950 x2 FETCH_SIZE correction for wide coalesced reads does not "
                      "apply: the read figure equals 2048 KiB of parameters + the 18.75 KiB time row + tables, i.e. every parameter byte "
                      "leaves HBM once per launch and all 75 time tiles re-read it from L2.",
        "valu_instr_per_partial_frame_group": out["SQ_INSTS_VALU"]["mean"] / (pf / 64),
        "clock_ghz_from_GRBM_GUI_ACTIVE": out["GRBM_GUI_ACTIVE"]["mean"] / 8 / avg_s / 1e9 if avg_s else None,
    },
}
# the HBM-bound variant (tools/track_bench.py --frames 4800): the hipRTC-built bank kernel reading track rows
tr = {}
for name in ("pmc_fetch_tracks", "pmc_write_tracks"):
    for f in glob.glob(f"{src}/{name}/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("jit_bank"):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            # (the priming call -- one long launch that reads no tracks -- is not a step: the median of the launches is a step's figure)
            v = sorted(v)
            tr[k] = {"launches": len(v), "mean": v[len(v) // 2], "min": v[0], "max": v[-1], "mean_is": "median of the launches (one of them is the priming call)"}
tstats = {}
for f in glob.glob(f"{src}/trace_tracks/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("jit_bank"):
            tstats = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
if "FETCH_SIZE" in tr and "WRITE_SIZE" in tr:
    Tt = 4800
    alg = 8.0 * V * P * Tt + 4.0 * Tt + 4.0 * V * Tt
    summ["tracks"] = {
        "command": "python3 tools/track_bench.py --frames 4800 (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; --kernel-trace --stats for the time)",
        "kernel": "jit_bank (hipRTC: the 35-op leaf reading two track rows per partial)", "per": "launch (one 4800-frame call, 524289 input rows: the jit_bank launch; its pieces are added up by chunk_combine_kernel, 5 MB more)",
        "kernel_trace_stats": tstats, "counters": tr,
        "derived": {"hbm_read_bytes": tr["FETCH_SIZE"]["mean"] * 1024 * 2, "hbm_write_bytes": tr["WRITE_SIZE"]["mean"] * 1024,
                    "hbm_traffic_bytes": (2 * tr["FETCH_SIZE"]["mean"] + tr["WRITE_SIZE"]["mean"]) * 1024, "algorithmic_bytes": alg,
                    "traffic_over_algorithmic": (2 * tr["FETCH_SIZE"]["mean"] + tr["WRITE_SIZE"]["mean"]) * 1024 / alg,
                    "fetch_size_raw_KiB": tr["FETCH_SIZE"]["mean"],
                    "fetch_note": "FETCH_SIZE / WRITE_SIZE are KiB.  The read figure carries the gfx950 correction of MI355X_MICROARCH.md (HBM section): "
                                  "FETCH_SIZE reports exactly half the bytes of a coalesced streaming read (128-B requests tallied at 64 B).  The guide "
                                  "calibrates that for 16 B per lane and asks for a calibration of other widths on a known byte count: here the wave's "
                                  "load instruction covers 256 contiguous bytes (4 B per lane), every byte of the 10.07 GB matrix is read exactly once per "
                                  "launch (no reuse; the matrix is 40x the Infinity Cache), so the launch cannot fetch less than the matrix -- the raw "
                                  "counter reads half of it, i.e. the same factor.  WRITE_SIZE is the pieces' workspace (4 pieces x 64 voices x 4800 frames x 4 B)."}}
json.dump(summ, open(f"{outdir}/{tag}_bank_pmc_summary.json", "w"), indent=1)
if "tracks" in summ:
    print(json.dumps(summ["tracks"]["derived"], indent=1))
print(json.dumps(summ["derived"], indent=1))
