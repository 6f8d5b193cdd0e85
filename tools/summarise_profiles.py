#!/usr/bin/env python3
"""Boil the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_*) down to the files kept
under profiles/: per workload the kernel-stats table, the per-launch counter rows of the dominant kernel,
the bench line printed under the profiler, and profiles/pmc_traffic.json (read by bench.py for
roofline.traffic).  HBM bytes per launch = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (MI355X_MICROARCH guide:
both counters are in KB; on gfx950 FETCH_SIZE reports half of a coalesced read)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def one(pattern):
    """the newest match (gpurun merges every call's output into the same local directory)"""
    hits = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit(f"no file for {pattern}")
    return hits[-1]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    wls = sys.argv[2:] or ["c2", "c3", "c3m", "c3a", "c5", "c4", "iat"]
    tpath = os.path.join(PROF, "pmc_traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    # YMC_PROFILE_VARIANT=name: a non-default build of the same workload (selected by an environment variable of the
    # library while collecting) gets its own files and its own key, `<workload>_<name>_K..._T...`
    variant = os.environ.get("YMC_PROFILE_VARIANT", "")
    for wl in wls:
        sdir = os.path.join(OUT, f"prof_{wl}_stats")
        if not os.path.isdir(sdir):
            continue
        line = json.loads(open(os.path.join(OUT, f"prof_{wl}_stats.json")).read().strip().splitlines()[-1])
        cfg = line["config"]
        K = cfg["chains_per_gpu"]
        T = cfg.get("transitions_per_launch") or cfg.get("transitions_per_step") or cfg.get("states")
        label = f"{wl}_{variant}" if variant else wl
        stem = os.path.join(PROF, f"{tag}_{label}_K{K}_T{T}")
        rows = list(csv.DictReader(open(one(os.path.join(sdir, "**", "*_kernel_stats.csv")))))
        with open(stem + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
            w.writeheader()
            w.writerows(rows)
        dominant = rows[0]["Name"]
        json.dump(line, open(stem + "_bench_under_rocprof.json", "w"))
        pmc_rows, sums = [], {}
        grid = None
        for ctr in ("WRITE_SIZE", "FETCH_SIZE", "SQ", "SQW", "GRBM"):
            cdir = os.path.join(OUT, f"prof_{wl}_{ctr}")
            if not os.path.isdir(cdir):
                continue
            try:
                path = one(os.path.join(cdir, "**", "*_counter_collection.csv"))
            except SystemExit:
                continue
            for r in csv.DictReader(open(path)):
                if r["Kernel_Name"] != dominant:
                    continue
                pmc_rows.append({k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count",
                                                   "SGPR_Count", "Counter_Name", "Counter_Value")}
                                | {"Duration_ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
                sums.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                grid, wg = int(r["Grid_Size"]), int(r["Workgroup_Size"])
        if pmc_rows:
            with open(stem + "_pmc.csv", "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=list(pmc_rows[0].keys()))
                w.writeheader()
                w.writerows(pmc_rows)
        med = {k: sorted(v)[len(v) // 2] for k, v in sums.items()}
        if "WRITE_SIZE" in med and "FETCH_SIZE" in med:
            d = 128 if wl.startswith("c3") else 2
            traffic[f"{label}_K{K}_T{T}"] = {
                "hbm_bytes_per_launch": med["WRITE_SIZE"] * 1024 + 2 * med["FETCH_SIZE"] * 1024,
                "WRITE_SIZE_KB": med["WRITE_SIZE"], "FETCH_SIZE_KB_raw": med["FETCH_SIZE"],
                "correction": "bytes = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (gfx950: FETCH_SIZE reports half of a "
                              "coalesced read; WRITE_SIZE exact for 16-B/lane stores, uncalibrated for the tile "
                              "kernel's 8-B/lane stores)",
                "algorithmic_bytes_per_launch": K * T * (8 * d + 1) if wl != "iat" else K * T * 8 * d,
                "dominant_kernel": dominant,
                "sq_per_launch": {k: v for k, v in med.items() if k.startswith("SQ_") or k.startswith("GRBM_")},
                "waves_per_launch": (grid // 64) if grid else None, "workgroup_size": wg if grid else None,
                "source": f"profiles/{os.path.basename(stem)}_pmc.csv (rocprofv3 --pmc, separate passes of the same "
                          "bench command; median over launches)"}
        print(wl, dominant[:60], {k: round(v, 1) for k, v in med.items()})
    if traffic:
        json.dump(traffic, open(tpath, "w"), indent=1)


if __name__ == "__main__":
    main()
