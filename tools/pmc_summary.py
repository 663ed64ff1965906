#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (one `--pmc FETCH_SIZE`, one `--pmc WRITE_SIZE`, each with --kernel-trace only) of
`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof` into profiles/<tag>_hbm_traffic_pmc.json.

    python tools/pmc_summary.py gpurun_out/pmc_fetch_d gpurun_out/pmc_write_d profiles/r01d_hbm_traffic_pmc.json [NRMS bf16 512 4]

The optional tail (model, dtype, per-GPU batch, steps the profiled command ran incl. warm-up) adds "step_total": the bytes
of ALL dispatches of the run (every kernel, torch's own included) divided by the step count -- what bench.py's
roofline.step compares with the fused-minimum bytes of SURVEY §8(d).

Per kernel the dispatch with the largest counter value (the news-level launch) is reported.  FETCH_SIZE / WRITE_SIZE
are in KiB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts half of wide coalesced read streams -> the x2 column.
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv")
    best, cnt, total = {}, defaultdict(int), 0.0
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        cnt[k] += 1
        best[k] = max(best.get(k, 0.0), float(r["Counter_Value"]))
        total += float(r["Counter_Value"])
    return best, cnt, total


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fe, cnt, fe_total = load(fetch_dir, "FETCH_SIZE")
    wr, _, wr_total = load(write_dir, "WRITE_SIZE")
    rows = []
    for k in sorted(fe, key=lambda k: -(fe[k] + wr.get(k, 0.0))):
        if fe[k] + wr.get(k, 0.0) < 50e3:      # < 50 MB per launch: not a news-level kernel
            continue
        rows.append({"kernel": k[:160], "dispatches": cnt[k],
                     "FETCH_SIZE_KB_max_dispatch": fe[k], "fetch_GB_raw": round(fe[k] * 1024 / 1e9, 3),
                     "fetch_GB_x2_gfx950_16B_correction": round(2 * fe[k] * 1024 / 1e9, 3),
                     "WRITE_SIZE_KB_max_dispatch": wr.get(k, 0.0), "write_GB": round(wr.get(k, 0.0) * 1024 / 1e9, 3)})
    note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 3 "
            "--warmup 1 --no-cpu-baseline --no-prof`; values of the largest (news-level) dispatch of each kernel. "
            "MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of wide (16 B/lane) coalesced streams on gfx950 -> x2 column; "
            "the attention kernels use 8 B/lane loads (uncalibrated width), so both raw and x2 are given.")
    doc = {"note": note, "kernels": rows}
    if len(sys.argv) >= 8:
        model, dtype, batch, steps = sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
        by = (2 * fe_total + wr_total) * 1024 / steps
        doc["step_total"] = {"model": model, "dtype": dtype, "batch": batch, "steps_in_run": steps,
                             "fetch_GB_x2_per_step": round(2 * fe_total * 1024 / steps / 1e9, 3),
                             "write_GB_per_step": round(wr_total * 1024 / steps / 1e9, 3), "bytes_per_step": round(by),
                             "note": "all dispatches of the run (incl. torch's own kernels and one-time packing) / steps"}
        print(f"per step: fetch x2 {doc['step_total']['fetch_GB_x2_per_step']} GB + write {doc['step_total']['write_GB_per_step']} GB")
    json.dump(doc, open(out, "w"), indent=1)
    for r in rows:
        print(f"{r['kernel'][:70]:70s} fetch x2 {r['fetch_GB_x2_gfx950_16B_correction']:7.3f} GB  write {r['write_GB']:6.3f} GB")


if __name__ == "__main__":
    main()
