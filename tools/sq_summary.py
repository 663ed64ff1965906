#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes of SQ counters over one short bench run into a per-kernel text summary.

    # on the GPU box (each pass its own run; --pmc is never combined with other trace domains)
    cd /tmp && export TMPDIR=/tmp
    for i in 1 2 3; do rocprofv3 --kernel-trace --pmc $(python tools/sq_summary.py --pass $i) -d gpurun_out/sq$i -o sq$i \
        -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof; done
    python tools/sq_summary.py gpurun_out/sq1 gpurun_out/sq2 gpurun_out/sq3 > profiles/rNN_sq_counters.txt

Per kernel the dispatch with the largest SQ_WAVE_CYCLES (the news-level launch) is reported.
"""
import csv
import glob
import sys
from collections import defaultdict

PASSES = {
    1: ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES"],
    2: ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_INST_LEVEL_VMEM", "SQ_VALU_MFMA_BUSY_CYCLES",
        "SQ_ACTIVE_INST_VMEM"],
    3: ["SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
        "SQ_INSTS_VMEM"],
}


def load(d):
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    rows = defaultdict(lambda: defaultdict(dict))          # kernel -> dispatch -> counter -> value
    for f in files:
        for r in csv.DictReader(open(f)):
            rows[r["Kernel_Name"]][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    out = {}
    for k, disp in rows.items():
        best = max(disp.values(), key=lambda c: c.get("SQ_WAVE_CYCLES", 0.0))
        out[k] = best
    return out


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--pass":
        print(" ".join(PASSES[int(sys.argv[2])]))
        return
    merged = defaultdict(dict)
    for d in sys.argv[1:]:
        for k, c in load(d).items():
            for name, v in c.items():
                merged[k].setdefault(name, v)
    for k, c in sorted(merged.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = c.get("SQ_WAVE_CYCLES", 0)
        if wc < 5e7:
            continue
        g = lambda n: c.get(n, 0.0)
        pct = lambda n: 100.0 * g(n) / wc if wc else 0.0
        print(k[:110])
        print(f"   wave-cycles {wc:.3e} (quad-cycles, summed over waves)   waiting on s_waitcnt/barrier {pct('SQ_WAIT_ANY'):.0f}%   "
              f"issue stalls {pct('SQ_WAIT_INST_ANY'):.0f}%   issuing {pct('SQ_ACTIVE_INST_ANY'):.0f}%")
        print(f"   of wave time: VALU {pct('SQ_ACTIVE_INST_VALU'):.0f}%  LDS {pct('SQ_ACTIVE_INST_LDS'):.0f}%  scalar {pct('SQ_ACTIVE_INST_SCA'):.0f}%  "
              f"VMEM {pct('SQ_ACTIVE_INST_VMEM'):.0f}%   LDS-issue stall {pct('SQ_WAIT_INST_LDS'):.0f}%   MFMA-busy/wave-cycles {pct('SQ_VALU_MFMA_BUSY_CYCLES'):.0f}%")
        la = g("SQ_LDS_IDX_ACTIVE")
        print(f"   LDS bank-conflict cycles / LDS active cycles {100.0 * g('SQ_LDS_BANK_CONFLICT') / la if la else 0:.0f}%   "
              f"instructions: VALU {g('SQ_INSTS_VALU'):.3e} SALU {g('SQ_INSTS_SALU'):.3e} LDS {g('SQ_INSTS_LDS'):.3e} VMEM {g('SQ_INSTS_VMEM'):.3e}")
        print("   raw: " + " ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))


if __name__ == "__main__":
    main()
