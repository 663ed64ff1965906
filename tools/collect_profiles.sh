#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/collect_profiles.sh r03 [bench]
# Writes profiles/<tag>_* summaries (bench lines, kernel stats, HBM traffic PMC with per-step totals, SQ counters).
# Every rocprofv3 pass is its own run (--pmc is never combined with other trace domains); raw output stays in /tmp.
set -u
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p profiles gpurun_out
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-also"
echo "[1/5] bench lines"
python3 bench.py > profiles/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || echo bench failed          # headline + also {naml, eval, dense, fp32} + cpu_baseline
python3 bench.py --feed device --no-cpu-baseline --no-also > profiles/${TAG}_bench_feed_device.json 2>> gpurun_out/${TAG}_bench.err || echo feed failed
python3 bench.py --deterministic --no-cpu-baseline --no-also > profiles/${TAG}_bench_deterministic.json 2>> gpurun_out/${TAG}_bench.err || echo det failed
NR_NO_COMPACT_ROWS=1 NR_NO_POOL_FUSED=1 python3 bench.py --no-cpu-baseline --no-also > profiles/${TAG}_bench_round2_paths.json 2>> gpurun_out/${TAG}_bench.err || echo old-path failed
if [ "${2:-all}" = "bench" ]; then mkdir -p gpurun_out/profiles_${TAG}; cp profiles/${TAG}_bench*.json gpurun_out/profiles_${TAG}/; exit 0; fi   # bench lines only
echo "[2/5] kernel trace + stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > /tmp/kt.log 2>&1 || echo kernel-trace failed
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" profiles/${TAG}_kernel_stats.csv
echo "[3/5] PMC FETCH_SIZE"; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pf -o pf -- $CMD > /tmp/pf.log 2>&1 || echo fetch failed
echo "[4/5] PMC WRITE_SIZE"; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/pw -o pw -- $CMD > /tmp/pw.log 2>&1 || echo write failed
python3 tools/pmc_summary.py /tmp/pf /tmp/pw profiles/${TAG}_hbm_traffic_pmc.json NRMS bf16 512 4 | tail -25
echo "[5/5] SQ counters"
for i in 1 2 3; do timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $(python3 tools/sq_summary.py --pass $i) -d /tmp/sq$i -o sq$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-also > /tmp/sq$i.log 2>&1 || echo sq pass $i failed; done
{ echo "rocprofv3 --pmc passes (tools/sq_summary.py) over \`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-also\`; largest dispatch per kernel."; python3 tools/sq_summary.py /tmp/sq1 /tmp/sq2 /tmp/sq3; } > profiles/${TAG}_sq_counters.txt
mkdir -p gpurun_out/profiles_${TAG}; cp profiles/${TAG}_* gpurun_out/profiles_${TAG}/
ls -la profiles/${TAG}_*
