#!/bin/bash
# run on the GPU box from the repo root:  scripts/evidence.sh r05zz
# Order matters: the PMC passes come first and refresh profiles/traffic_latest.json, so that the bench lines written afterwards cite THIS set
# as their traffic_source (round 3's cited the set before).
R=${1:-r04}
mkdir -p gpurun_out/$R
REPO=$(pwd)
python -m pytest tests -m gpu -x -q > gpurun_out/$R/pytest_gpu.log 2>&1; tail -2 gpurun_out/$R/pytest_gpu.log
# (--no-e2e --no-verify: only the launches of the timed loop, its warm-up and prewarm - the end-to-end leg launches the same kernel on row bands)
bash scripts/pmc_passes.sh gpurun_out/$R/pmc -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu --new-frames 0 --no-e2e --no-verify > gpurun_out/$R/pmc.log 2>&1
cp gpurun_out/$R/pmc/summary.txt gpurun_out/$R/bench_pmc_summary.txt; head -30 gpurun_out/$R/bench_pmc_summary.txt
python3 scripts/traffic.py gpurun_out/$R/bench_pmc_summary.txt "profiles/${R}_bench_pmc_summary.txt" > gpurun_out/$R/traffic.json; cat gpurun_out/$R/traffic.json
cp gpurun_out/$R/traffic.json profiles/traffic_latest.json
python bench.py --steps 50 --warmup 5 > gpurun_out/$R/bench.json 2> gpurun_out/$R/bench.err; cat gpurun_out/$R/bench.json | cut -c1-300
python bench.py --config 2 --steps 50 --warmup 5 > gpurun_out/$R/bench_config2.json 2>/dev/null
python bench.py --config 4 --steps 20 --warmup 3 > gpurun_out/$R/bench_config4_8192.json 2>/dev/null
python bench.py --config 5 --steps 20 --warmup 3 --streamed > gpurun_out/$R/bench_config5_batch.json 2>/dev/null
python bench.py --config 1 --steps 50 --warmup 5 --new-frames 0 > gpurun_out/$R/bench_config1.json 2>/dev/null
python bench.py --config 6 --steps 50 --warmup 5 --new-frames 0 --no-cpu > gpurun_out/$R/bench_config6_4096_n64.json 2>/dev/null
python bench.py --config construct > gpurun_out/$R/bench_construct.json 2>/dev/null
for f in config1 config2 config4_8192 config5_batch config6_4096_n64 construct; do cut -c1-200 gpurun_out/$R/bench_$f.json; done
python scripts/notebook_workload.py > gpurun_out/$R/notebook_workload.log 2>&1; grep frames gpurun_out/$R/notebook_workload.log
python scripts/band_times.py --steps 200 --seam recompute > gpurun_out/$R/band_times_recompute.log 2>&1; tail -1 gpurun_out/$R/band_times_recompute.log | cut -c1-300
python scripts/band_times.py --steps 200 --seam exchange --pipeline --worlds 1,8 > gpurun_out/$R/band_times_exchange_pipeline.log 2>&1; tail -1 gpurun_out/$R/band_times_exchange_pipeline.log | cut -c1-300
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/$R/stats -- python3 $REPO/bench.py --steps 100 --warmup 5 --no-cpu --new-frames 0 --no-e2e --no-verify > $REPO/gpurun_out/$R/stats.log 2>&1)
find gpurun_out/$R/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$R/bench_kernel_stats.csv
head -4 gpurun_out/$R/bench_kernel_stats.csv
rm -rf gpurun_out/$R/stats gpurun_out/$R/pmc/pass*
