#!/bin/bash
# run on the GPU box from the repo root:  scripts/evidence_r06.sh r06zz   - the round's whole evidence set (headline first, then the sweep kernel's)
R=${1:-r06zz}
bash scripts/evidence.sh $R > gpurun_out/${R}_evidence.log 2>&1
REPO=$(pwd)
for c in 6 7; do
  bash scripts/pmc_passes.sh gpurun_out/$R/pmc$c -- python3 $REPO/bench.py --config $c --steps 10 --warmup 2 --no-cpu --new-frames 0 --no-e2e --no-verify > gpurun_out/$R/pmc$c.log 2>&1
  cp gpurun_out/$R/pmc$c/summary.txt gpurun_out/$R/bench_config${c}_pmc_summary.txt
  python3 scripts/traffic.py gpurun_out/$R/bench_config${c}_pmc_summary.txt "profiles/${R}_bench_config${c}_pmc_summary.txt" "4096x4096 / $([ $c = 6 ] && echo 64 || echo 32)-px patches, 1 GPU" > gpurun_out/$R/traffic_config$c.json
  cp gpurun_out/$R/traffic_config$c.json profiles/traffic_config$c.json
  rm -rf gpurun_out/$R/pmc$c/pass*
done
for c in 6 7 8; do python bench.py --config $c --steps 50 --warmup 5 --no-cpu > gpurun_out/$R/bench_config$c.json 2>/dev/null; cut -c1-250 gpurun_out/$R/bench_config$c.json; done >> gpurun_out/${R}_evidence.log
python bench.py --patch 96 --steps 20 > gpurun_out/$R/bench_patch96.json 2>/dev/null; cut -c1-250 gpurun_out/$R/bench_patch96.json >> gpurun_out/${R}_evidence.log
for c in 6 7; do
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/$R/stats$c -- python3 $REPO/bench.py --config $c --steps 100 --warmup 5 --no-cpu --new-frames 0 --no-e2e --no-verify > $REPO/gpurun_out/$R/stats$c.log 2>&1)
  find gpurun_out/$R/stats$c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$R/bench_config${c}_kernel_stats.csv
  head -3 gpurun_out/$R/bench_config${c}_kernel_stats.csv >> gpurun_out/${R}_evidence.log
  rm -rf gpurun_out/$R/stats$c
done
tail -60 gpurun_out/${R}_evidence.log | cut -c1-300
