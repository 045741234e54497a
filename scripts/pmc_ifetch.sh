#!/bin/bash
# Development aid (GPU box, from the repo root): instruction-fetch counters of the patch kernel in separate passes.
#   scripts/pmc_ifetch.sh <outdir> [bench args]
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_TC_INST_REQ SQC_TC_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$REPO/$OUT/pass$i" -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu --new-frames 0 "$@" > "$REPO/$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(f"== {k}")
    for c, v in sorted(d.items()):
        print(f"  {c:40s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
PY
rm -rf $OUT/pass*/
