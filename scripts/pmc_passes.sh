#!/bin/bash
# Collect PMC counters for the patch kernel in separate passes (no trace domains besides --kernel-trace).
# usage: scripts/pmc_passes.sh <outdir> -- <program> [args...]   (run on the GPU box, from the repo root)
set -u
OUT=$1; shift; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SALU SQ_LDS_UNALIGNED_STALL" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$REPO/$OUT/pass$i" -- "$@" > "$REPO/$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(f"{out}/summary.txt", "w") as fh:
    for k, d in agg.items():
        fh.write(f"== {k}\n")
        for c, v in sorted(d.items()):
            fh.write(f"  {c:40s} n={len(v):4d} mean={sum(v)/len(v):.6g}\n")
print(open(f"{out}/summary.txt").read())
PY
