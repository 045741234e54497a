#!/bin/bash
REPO=$(pwd)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/gaps -- python3 $REPO/bench.py --steps 60 --warmup 5 > $REPO/gpurun_out/gaps.log 2>&1)
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/gaps/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'][:20], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
import statistics
g15, g51, d1, d5 = [], [], [], []
for a, b in zip(seq, seq[1:]):
    gap = b[1] - a[2]
    if 'patch_kernel' in a[0] and 'sum_planes' in b[0]: g15.append(gap); d1.append(a[2]-a[1])
    if 'sum_planes' in a[0] and 'patch_kernel' in b[0]: g51.append(gap); d5.append(a[2]-a[1])
print('K1 dur med', statistics.median(d1), 'K5 dur med', statistics.median(d5))
print('gap K1->K5 med', statistics.median(g15), 'ns; gap K5->K1 med', statistics.median(g51), 'ns; n', len(g15), len(g51))
PY
rm -rf gpurun_out/gaps
