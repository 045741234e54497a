"""Reads the kernel and memory-copy traces rocprofv3 wrote for scripts/host_frame_timeline.py (argv: the output directory) and prints the
last apply's copies and launches on one time axis (ms from its first copy)."""
import csv
import pathlib
import sys

root = pathlib.Path(sys.argv[1])
rows = []
for f in root.rglob("*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", ""), None))
for f in root.rglob("*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", r["Kernel_Name"][:40]))
rows.sort()
# applies are separated by >= 10 ms of nothing
groups, cur = [], []
for r in rows:
    if cur and r[0] - max(x[1] for x in cur) > 10_000_000:
        groups.append(cur), (cur := [])
    cur.append(r)
groups.append(cur)
last = groups[-1]
t0 = last[0][0]
for s, e, kind, name in last:
    print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f}  ({(e - s) / 1e3:8.1f} us)  {kind:16s} {name or ''}")
for kind in ("HOST_TO_DEVICE", "DEVICE_TO_HOST", "kernel"):
    sel = [(s, e) for s, e, k, _ in last if k == kind]
    if sel:
        busy = sum(e - s for s, e in sel)
        print(f"{kind}: {len(sel)} ops, busy {busy / 1e6:.3f} ms, span {(min(s for s, _ in sel) - t0) / 1e6:.3f} .. {(max(e for _, e in sel) - t0) / 1e6:.3f} ms")
