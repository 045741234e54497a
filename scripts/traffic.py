"""HBM-side traffic per apply from a PMC summary (scripts/pmc_passes.sh): FETCH_SIZE / WRITE_SIZE of every kernel of the apply.

    python scripts/traffic.py <summary.txt> <name the summary is committed under> [workload]  > traffic.json

gfx950 counts 16-byte-per-lane coalesced reads at half their bytes in FETCH_SIZE (MI355X_MICROARCH.md, HBM section); every
read of the second-generation path (packed K, pixel gather, plane sum) is such a load, so the read figure is 2 x FETCH_SIZE;
WRITE_SIZE is exact for 16-byte streaming stores.  Counters are in KiB per dispatch, averaged over the dispatches of the run.
"""
import json
import re
import sys

text = open(sys.argv[1]).read()
kernels = {}
cur = None
for line in text.splitlines():
    if line.startswith("== "):
        cur = line[3:].strip()
        kernels[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+mean=(\S+)", line)
        if m and cur:
            kernels[cur][m.group(1)] = float(m.group(3))
out = {"source": f"{sys.argv[2]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, `python3 bench.py --steps 10 --warmup 2 --no-cpu --new-frames 0` via "
                 "scripts/evidence.sh; mean over the dispatches of each kernel, prewarm included)",
       "workload": sys.argv[3] if len(sys.argv) > 3 else "4096x4096 / 256-px patches, 1 GPU", "kernels": {}}
total = 0
for name, c in kernels.items():
    if "FETCH_SIZE" not in c or not any(k in name for k in ("patch_kernel", "sum_planes", "sum_tiles", "fixup", "sweep_kernel")):
        continue
    read_b, write_b = 2 * c["FETCH_SIZE"] * 1024, c.get("WRITE_SIZE", 0.0) * 1024
    out["kernels"][name[:60]] = {"fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c.get("WRITE_SIZE", 0.0),
                                 "read_bytes_corrected": int(read_b), "write_bytes": int(write_b)}
    total += read_b + write_b
out["correction"] = "reads = 2 x FETCH_SIZE (all reads are 16 B per lane: K stream, gather, plane sum); writes = WRITE_SIZE"
out["traffic_bytes_per_launch"] = int(total)
print(json.dumps(out, indent=1))
