"""One-off robustness run: the persistent fused launch against the one-patch-per-workgroup launch and the separate sum kernel, bit
for bit, over many frame sizes (whole and partial rounds, one to many rounds, square and not), repeated applies of each.
    python scripts/stress_sizes.py [repeats]"""
import pathlib, sys
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering

repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = 256
rng = np.random.default_rng(11)
bad = 0
for (h, w) in ((512, 512), (768, 2048), (1024, 1280), (2048, 2048), (2304, 2304), (2560, 4096), (3968, 3968), (4096, 4096), (4224, 4224), (6144, 3072)):
    coords = [tuple(int(v) for v in t) for t in calculate_covering((h, w), n)]
    k = (rng.standard_normal((len(coords), n, n), dtype=np.float32) + 1j * rng.standard_normal((len(coords), n, n), dtype=np.float32)).astype(np.complex64)
    img = (100 + 5 * rng.standard_normal((h, w), dtype=np.float32)).astype(np.float32)
    outs = {}
    for mode, options in (("persistent", {}), ("one patch per workgroup", {"persist": 0}), ("separate sum", {"fuse": 0})):
        plan = _native.Plan(n, coords)
        for name, value in options.items():  # (plan options since round 6: the shipped library reads no such environment variable)
            plan.set_option(name, value)
        plan.set_transfer(k)
        first = plan.apply(img, 1)
        reps = repeats if mode == "persistent" else 3
        same = all(np.array_equal(plan.apply(img, 1), first) for _ in range(reps))
        outs[mode] = first
        bad += not same
        plan.close()
    agree = np.array_equal(outs["persistent"], outs["one patch per workgroup"]) and np.array_equal(outs["persistent"], outs["separate sum"])
    bad += not agree
    print(f"{h}x{w}: {len(coords)} patches, the three forms agree bit for bit: {agree}", flush=True)
print(f"stress_sizes: {bad} problems")
sys.exit(1 if bad else 0)
