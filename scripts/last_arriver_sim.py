"""Development aid: would "the last of a tile's contributors sums the tile" stall?  Takes the per-patch phase stamps of a real
run of the persistent kernel (RPSF_STAMPS build) and replays the protocol on them: a contributor *arrives* at a tile when its
store phase starts (stamp 9), it has *published* its quadrant `publish_us` after the end of its store phase (stamp 12); the
last arriver handles its other quadrants first (`early_us`) and then needs every other contributor of the tile published.
Prints the distribution of the extra wait per patch for several stagger settings.
    RPSF_LIB=devlibs/stamps_p.so python scripts/last_arriver_sim.py [--size 4096]"""
import argparse, pathlib, sys
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from regularizepsf_amd import _native, calculate_covering
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
a = ap.parse_args()
n, size = 256, a.size
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in calculate_covering((size, size), n)]
k = np.empty((len(coords), n, n), np.complex64); k.real = rng.standard_normal(k.shape, dtype=np.float32); k.imag = rng.standard_normal(k.shape, dtype=np.float32)
img = (100 + 5 * rng.standard_normal((size, size), dtype=np.float32)).astype(np.float32)
c = np.array(coords); half = n // 2
li, lj = (c[:, 0] - c[:, 0].min()) // half, (c[:, 1] - c[:, 1].min()) // half
for stagger in (0, 10, 20, 37):
    plan = _native.Plan(n, coords)
    plan.set_transfer(k)
    plan.set_stagger(stagger)
    d_img = _native.DeviceBuffer(img.nbytes).upload(img); d_out = _native.DeviceBuffer(img.nbytes)
    geom = _native.Geometry.whole(size, size, 1)
    plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 3)
    tot, ker = plan.apply_device_timed(d_img.ptr, d_out.ptr, geom, 1)
    st = plan.debug_stamps().astype(np.int64) * 0.01  # us
    arrive, stored = st[:, 9], st[:, 12]
    tiles = {}
    for p in range(len(coords)):
        for q in range(4):
            tiles.setdefault((li[p] + (q >> 1), lj[p] + (q & 1)), []).append(p)
    for mode, publish_us, early_us in (("dynamic", 1.0, 4.0), ("dynamic", 4.0, 4.0), ("static", 1.0, 4.0), ("static", 4.0, 4.0), ("static", 4.0, 0.0)):
        wait = np.zeros(len(coords))
        for who in tiles.values():
            if len(who) < 2:
                continue
            t = arrive[who] if mode == "dynamic" else st[who, 0]  # static: the contributor that started last (queue order)
            last = who[int(np.argmax(t))]
            others = [w for w in who if w != last]
            ready = max(stored[w] + publish_us for w in others)
            wait[last] = max(wait[last], ready - (arrive[last] + early_us))
        wait = np.maximum(wait, 0)
        print(f"stagger {stagger:2d} us, kernel {ker[0]*1e3:6.1f} us | {mode:7s} finisher, publish +{publish_us} us, early stores {early_us} us: extra wait per patch "
              f"mean {wait.mean():5.2f}  median {np.median(wait):5.2f}  p90 {np.percentile(wait, 90):5.2f}  max {wait.max():5.1f} us; patches that wait {100*(wait>0).mean():4.1f} %")
    plan.close(); d_img.free(); d_out.free()
