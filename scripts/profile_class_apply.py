import sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np
import regularizepsf_amd as rp
n, size = 64, 512
rng = np.random.default_rng(0)
coords = [tuple(int(v) for v in t) for t in rp.calculate_covering((size, size), n)]
k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
t = rp.ArrayPSFTransform(rp.IndexedCube(coords, k))
img = rng.standard_normal((size, size))
for _ in range(5): t.apply(img)
ts = []
for _ in range(300):
    t0 = time.perf_counter(); t.apply(img); ts.append(time.perf_counter() - t0)
print("transform.apply float64 image: median %.1f us" % (1e6 * np.median(ts)))
img32 = img.astype(np.float32)
ts = []
for _ in range(300):
    t0 = time.perf_counter(); t.apply(img32); ts.append(time.perf_counter() - t0)
print("transform.apply float32 image: median %.1f us" % (1e6 * np.median(ts)))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): t.apply(img)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
