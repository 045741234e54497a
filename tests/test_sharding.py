"""Row-band sharding: plan geometry (single process) and the seam exchange with world_size 2 over gloo.

The per-band compute is done here by the CPU oracle restricted to the band's patches (the oracle is the
checker; the product's band compute is the HIP kernel, covered on the GPU by
tests/test_gpu_parity.py::test_row_bands_on_one_gpu).  What this pins is the product's band plan:
which patches, which output rows are owned, which rows are sent / received / added where.
"""

import os
import socket

import numpy as np
import pytest

from oracle import regpsf_oracle as orc
from regularizepsf_amd.sharding import make_band_plans, pad_rows


def small_case():
    h, w, n = 256, 96, 32
    coords = [tuple(int(v) for v in t) for t in orc.calculate_covering((h, w), n)]
    rng = np.random.default_rng(4)
    k = (rng.standard_normal((len(coords), n, n)) + 1j * rng.standard_normal((len(coords), n, n))).astype(np.complex64)
    image = rng.standard_normal((h, w)).astype(np.float32)
    return h, w, n, coords, k, image


def band_buffer(plan, image, coords, k):
    """What the rank's device buffer holds after its local apply: its patches only, rows [out_row0, +out_rows)."""
    part = orc.apply_transfer(image, [coords[i] for i in plan.patch_index], k[plan.patch_index])
    return part[plan.out_row0 : plan.out_row0 + plan.out_rows].copy()


@pytest.mark.parametrize("world", [1, 2, 3, 4])
def test_band_plans_partition_patches_and_rows(world):
    h, w, n, coords, k, image = small_case()
    plans = make_band_plans(coords, n, h, world)
    seen = sorted(i for p in plans for i in p.patch_index)
    assert seen == list(range(len(coords)))
    assert sum(p.own_rows for p in plans) == h
    assert plans[0].recv_rows == 0 and plans[-1].send_rows == 0
    counts = [len(p.patch_index) for p in plans]
    assert max(counts) - min(counts) <= len({c for _, c in coords}) * 2  # balanced to within two lattice rows
    full = orc.apply_transfer(image, coords, k)
    bufs = [band_buffer(p, image, coords, k) for p in plans]
    for g in range(1, world):  # sequential emulation of the one-directional seam exchange
        send = bufs[g - 1][plans[g - 1].send_offset_rows : plans[g - 1].send_offset_rows + plans[g - 1].send_rows]
        assert send.shape[0] == plans[g].recv_rows
        bufs[g][: plans[g].recv_rows] += send
    got = np.concatenate([b[: p.own_rows] for b, p in zip(bufs, plans)])
    assert np.allclose(got, full, rtol=0, atol=1e-9 * np.abs(full).max())
    for p in plans:  # every row a band's patches read (after padding) is inside its resident window
        rows = [r for i in p.patch_index for r in range(coords[i][0], coords[i][0] + n)]
        mapped = pad_rows(np.array(rows), h, "symmetric")
        assert mapped.min() >= p.image_row0 and mapped.max() < p.image_row0 + p.image_rows


@pytest.mark.parametrize("world", [1, 2, 3, 4])
def test_recompute_bands_need_no_exchange(world):
    """seam="recompute": a band runs every patch that reaches into its rows and keeps its own rows only."""
    h, w, n, coords, k, image = small_case()
    plans = make_band_plans(coords, n, h, world, seam="recompute")
    assert sum(p.own_rows for p in plans) == h and all(p.send_rows == 0 and p.recv_rows == 0 for p in plans)
    assert all(p.out_rows == p.own_rows for p in plans)
    full = orc.apply_transfer(image, coords, k)
    got = np.concatenate([band_buffer(p, image, coords, k) for p in plans])
    assert np.allclose(got, full, rtol=0, atol=1e-9 * np.abs(full).max())
    exchange = make_band_plans(coords, n, h, world)
    extra = sum(len(p.patch_index) for p in plans) - sum(len(p.patch_index) for p in exchange)
    assert extra == (world - 1) * len({c for _, c in coords})  # one lattice row recomputed per seam
    for p in plans:
        rows = [r for i in p.patch_index for r in range(coords[i][0], coords[i][0] + n)]
        mapped = pad_rows(np.array(rows), h, "symmetric")
        assert mapped.min() >= p.image_row0 and mapped.max() < p.image_row0 + p.image_rows
    with pytest.raises(ValueError):
        make_band_plans(coords, n, h, world, seam="magic")


def test_headline_lattice_splits_into_eight_even_bands():
    coords = orc.calculate_covering((8 * 4096, 4096), 256)
    plans = make_band_plans(coords, 256, 8 * 4096, 8)
    counts = [len(p.patch_index) for p in plans]
    assert max(counts) / min(counts) < 1.04
    assert all(p.send_rows == 128 for p in plans[:-1]) and all(p.recv_rows == 128 for p in plans[1:])


def test_too_many_ranks_is_an_error():
    h, w, n, coords, _, _ = small_case()
    with pytest.raises(ValueError):
        make_band_plans(coords, n, h, 64)


def _worker(rank, world, port, result_queue):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        h, w, n, coords, k, image = small_case()
        plan = make_band_plans(coords, n, h, world)[rank]
        buf = torch.from_numpy(band_buffer(plan, image, coords, k))
        reqs = []
        if plan.send_rows:
            send = buf[plan.send_offset_rows : plan.send_offset_rows + plan.send_rows].contiguous()
            reqs.append(dist.isend(send, rank + 1))
        if plan.recv_rows:
            recv = torch.empty((plan.recv_rows, w), dtype=buf.dtype)
            dist.recv(recv, rank - 1)
            buf[: plan.recv_rows] += recv
        for r in reqs:
            r.wait()
        full = orc.apply_transfer(image, coords, k)
        own = buf[: plan.own_rows].numpy()
        err = float(np.abs(own - full[plan.out_row0 : plan.out_row0 + plan.own_rows]).max() / np.abs(full).max())
        result_queue.put((rank, err))
    finally:
        dist.destroy_process_group()


def test_seam_exchange_world_size_2_gloo():
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0] < 1e-9 and results[1] < 1e-9


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_band_plan_of_config4(world):
    """BASELINE.json configs[3]: 8192^2 / 256-px patches (65 lattice rows of 65 patches) cut into `world` row bands - what
    `bench.py --gpus N` runs for N > 1.  Whole lattice rows per band, every patch in exactly one band, 128 spill rows
    (N/2) from each band to the next, owned rows tiling the image."""
    import numpy as np

    from regularizepsf_amd.sharding import make_band_plans
    from regularizepsf_amd.util import calculate_covering

    coords = [tuple(int(v) for v in t) for t in calculate_covering((8192, 8192), 256)]
    plans = make_band_plans(coords, 256, 8192, world)
    counts = [len(b.patch_index) for b in plans]
    assert sum(counts) == 4225 and all(c % 65 == 0 for c in counts)
    assert max(counts) - min(counts) <= 65
    assert sorted(i for b in plans for i in b.patch_index) == list(range(4225))
    assert [b.send_rows for b in plans] == [128] * (world - 1) + [0]
    assert [b.recv_rows for b in plans] == [0] + [128] * (world - 1)
    assert sum(b.own_rows for b in plans) == 8192
    assert [b.out_row0 for b in plans] == list(np.cumsum([0] + [b.own_rows for b in plans[:-1]]))
    recompute = make_band_plans(coords, 256, 8192, world, seam="recompute")
    assert all(b.send_rows == 0 and b.recv_rows == 0 and b.out_rows == b.own_rows for b in recompute)
    assert sum(len(b.patch_index) for b in recompute) == 4225 + 65 * (world - 1)  # one shared lattice row per seam
    # ... and the bands are cut so that own + recomputed patches balance: the largest band is as small as whole lattice rows allow
    sizes = [len(b.patch_index) for b in recompute]
    assert max(sizes) == -(-(65 + world - 1) // world) * 65 and sum(b.own_rows for b in recompute) == 8192
    if world == 8:
        assert sizes == [585] * 8 and [b.own_rows for b in recompute] == [1024] * 8
