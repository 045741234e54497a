"""Row-band sharding of one apply across GPUs (one process per GPU).

Patches are independent; the only coupling is the overlap-add across the seam between two bands of
lattice rows (regularizepsf/transform.py:167-169 adds every patch into one image).  A band owns a
contiguous run of lattice rows and the output rows from its first patch row down to the next band's
first patch row.  Its last lattice row spills below that line; the spill (N/2 rows on the regular
lattice) is sent to the next rank, which adds it to the top of its own band.  Nothing flows upwards,
so the exchange is a single one-directional neighbour send/recv - no all-reduce.

This module is geometry only (NumPy); the transport is supplied by the caller: RCCL through
``_native.Comm`` in production, ``torch.distributed`` (gloo) in the CPU tests.

``seam="recompute"`` removes the exchange altogether: a band also processes the last lattice row of the
band above it (every patch that reaches into its rows) and keeps only its own output rows, so the ranks
never talk on the data path.  On the regular lattice that is one extra lattice row per band (+3 % work at
eight bands of the 4096-wide, 256-px configuration) against a 2 MiB send/recv per step.
"""

from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

from regularizepsf_amd import _native


def pad_rows(rows: np.ndarray, height: int, pad_mode: str) -> np.ndarray:
    """Image row that padded row index ``rows`` reads under np.pad ``pad_mode`` (-1: constant fill)."""
    r = np.asarray(rows, dtype=np.int64)
    inside = (r >= 0) & (r < height)
    if pad_mode == "symmetric":
        k = np.mod(r, 2 * height)
        m = np.where(k < height, k, 2 * height - 1 - k)
    elif pad_mode == "reflect":
        if height == 1:
            m = np.zeros_like(r)
        else:
            k = np.mod(r, 2 * height - 2)
            m = np.where(k < height, k, 2 * height - 2 - k)
    elif pad_mode == "edge":
        m = np.clip(r, 0, height - 1)
    elif pad_mode == "wrap":
        m = np.mod(r, height)
    elif pad_mode == "constant":
        m = np.full_like(r, -1)
    else:
        msg = f"pad mode {pad_mode!r} is not evaluated by the kernel"
        raise ValueError(msg)
    return np.where(inside, r, m)


@dataclass
class BandPlan:
    """What one rank holds and exchanges.  Row numbers are rows of the full image / full output."""

    rank: int
    patch_index: list[int]   # indices into the full coordinate list, in their original order
    image_row0: int          # image rows [image_row0, image_row0 + image_rows) must be resident
    image_rows: int
    out_row0: int            # the rank's output buffer holds rows [out_row0, out_row0 + out_rows)
    out_rows: int
    own_rows: int            # of which the first own_rows are final after the exchange
    send_offset_rows: int    # spill = buffer rows [send_offset_rows, send_offset_rows + send_rows) -> rank + 1
    send_rows: int
    recv_rows: int           # rows received from rank - 1, added to buffer rows [0, recv_rows)

    def geometry(self, height: int, width: int, pad_mode: int, pad_value: float = 0.0) -> _native.Geometry:
        return _native.Geometry(height, width, pad_mode, pad_value, 0, 0, self.image_row0, self.image_rows, width,
                                self.out_row0, self.out_rows, width)


def make_band_plans(coordinates, patch_size: int, height: int, world: int, pad_mode: str = "symmetric",
                    seam: str = "exchange") -> list[BandPlan]:
    """Split the patch list into ``world`` bands of whole lattice rows with (almost) equal patch counts.

    ``seam``: "exchange" (a band's spill rows go to the next rank) or "recompute" (a band also runs the patches of
    the band above that reach into its rows; nothing is sent)."""
    if seam not in ("exchange", "recompute"):
        msg = f"seam must be 'exchange' or 'recompute', got {seam!r}"
        raise ValueError(msg)
    coords = np.asarray(coordinates, dtype=np.int64).reshape(-1, 2)
    n = int(patch_size)
    lattice_rows = np.unique(coords[:, 0])
    if world < 1 or world > len(lattice_rows):
        msg = f"cannot split {len(lattice_rows)} lattice rows into {world} bands"
        raise ValueError(msg)
    counts = np.array([(coords[:, 0] == r).sum() for r in lattice_rows])
    cum = np.concatenate([[0], np.cumsum(counts)])
    nrows = len(lattice_rows)
    if seam == "recompute" and world > 1:
        # A band also runs the patches above it that reach into its rows, so what has to be balanced is own + recomputed patches:
        # the cuts that minimise the largest band (dynamic programme over the lattice rows; 65 rows x 8 bands is nothing).  On the
        # 8192-wide, 256-px lattice this gives 9 | 8 + 1 | ... | 8 + 1 rows = 585 patches on every rank instead of 520 ... 650.
        lo = np.maximum(lattice_rows, 0)                               # first own output row of a band that starts at lattice row a
        first_above = np.searchsorted(lattice_rows, lo - n, side="right")  # rows j < a with lattice_rows[j] + n > lo[a] start here
        above = cum[np.arange(nrows)] - cum[np.minimum(first_above, np.arange(nrows))]
        above[0] = 0
        a_idx, b_idx = np.arange(nrows)[:, None], np.arange(nrows + 1)[None, :]
        cost = np.where(b_idx > a_idx, cum[None, :] - cum[:nrows, None] + above[:, None], np.iinfo(np.int64).max)  # [a, b]: band = rows [a, b)
        inf = np.iinfo(np.int64).max
        best = np.full((world + 1, nrows + 1), inf, dtype=np.int64)  # [bands used, rows used] -> largest band so far
        prev = np.zeros((world + 1, nrows + 1), dtype=np.int64)
        best[0, 0] = 0
        for g in range(1, world + 1):
            cand = np.maximum(best[g - 1, :nrows, None], cost)       # [a, b]
            cand[best[g - 1, :nrows] == inf, :] = inf
            prev[g] = np.argmin(cand, axis=0)
            best[g] = cand[prev[g], np.arange(nrows + 1)]
        cuts, b_ = [nrows], nrows
        for g in range(world, 0, -1):
            b_ = int(prev[g, b_])
            cuts.append(b_)
        cuts = cuts[::-1]
    else:
        cuts = [0]
        for g in range(1, world):  # cut where the cumulative patch count is closest to g/world of the total
            target = cum[-1] * g / world
            j = int(np.argmin(np.abs(cum - target)))
            cuts.append(min(max(j, cuts[-1] + 1), nrows - (world - g)))
        cuts.append(nrows)

    first = [int(lattice_rows[cuts[g]]) for g in range(world)]
    last = [int(lattice_rows[cuts[g + 1] - 1]) for g in range(world)]
    own0 = [0] + [min(max(first[g], 0), height) for g in range(1, world)] + [height]
    plans = []
    if seam == "recompute":
        for g in range(world):
            lo_row, hi_row = own0[g], own0[g + 1]
            # every patch whose footprint [r, r + n) meets the band's own output rows (clipped to the image)
            index = [i for i, r in enumerate(coords[:, 0]) if r < hi_row and r + n > lo_row]
            rows = coords[index, 0]
            touched = pad_rows(np.arange(int(rows.min()), int(rows.max()) + n), height, pad_mode)
            touched = touched[touched >= 0]
            lo, hi = (int(touched.min()), int(touched.max()) + 1) if touched.size else (0, 1)
            plans.append(BandPlan(rank=g, patch_index=index, image_row0=lo, image_rows=hi - lo, out_row0=lo_row,
                                  out_rows=hi_row - lo_row, own_rows=hi_row - lo_row, send_offset_rows=hi_row - lo_row,
                                  send_rows=0, recv_rows=0))
        return plans
    for g in range(world):
        index = [i for i, r in enumerate(coords[:, 0]) if first[g] <= r <= last[g]]
        reach_end = min(height, last[g] + n)  # last output row this band's patches touch, exclusive
        if g + 1 < world and reach_end > own0[g + 2]:
            msg = "band too thin: a patch would spill past the next band; use fewer ranks"
            raise ValueError(msg)
        out_row0 = own0[g]
        out_end = max(reach_end, own0[g + 1]) if g + 1 < world else height
        touched = pad_rows(np.arange(first[g], last[g] + n), height, pad_mode)
        touched = touched[touched >= 0]
        lo, hi = (int(touched.min()), int(touched.max()) + 1) if touched.size else (0, 1)
        plans.append(BandPlan(
            rank=g, patch_index=index, image_row0=lo, image_rows=hi - lo, out_row0=out_row0,
            out_rows=out_end - out_row0, own_rows=own0[g + 1] - own0[g],
            send_offset_rows=own0[g + 1] - out_row0, send_rows=out_end - own0[g + 1] if g + 1 < world else 0,
            recv_rows=0))
    for g in range(1, world):
        plans[g].recv_rows = plans[g - 1].send_rows
        if plans[g].recv_rows > plans[g].out_rows:
            msg = "band too thin for the incoming seam rows; use fewer ranks"
            raise ValueError(msg)
    return plans


class ShardedApply:
    """One rank's share of a row-band-sharded apply: local K1 launch(es), then the RCCL seam exchange.

    ``seam="exchange"`` with ``overlap=True`` (default) hides the transfer behind the band's own work: the spill rows depend
    only on the band's LAST lattice row of patches (in general: on every patch that reaches below the band's own rows), so
    those run first, as a plan of their own on a stream of their own, into a seam buffer that holds ALL their output rows: the
    rows below the band's line are sent to the next rank (and the rows from the rank above received) on that stream while the
    main plan - every OTHER patch of the band, output cropped to the rows the band owns - runs on the other; the seam patches'
    rows above the line are added to the band's own rows (K4), and so are the received ones, once both are done.  Every patch
    runs once (until round 4 the seam row ran twice, in both plans: 650 instead of 585 patches per band at eight bands of the
    8192-wide configuration).  The seam buffer is double-buffered by step, so that the seam patches of step k + 1 may start
    while the main plan of step k still runs.  ``overlap=False`` is the plain sequence apply -> exchange -> add on one stream.

    ``overlap="pipeline"`` hides the transfer behind the NEXT step instead: one launch per step (all patches of the band, the
    spill rows at the end of its buffer), and the send / receive / K4 add of step k run on the exchange stream while the launch
    of step k + 1 runs on the main one (two output buffers in turn).  A stream of frames then pays one launch per step - a band
    of 520 patches, fewer than `recompute`'s 585 - and nothing for the link; a single step takes apply + exchange + add in
    sequence.  Measured band by band on one GPU (scripts/band_times.py, profiles/r05m_*): the two-plan form pays for its second
    launch what the overlap saves.
    """

    def __init__(self, coordinates, kernel_for, patch_size: int, height: int, width: int, rank: int, world: int,
                 device: int, comm: "_native.Comm | None", pad_mode: str = "symmetric", seam: str = "exchange",
                 overlap: bool = True) -> None:
        """``kernel_for(index_list)`` returns the (len, N, N) complex64 transfer kernels of those patches."""
        self.height, self.width, self.rank, self.world, self.device = height, width, rank, world, device
        self.seam = seam
        self.band = make_band_plans(coordinates, patch_size, height, world, pad_mode, seam)[rank]
        b = self.band
        coords = [tuple(int(v) for v in coordinates[i]) for i in b.patch_index]
        self.plan = _native.Plan(patch_size, coords, device=device)
        self.plan.set_transfer(kernel_for(b.patch_index))
        self.geometry = b.geometry(height, width, _native.PAD_MODES[pad_mode])
        self.comm = comm
        self.pipeline = overlap == "pipeline" and seam == "exchange" and world > 1 and b.send_rows + b.recv_rows > 0
        self.overlap = bool(overlap and not self.pipeline and seam == "exchange" and world > 1 and b.send_rows + b.recv_rows > 0)
        self.seam_plan = None
        self.seam_once = False
        self.steps = 0
        self.d_img = _native.DeviceBuffer(b.image_rows * width * 4, device)
        self.d_recv = _native.DeviceBuffer(max(1, b.recv_rows) * width * 4, device)
        if self.overlap:
            pm = _native.PAD_MODES[pad_mode]
            if comm is not None:
                # the main plan's persistent workgroups would otherwise hold every CU until their patches are done, and the
                # send/recv kernels of the seam exchange - enqueued beside them - would start only then
                self.plan.set_reserved_cus(8)
            # main plan: output window = the rows this band owns (the spill is the seam plan's business)
            self.geometry = _native.Geometry(height, width, pm, 0.0, 0, 0, b.image_row0, b.image_rows, width,
                                             b.out_row0, b.own_rows, width)
            self.d_out = _native.DeviceBuffer(b.own_rows * width * 4, device)
            if b.send_rows > 0:
                # every patch of the band that reaches below the rows the band owns (on the half-overlap lattice of
                # calculate_covering: the band's last lattice row; with a finer or irregular row spacing earlier rows too)
                own_end = b.out_row0 + b.own_rows
                seam_index = [i for i in b.patch_index if int(coordinates[i][0]) + patch_size > own_end]
                seam_set = set(seam_index)
                rest_index = [i for i in b.patch_index if i not in seam_set]
                seam_top = min(int(coordinates[i][0]) for i in seam_index)
                # The seam patches run ONCE when they are a run of rows at the end of the band: the main plan then takes the other
                # patches, and the seam patches' rows above the band's line come back through K4 (rpsf_add_rows).
                self.seam_once = bool(rest_index) and seam_top >= b.out_row0
                self.seam_plan = _native.Plan(patch_size, [tuple(int(v) for v in coordinates[i]) for i in seam_index], device=device)
                self.seam_plan.set_transfer(kernel_for(seam_index))
                if self.seam_once:
                    self.plan.close()
                    self.plan = _native.Plan(patch_size, [tuple(int(v) for v in coordinates[i]) for i in rest_index], device=device)
                    self.plan.set_transfer(kernel_for(rest_index))
                    if comm is not None:
                        self.plan.set_reserved_cus(8)
                    self.seam_upper_rows = own_end - seam_top
                    self.seam_geometry = _native.Geometry(height, width, pm, 0.0, 0, 0, b.image_row0, b.image_rows, width,
                                                          seam_top, self.seam_upper_rows + b.send_rows, width)
                    rows = self.seam_upper_rows + b.send_rows
                    self.d_seam = [_native.DeviceBuffer(rows * width * 4, device) for _ in range(2)]
                    self.seam_read = [_native.Event(device) for _ in range(2)]  # "the add of the step that used this buffer has read it"
                    self.d_spill = None
                else:
                    self.seam_geometry = _native.Geometry(height, width, pm, 0.0, 0, 0, b.image_row0, b.image_rows, width,
                                                          b.out_row0 + b.own_rows, b.send_rows, width)
                    self.d_spill = _native.DeviceBuffer(b.send_rows * width * 4, device)
        elif self.pipeline:
            # The exchange of step k runs beside the persistent launch of step k + 1, whose workgroups need whole CUs: the chip is
            # partitioned - the plan's stream keeps all but `carve` CUs, the exchange stream (RCCL's send / recv kernels, K4) gets those
            carve = int(os.environ.get("RPSF_EXCHANGE_CUS", "8"))
            cus, _ = _native.device_info(device)
            self._xstream = None
            if carve > 0:
                main_mask, side_mask = _native.cu_masks(cus, carve)
                self.plan.set_cu_mask(main_mask)
                self._xstream = _native.Stream(device, side_mask)
            else:
                if comm is not None:
                    self.plan.set_reserved_cus(8)
                self._xstream = _native.Stream(device)
            self.xstream = self._xstream.ptr
            self.d_outs = [_native.DeviceBuffer(b.out_rows * width * 4, device) for _ in range(2)]
            self.d_out = self.d_outs[0]
            self.ev_applied = [_native.Event(device) for _ in range(2)]   # the launch of the step that used this buffer is done
            self.ev_exchanged = [_native.Event(device) for _ in range(2)]  # ... and so are its send, its receive and its add
        else:
            self.d_out = _native.DeviceBuffer(b.out_rows * width * 4, device)

    def upload_rows(self, band_image: np.ndarray) -> None:
        """``band_image`` = image rows [image_row0, image_row0 + image_rows) as float32."""
        if band_image.shape != (self.band.image_rows, self.width):
            msg = f"expected image rows of shape {(self.band.image_rows, self.width)}, got {band_image.shape}"
            raise ValueError(msg)
        self.d_img.upload(np.ascontiguousarray(band_image, np.float32))

    def step(self) -> None:
        """Enqueue one apply + seam exchange (asynchronous; ``synchronize`` waits for all of it)."""
        b, w = self.band, self.width
        if self.pipeline:
            slot = self.steps & 1
            self.steps += 1
            out = self.d_out = self.d_outs[slot]
            self.ev_exchanged[slot].make_wait(self.plan.stream)  # the exchange of two steps ago has sent from / added into this buffer
            self.plan.apply_device(self.d_img.ptr, out.ptr, self.geometry)
            self.ev_applied[slot].record(self.plan.stream)
            self.ev_applied[slot].make_wait(self.xstream)
            if self.comm is not None:
                # (d_recv is safe to overwrite: the previous step's add ran on this same stream)
                self.comm.seam_exchange(out.at(b.send_offset_rows * w * 4), b.send_rows * w, self.d_recv.ptr, b.recv_rows * w, self.xstream)
                if b.recv_rows > 0:
                    # K4 on the CUs the persistent launch of the next step leaves free (a patch workgroup needs a whole CU: a thousand
                    # small workgroups dispatched a moment before it cost that launch 16 us, profiles/r05m)
                    _native.add_rows(out.ptr, self.d_recv.ptr, b.recv_rows * w, self.device, self.xstream,
                                     max_workgroups=int(os.environ.get("RPSF_EXCHANGE_WGS", "64")))
            self.ev_exchanged[slot].record(self.xstream)
            return
        if not self.overlap:
            self.plan.apply_device(self.d_img.ptr, self.d_out.ptr, self.geometry)
            if self.comm is not None and self.world > 1 and b.send_rows + b.recv_rows > 0:
                self.comm.seam_exchange_add(self.d_out.at(b.send_offset_rows * w * 4), b.send_rows * w,
                                            self.d_recv.ptr, b.recv_rows * w, self.d_out.ptr, self.plan.stream)
            return
        xstream = self.seam_plan.stream if self.seam_plan is not None else (self.comm.stream if self.comm is not None else None)
        slot = self.steps & 1
        self.steps += 1
        if self.seam_plan is not None:
            # First in line, and not ordered against the main stream's CURRENT work: the seam patches of step k + 1 may start while
            # the main plan of step k still runs - they take the CUs it leaves free.  Two persistent launches side by side are safe:
            # every slot of both is drawn from a queue by whichever workgroups are resident (rpsf.hip, launch_patches).  What the
            # launch does have to wait for is the K4 add of two steps ago, which read the buffer it is about to overwrite (seam rows
            # computed once: two buffers in turn); the spill buffer of the older scheme is only ever touched on this stream.
            if self.seam_once:
                self.seam_read[slot].make_wait(xstream)
                self.seam_plan.apply_device(self.d_img.ptr, self.d_seam[slot].ptr, self.seam_geometry)
            else:
                self.seam_plan.apply_device(self.d_img.ptr, self.d_spill.ptr, self.seam_geometry)
        if xstream is not None and self.comm is not None and b.recv_rows > 0:
            # ... but the receive must not land before the previous step's add has read d_recv: the exchange waits for
            # everything enqueued on the main stream so far (its tail is that add; this step's main apply comes after this line)
            _native.stream_wait(xstream, self.plan.stream, self.device)
        self.plan.apply_device(self.d_img.ptr, self.d_out.ptr, self.geometry)
        if self.seam_once:
            # the seam patches' rows above the band's line: onto the band's own last rows, behind the main plan (the event is recorded
            # on the seam stream here, behind the seam apply and in front of the exchange: the add does not wait for the link)
            _native.stream_wait(self.plan.stream, xstream, self.device)
            upper = self.seam_upper_rows
            _native.add_rows(self.d_out.at((b.own_rows - upper) * w * 4), self.d_seam[slot].ptr, upper * w, self.device, self.plan.stream)
            self.seam_read[slot].record(self.plan.stream)
        if self.comm is None:
            return
        if self.seam_plan is None:
            send_ptr = None
        elif self.seam_once:
            send_ptr = self.d_seam[slot].at(self.seam_upper_rows * w * 4)
        else:
            send_ptr = self.d_spill.ptr
        self.comm.seam_exchange(send_ptr, b.send_rows * w, self.d_recv.ptr, b.recv_rows * w, xstream)
        if b.recv_rows > 0:
            if xstream is not None:
                _native.stream_wait(self.plan.stream, xstream, self.device)
            _native.add_rows(self.d_out.ptr, self.d_recv.ptr, b.recv_rows * w, self.device, self.plan.stream)

    def spill_ptr(self):
        """Device pointer of the rows the last ``step`` left for the next band (``send_rows * width`` floats)."""
        b, w = self.band, self.width
        if not self.overlap:
            return self.d_out.at(b.send_offset_rows * w * 4)
        if self.seam_plan is None:
            return None
        if self.seam_once:
            return self.d_seam[(self.steps - 1) & 1].at(self.seam_upper_rows * w * 4)
        return self.d_spill.ptr

    def spill_rows(self) -> np.ndarray:
        """The rows this band's last lattice row leaves for the next band (after ``step`` + ``synchronize``)."""
        b, w = self.band, self.width
        self.synchronize()
        if self.overlap:
            if self.seam_plan is None:
                return np.zeros((0, w), np.float32)
            if self.seam_once:
                last = self.d_seam[(self.steps - 1) & 1]
                return last.download((b.send_rows, w), offset_bytes=self.seam_upper_rows * w * 4)
            return self.d_spill.download((b.send_rows, w))
        return self.d_out.download((b.send_rows, w), offset_bytes=b.send_offset_rows * w * 4)

    def synchronize(self) -> None:
        self.plan.synchronize()  # device-wide: covers the seam plan's and the communicator's streams too

    def owned_rows(self) -> np.ndarray:
        """The final output rows this rank owns: rows [out_row0, out_row0 + own_rows) of the full result."""
        self.synchronize()
        return self.d_out.download((self.band.own_rows, self.width))
