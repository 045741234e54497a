"""Minimal HDF5 reader/writer for the files regularizepsf's ``save`` methods produce.

The reference stores an ``ArrayPSFTransform`` / ``ArrayPSF`` as an ``.h5`` file with a handful of plain
datasets in the root group (``regularizepsf/transform.py:236-240,266-270``, ``psf.py:276-280,307-313``):
``coordinates`` (n, 2) int64 and ``transfer_kernel`` / ``values`` / ``fft_evaluations`` (n, N, N) real or
complex, written by h5py with default settings - i.e. the oldest, simplest on-disk structures: version-0
superblock, version-1 object headers, a symbol-table root group (one B-tree leaf, a local heap, symbol
nodes), contiguous un-filtered dataset layout, complex numbers as the compound ``{r, i}``.  h5py is not
available in the target image, so this module reads and writes exactly that subset (HDF5 File Format
Specification version 1.1/2.0, sections III.A-III.E, IV.A.1-IV.A.2) with nothing but ``struct`` and
NumPy.  Anything outside the subset (chunked or compressed datasets, new-style groups, other type
classes) raises ``NotImplementedError`` naming the feature - use h5py for such files.

The writer's output is checked against libhdf5 itself (h5py, where an interpreter that has it exists)
and the reader against files written by the reference through h5py (``tests/golden/h5_*.h5``).
"""

from __future__ import annotations

import pathlib
import struct
import time

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
GROUP_LEAF_K, GROUP_INTERNAL_K = 4, 16  # library defaults, as h5py writes them


class H5FormatError(ValueError):
    """The file is not HDF5 or is damaged."""


# ---------------------------------------------------------------------------------------------- datatypes
def _float_type(size: int) -> bytes:
    if size == 4:
        return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    if size == 8:
        return struct.pack("<BBBBI", 0x11, 0x20, 0x3F, 0x00, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    msg = f"float size {size}"
    raise NotImplementedError(msg)


def _int_type(size: int, signed: bool) -> bytes:
    return struct.pack("<BBBBI", 0x10, 0x08 if signed else 0x00, 0, 0, size) + struct.pack("<HH", 0, 8 * size)


def _encode_dtype(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.byteorder == ">":
        msg = "big-endian arrays are not supported; convert with .astype(dtype.newbyteorder('<'))"
        raise NotImplementedError(msg)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        return _float_type(dt.itemsize)
    if dt.kind in "iu" and dt.itemsize in (1, 2, 4, 8):
        return _int_type(dt.itemsize, dt.kind == "i")
    if dt.kind == "c" and dt.itemsize in (8, 16):  # h5py's convention: compound {r, i}
        half = dt.itemsize // 2
        out = struct.pack("<BBBBI", 0x16, 2, 0, 0, dt.itemsize)
        for name, offset in ((b"r", 0), (b"i", half)):
            out += name.ljust(8, b"\0") + struct.pack("<IB3xI4x4I", offset, 0, 0, 0, 0, 0, 0) + _float_type(half)
        return out
    msg = f"dtype {dt} is outside the subset this writer supports"
    raise NotImplementedError(msg)


def _decode_dtype(buf: bytes, pos: int = 0) -> tuple[np.dtype, int]:
    """Return (numpy dtype, bytes consumed) for the datatype message starting at buf[pos]."""
    cls_ver, b0, b1, _b2, size = struct.unpack_from("<BBBBI", buf, pos)
    cls, version = cls_ver & 0x0F, cls_ver >> 4
    order = ">" if b0 & 1 else "<"
    if cls == 0:  # fixed point
        return np.dtype(f"{order}{'i' if b0 & 0x08 else 'u'}{size}"), 8 + 4
    if cls == 1:  # floating point
        if size not in (2, 4, 8):
            msg = f"{size}-byte floating point"
            raise NotImplementedError(msg)
        return np.dtype(f"{order}f{size}"), 8 + 12
    if cls == 6:  # compound: only the complex-number convention {r, i}
        n_members = b0 | (b1 << 8)
        p = pos + 8
        members = []
        for _ in range(n_members):
            end = buf.index(b"\0", p)
            name = buf[p:end].decode()
            if version < 3:
                p += (end - p + 8) // 8 * 8  # name padded to a multiple of 8 including the terminator
            else:
                p = end + 1
            if version == 1:
                (offset,) = struct.unpack_from("<I", buf, p)
                p += 4 + 1 + 3 + 4 + 4 + 16
            elif version == 2:
                (offset,) = struct.unpack_from("<I", buf, p)
                p += 4
            else:
                nbytes = 1 if size < 256 else 2 if size < 65536 else 4
                offset = int.from_bytes(buf[p:p + nbytes], "little")
                p += nbytes
            mdt, used = _decode_dtype(buf, p)
            p += used
            members.append((name, offset, mdt))
        names = [m[0] for m in members]
        if names == ["r", "i"] and members[0][2] == members[1][2] and members[0][2].kind == "f" \
                and members[0][1] == 0 and members[1][1] == members[0][2].itemsize and size == 2 * members[0][2].itemsize:
            base = members[0][2]
            return np.dtype(f"{base.byteorder if base.byteorder != '=' else '<'}c{size}"), p - pos
        return np.dtype({"names": names, "formats": [m[2] for m in members], "offsets": [m[1] for m in members],
                         "itemsize": size}), p - pos
    msg = f"HDF5 datatype class {cls}"
    raise NotImplementedError(msg)


# ---------------------------------------------------------------------------------------------- writer
def _message(kind: int, body: bytes) -> bytes:
    body = body + b"\0" * (-len(body) % 8)
    return struct.pack("<HHB3x", kind, len(body), 0) + body


def _object_header(messages: list[bytes], min_space: int = 0) -> bytes:
    body = b"".join(messages)
    n = len(messages)
    if len(body) + 8 <= min_space:  # pad with one NIL message, like the library does
        body += _message(0x0000, b"\0" * (min_space - len(body) - 8))
        n += 1
    return struct.pack("<BxHII4x", 1, n, 1, len(body)) + body


def _dataset_header(arr: np.ndarray, data_address: int, mtime: int) -> bytes:
    rank = arr.ndim
    dims = struct.pack(f"<{rank}Q", *arr.shape)
    dataspace = struct.pack("<BBB5x", 1, rank, 1) + dims + dims
    fill = struct.pack("<BBBBI", 2, 2, 2, 1, 0)
    layout = struct.pack("<BBQQ", 3, 1, data_address if arr.nbytes else UNDEF, arr.nbytes)
    return _object_header([
        _message(0x0001, dataspace), _message(0x0003, _encode_dtype(arr.dtype)), _message(0x0005, fill),
        _message(0x0008, layout), _message(0x0012, struct.pack("<B3xI", 1, mtime & 0xFFFFFFFF)),
    ], min_space=256)


def write_datasets(path, datasets: dict[str, np.ndarray], overwrite: bool = False) -> None:
    """Write ``datasets`` (name -> array) as contiguous datasets of the root group of a new HDF5 file.

    ``overwrite=False`` raises ``FileExistsError`` for an existing file (h5py's mode ``"w-"``, which the
    reference uses, ``transform.py:236``)."""
    path = pathlib.Path(path)
    if len(datasets) > 2 * GROUP_LEAF_K:
        msg = f"at most {2 * GROUP_LEAF_K} datasets per file in this minimal writer"
        raise NotImplementedError(msg)
    arrays = {}
    for name, value in datasets.items():
        if not name or "/" in name or "\0" in name:
            msg = f"bad dataset name {name!r}"
            raise ValueError(msg)
        arrays[name] = np.require(value, requirements="C")
    names = sorted(arrays, key=lambda s: s.encode())  # symbol nodes are ordered by strcmp
    mtime = int(time.time())

    # local heap data segment: offset 0 = empty string (the B-tree's first key), then the names, 8-aligned
    heap = bytearray(8)
    name_offset = {}
    for name in names:
        name_offset[name] = len(heap)
        raw = name.encode() + b"\0"
        heap += raw + b"\0" * (-len(raw) % 8)
    seg_size = max(88, len(heap) + 16)
    free_head = len(heap)
    heap += struct.pack("<QQ", 1, seg_size - free_head)  # one free block: next = none (1), its size
    heap += b"\0" * (seg_size - len(heap))

    root_oh_at = 0x60
    root_oh_len = 16 + 8 + 16
    btree_at = root_oh_at + root_oh_len
    btree_len = 24 + (2 * GROUP_INTERNAL_K + 1) * 8 + 2 * GROUP_INTERNAL_K * 8
    heap_at = btree_at + btree_len
    heap_data_at = heap_at + 32
    snod_at = heap_data_at + seg_size
    snod_len = 8 + 2 * GROUP_LEAF_K * 40
    pos = snod_at + snod_len
    header_at, headers = {}, {}
    probe = {n: _dataset_header(arrays[n], 0, mtime) for n in names}  # sizes do not depend on the address
    for n in names:
        header_at[n] = pos
        pos += len(probe[n])
    data_at = {}
    for n in names:
        pos += -pos % 8
        data_at[n] = pos
        pos += arrays[n].nbytes
    eof = pos
    for n in names:
        headers[n] = _dataset_header(arrays[n], data_at[n], mtime)

    out = bytearray()
    out += SIGNATURE + struct.pack("<BBBBBBBxHHI", 0, 0, 0, 0, 0, 8, 8, GROUP_LEAF_K, GROUP_INTERNAL_K, 0)
    out += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    out += struct.pack("<QQI4xQQ", 0, root_oh_at, 1, btree_at, heap_at)  # root symbol-table entry (cached B-tree/heap)
    assert len(out) == root_oh_at
    out += _object_header([_message(0x0011, struct.pack("<QQ", btree_at, heap_at))])
    assert len(out) == btree_at
    node = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF)
    node += struct.pack("<QQQ", 0, snod_at, name_offset[names[-1]] if names else 0)
    out += node + b"\0" * (btree_len - len(node))
    out += b"HEAP" + struct.pack("<B3xQQQ", 0, seg_size, free_head, heap_data_at) + heap
    assert len(out) == snod_at
    snod = b"SNOD" + struct.pack("<BxH", 1, len(names))
    for n in names:
        snod += struct.pack("<QQI4x16x", name_offset[n], header_at[n], 0)
    out += snod + b"\0" * (snod_len - len(snod))
    for n in names:
        assert len(out) == header_at[n]
        out += headers[n]
    for n in names:
        out += b"\0" * (data_at[n] - len(out))
        out += arrays[n].tobytes()
    assert len(out) == eof
    with open(path, "wb" if overwrite else "xb") as f:
        f.write(out)


# ---------------------------------------------------------------------------------------------- reader
class _Reader:
    def __init__(self, data: bytes) -> None:
        self.d = data
        for base in [0] + [512 << i for i in range(12)]:  # the superblock may follow a user block
            if data[base:base + 8] == SIGNATURE:
                self.base = base
                break
        else:
            msg = "not an HDF5 file (signature not found)"
            raise H5FormatError(msg)
        version = data[self.base + 8]
        if version not in (0, 1):
            msg = f"superblock version {version} (file written with a newer format); open it with h5py"
            raise NotImplementedError(msg)
        so, sl = data[self.base + 13], data[self.base + 14]
        if (so, sl) != (8, 8):
            msg = f"{so}-byte offsets / {sl}-byte lengths"
            raise NotImplementedError(msg)
        p = self.base + 24 + (4 if version == 1 else 0)
        self.base_address = struct.unpack_from("<Q", data, p)[0]
        root_entry = p + 32
        _, self.root_header, cache_type = struct.unpack_from("<QQI", data, root_entry)

    def at(self, address: int) -> int:
        if address == UNDEF or address + self.base_address >= len(self.d):
            msg = "address outside the file"
            raise H5FormatError(msg)
        return address + self.base_address

    def messages(self, header_address: int):
        """Yield (type, body bytes) of a version-1 object header, following continuation blocks."""
        p = self.at(header_address)
        if self.d[p:p + 4] == b"OHDR":
            msg = "version-2 object headers (file written with libver='latest'); open it with h5py"
            raise NotImplementedError(msg)
        version, n_msgs, _ref, size = struct.unpack_from("<BxHII", self.d, p)
        if version != 1:
            msg = f"object header version {version}"
            raise H5FormatError(msg)
        blocks = [(p + 16, size)]
        seen = 0
        while blocks and seen < n_msgs:
            q, left = blocks.pop(0)
            end = q + left
            while q + 8 <= end and seen < n_msgs:
                kind, length, flags = struct.unpack_from("<HHB", self.d, q)
                body = self.d[q + 8:q + 8 + length]
                q += 8 + length
                seen += 1
                if kind == 0x0010:  # continuation
                    off, ln = struct.unpack_from("<QQ", body)
                    blocks.append((self.at(off), ln))
                else:
                    yield kind, body

    def group_entries(self, btree: int, heap: int) -> dict[str, int]:
        hp = self.at(heap)
        if self.d[hp:hp + 4] != b"HEAP":
            msg = "local heap signature missing"
            raise H5FormatError(msg)
        heap_data = self.at(struct.unpack_from("<Q", self.d, hp + 24)[0])
        out: dict[str, int] = {}

        def walk(node: int) -> None:
            p = self.at(node)
            if self.d[p:p + 4] == b"SNOD":
                (count,) = struct.unpack_from("<H", self.d, p + 6)
                for i in range(count):
                    name_off, header = struct.unpack_from("<QQ", self.d, p + 8 + 40 * i)
                    s = heap_data + name_off
                    out[self.d[s:self.d.index(b"\0", s)].decode()] = header
                return
            if self.d[p:p + 4] != b"TREE":
                msg = "B-tree node signature missing"
                raise H5FormatError(msg)
            _type, _level, used = struct.unpack_from("<BBH", self.d, p + 4)
            for i in range(used):
                (child,) = struct.unpack_from("<Q", self.d, p + 24 + 8 + 16 * i)
                walk(child)

        walk(btree)
        return out

    def root(self) -> dict[str, int]:
        for kind, body in self.messages(self.root_header):
            if kind == 0x0011:
                btree, heap = struct.unpack_from("<QQ", body)
                return self.group_entries(btree, heap)
            if kind in (0x0002, 0x0006):
                msg = "new-style (link message) groups; open the file with h5py"
                raise NotImplementedError(msg)
        msg = "root group has no symbol table"
        raise H5FormatError(msg)

    def dataset(self, header: int) -> np.ndarray:
        shape = dtype = None
        raw = None
        for kind, body in self.messages(header):
            if kind == 0x0001:
                version, rank = body[0], body[1]
                start = 8 if version == 1 else 4
                shape = struct.unpack_from(f"<{rank}Q", body, start)
            elif kind == 0x0003:
                dtype, _ = _decode_dtype(body)
            elif kind == 0x000B:
                msg = "filtered (compressed) datasets; open the file with h5py"
                raise NotImplementedError(msg)
            elif kind == 0x0008:
                version = body[0]
                if version == 3:
                    cls = body[1]
                    if cls == 1:
                        address, size = struct.unpack_from("<QQ", body, 2)
                        raw = b"" if address == UNDEF else self.d[self.at(address):self.at(address) + size]
                    elif cls == 0:
                        (size,) = struct.unpack_from("<H", body, 2)
                        raw = body[4:4 + size]
                    else:
                        msg = "chunked datasets; open the file with h5py"
                        raise NotImplementedError(msg)
                elif version in (1, 2):
                    rank, cls = body[1], body[2]
                    if cls != 1:
                        msg = "non-contiguous dataset layout; open the file with h5py"
                        raise NotImplementedError(msg)
                    (address,) = struct.unpack_from("<Q", body, 8)
                    dims = struct.unpack_from(f"<{rank}I", body, 16)
                    size = int(np.prod(dims[:-1], dtype=np.int64)) * dims[-1] if rank else 0
                    raw = b"" if address == UNDEF else self.d[self.at(address):self.at(address) + size]
                else:
                    msg = f"data layout message version {version}; open the file with h5py"
                    raise NotImplementedError(msg)
        if shape is None or dtype is None or raw is None:
            msg = "object is not a simple dataset"
            raise NotImplementedError(msg)
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if len(raw) < count * dtype.itemsize:
            if len(raw) == 0:  # never written: HDF5 returns the fill value (zero)
                return np.zeros(shape, dtype.newbyteorder("=") if dtype.fields is None else dtype)
            msg = "dataset is truncated"
            raise H5FormatError(msg)
        arr = np.frombuffer(raw, dtype=dtype, count=count).reshape(shape)
        return arr.astype(dtype.newbyteorder("="), copy=True) if dtype.fields is None else arr.copy()


def read_datasets(path, names: list[str] | None = None) -> dict[str, np.ndarray]:
    """Read the named (default: all) datasets of the root group into native-endian arrays."""
    data = pathlib.Path(path).read_bytes()
    reader = _Reader(data)
    entries = reader.root()
    wanted = list(entries) if names is None else names
    out = {}
    for name in wanted:
        if name not in entries:
            msg = f"Unable to open object (object '{name}' doesn't exist)"
            raise KeyError(msg)
        out[name] = reader.dataset(entries[name])
    return out
