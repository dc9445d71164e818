"""A small pure-Python HDF5 reader/writer for the files on this path's boundary.

The reference reads its inputs and writes its dumps with ``h5py``
(annealing_sign_problem/common.py:750-780): ground states produced by SpinED
(``/hamiltonian/eigenvectors``, ``/hamiltonian/eigenvalues``,
``/basis/representatives``) and Ising-model dumps (``elements f64``, ``indices i32``,
``indptr i32``, ``field f64``, ``energy f64``, ``signs u64`` at the root).  ``h5py`` is not part
of this image's main interpreter, so the subset of the format those files use is implemented
here, following the published HDF5 File Format Specification (version 1.x objects):

  writer  superblock v0, "old style" groups (symbol table: v1 B-tree + local heap + SNODs),
          v1 object headers, contiguous little-endian integer / IEEE float datasets of any
          rank (scalars included) — byte-for-byte the structure h5py's default settings
          produce, so h5py / libhdf5 open the files;
  reader  superblock v0/v1, v1 object headers with continuation blocks, symbol-table groups
          (multi-level B-trees), contiguous, compact and chunked layouts (v1 chunk B-tree),
          deflate and shuffle filters, fixed-point and floating-point types of either byte order.

``common.py`` prefers ``h5py`` when it is importable and falls back to this module; the tests
cross-check the two wherever an interpreter with h5py exists (tests/test_hdf5.py).
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, List, Tuple, Union

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"
Tree = Dict[str, Union[np.ndarray, "Tree"]]


# ------------------------------------------------------------------------------------------
# writer
# ------------------------------------------------------------------------------------------

def _pad8(b: bytes) -> bytes:
    return b + b"\x00" * (-len(b) % 8)


def _datatype_message(dtype: np.dtype) -> bytes:
    dtype = np.dtype(dtype)
    if dtype.byteorder == ">":
        raise ValueError("big-endian arrays are not written")
    if dtype.kind in "iu":
        bits = 0x08 if dtype.kind == "i" else 0x00
        return struct.pack("<BBBBIHH", 0x10, bits, 0, 0, dtype.itemsize, 0, 8 * dtype.itemsize)
    if dtype == np.float64:
        return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 63, 0, 8, 0, 64, 52, 11, 0, 52, 1023)
    if dtype == np.float32:
        return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 31, 0, 4, 0, 32, 23, 8, 0, 23, 127)
    raise ValueError("unsupported dtype {}".format(dtype))


def _message(kind: int, body: bytes, flags: int = 0) -> bytes:
    body = _pad8(body)
    return struct.pack("<HHBBBB", kind, len(body), flags, 0, 0, 0) + body


def _object_header(messages: List[bytes]) -> bytes:
    data = b"".join(messages)
    return struct.pack("<BBHII", 1, 0, len(messages), 1, len(data)) + b"\x00" * 4 + data


class _Writer:
    LEAF_K, INTERNAL_K = 4, 16

    def __init__(self):
        self.buf = bytearray(96)  # the superblock is filled in last

    def _alloc(self, data: bytes) -> int:
        self.buf += b"\x00" * (-len(self.buf) % 8)
        at = len(self.buf)
        self.buf += data
        return at

    def dataset(self, value) -> int:
        a = np.asarray(value)
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)
        if a.dtype.byteorder == ">":
            a = a.astype(a.dtype.newbyteorder("<"))
        shape = a.shape  # (ascontiguousarray would turn a scalar into a 1-element vector)
        raw = a.tobytes(order="C")
        at = self._alloc(raw) if raw else UNDEF
        if len(shape) == 0:
            space = struct.pack("<BBBBI", 1, 0, 0, 0, 0)
        else:
            space = struct.pack("<BBBBI", 1, len(shape), 1, 0, 0) + struct.pack(
                "<%dQ" % (2 * len(shape)), *(list(shape) * 2))
        messages = [
            _message(0x0001, space),
            _message(0x0003, _datatype_message(a.dtype), flags=1),
            _message(0x0005, struct.pack("<BBBBI", 2, 2, 2, 1, 0), flags=1),
            _message(0x0008, struct.pack("<BBQQ", 3, 1, at, len(raw))),
        ]
        return self._alloc(_object_header(messages))

    def group(self, tree: Tree) -> Tuple[int, int, int]:
        """Returns (object header, B-tree, heap) addresses."""
        names = sorted(tree, key=lambda s: s.encode())
        if len(names) > 2 * self.LEAF_K * 2 * self.INTERNAL_K:
            raise ValueError("too many links in one group")
        entries = []
        for name in names:
            if not name or "/" in name:
                raise ValueError("invalid link name {!r}".format(name))
            child = tree[name]
            if isinstance(child, dict):
                header, btree, heap = self.group(child)
                entries.append((name, header, 1, struct.pack("<QQ", btree, heap)))
            else:
                entries.append((name, self.dataset(child), 0, b"\x00" * 16))
        # local heap: the empty string at offset 0 (key of the B-tree's left edge), then the names
        heap_data = bytearray(8)
        offsets = {}
        for name in names:
            offsets[name] = len(heap_data)
            heap_data += _pad8(name.encode() + b"\x00")
        heap_data += struct.pack("<QQ", 1, 16)  # one free block, end of the free list
        free_at = len(heap_data) - 16
        heap_addr = self._alloc(b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(heap_data), free_at, 0))
        data_addr = self._alloc(bytes(heap_data))
        struct.pack_into("<Q", self.buf, heap_addr + 24, data_addr)
        # symbol table nodes of up to 2 * LEAF_K entries, in name order
        per_node = 2 * self.LEAF_K
        children, keys = [], [0]
        for start in range(0, max(len(entries), 1), per_node):
            chunk = entries[start:start + per_node]
            node = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(chunk)))
            for name, header, cache, scratch in chunk:
                node += struct.pack("<QQII", offsets[name], header, cache, 0) + scratch
            node += b"\x00" * (8 + 40 * per_node - len(node))
            children.append(self._alloc(bytes(node)))
            keys.append(offsets[chunk[-1][0]] if chunk else 0)
        tree_node = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, len(children), UNDEF, UNDEF))
        for i, child in enumerate(children):
            tree_node += struct.pack("<QQ", keys[i], child)
        tree_node += struct.pack("<Q", keys[len(children)])
        tree_node += b"\x00" * (24 + 8 * (4 * self.INTERNAL_K + 1) - len(tree_node))
        btree_addr = self._alloc(bytes(tree_node))
        header = self._alloc(_object_header([_message(0x0011, struct.pack("<QQ", btree_addr, heap_addr))]))
        return header, btree_addr, heap_addr

    def finish(self, root: Tuple[int, int, int]) -> bytes:
        header, btree, heap = root
        self.buf += b"\x00" * (-len(self.buf) % 8)
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K,
                                     self.INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQII", 0, header, 1, 0) + struct.pack("<QQ", btree, heap)
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def write(filename: str, tree: Tree) -> None:
    """Write nested dicts of arrays / scalars as groups and datasets."""
    w = _Writer()
    data = w.finish(w.group(tree))
    with open(filename, "wb") as f:
        f.write(data)


# ------------------------------------------------------------------------------------------
# reader
# ------------------------------------------------------------------------------------------

class _Reader:
    def __init__(self, data: bytes):
        self.d = data
        if data[:8] != SIGNATURE:
            raise ValueError("not an HDF5 file (signature at offset 0 missing)")
        version = data[8]
        if version not in (0, 1):
            raise ValueError("HDF5 superblock version {} is not supported (only 0 and 1: files "
                             "written with the library's default format)".format(version))
        if data[13] != 8 or data[14] != 8:
            raise ValueError("only 8-byte offsets and lengths are supported")
        at = 24 + (4 if version == 1 else 0)
        self.base = struct.unpack_from("<Q", data, at)[0]
        self.root_header = struct.unpack_from("<Q", data, at + 32 + 8)[0]

    # -- object headers ---------------------------------------------------------------------
    def messages(self, addr: int) -> List[Tuple[int, bytes]]:
        d = self.d
        addr += self.base
        version, _, count, _, size = struct.unpack_from("<BBHII", d, addr)
        if version != 1:
            raise ValueError("object header version {} is not supported".format(version))
        out: List[Tuple[int, bytes]] = []
        blocks = [(addr + 16, size)]
        while blocks and len(out) < count:
            at, left = blocks.pop(0)
            while left >= 8 and len(out) < count:
                kind, length, _flags = struct.unpack_from("<HHB", d, at)
                body = d[at + 8: at + 8 + length]
                if kind == 0x0010:  # continuation
                    more, more_size = struct.unpack_from("<QQ", body)
                    blocks.append((more + self.base, more_size))
                out.append((kind, body))
                at += 8 + length
                left -= 8 + length
        return out

    # -- groups -------------------------------------------------------------------------------
    def _heap_name(self, heap_data: int, offset: int) -> str:
        end = self.d.index(b"\x00", heap_data + offset)
        return self.d[heap_data + offset:end].decode()

    def _group_entries(self, btree: int, heap: int):
        d = self.d
        if d[heap + self.base: heap + self.base + 4] != b"HEAP":
            raise ValueError("local heap signature missing")
        heap_data = struct.unpack_from("<Q", d, heap + self.base + 24)[0] + self.base

        def walk(node):
            node += self.base
            if d[node:node + 4] != b"TREE":
                raise ValueError("B-tree signature missing")
            kind, level, used = struct.unpack_from("<BBH", d, node + 4)
            if kind != 0:
                raise ValueError("expected a group B-tree")
            for i in range(used):
                child = struct.unpack_from("<Q", d, node + 24 + 8 + 16 * i)[0]
                if level > 0:
                    yield from walk(child)
                    continue
                child += self.base
                if d[child:child + 4] != b"SNOD":
                    raise ValueError("symbol table node signature missing")
                symbols = struct.unpack_from("<H", d, child + 6)[0]
                for s in range(symbols):
                    name_at, header, cache = struct.unpack_from("<QQI", d, child + 8 + 40 * s)
                    yield self._heap_name(heap_data, name_at), header

        return list(walk(btree))

    def read_object(self, header: int):
        kinds = dict()
        for kind, body in self.messages(header):
            kinds.setdefault(kind, body)
        if 0x0011 in kinds:  # symbol table message: a group
            btree, heap = struct.unpack_from("<QQ", kinds[0x0011])
            return {name: self.read_object(child) for name, child in self._group_entries(btree, heap)}
        if 0x0008 not in kinds:
            raise ValueError("object is neither an old-style group nor a dataset")
        return self._dataset(kinds)

    # -- datasets -----------------------------------------------------------------------------
    @staticmethod
    def _dtype(body: bytes) -> np.dtype:
        cls, version = body[0] & 0x0F, body[0] >> 4
        size = struct.unpack_from("<I", body, 4)[0]
        order = ">" if body[1] & 1 else "<"
        if cls == 0:
            return np.dtype("{}{}{}".format(order, "i" if body[1] & 0x08 else "u", size))
        if cls == 1 and size in (4, 8):
            return np.dtype("{}f{}".format(order, size))
        raise ValueError("datatype class {} (version {}) of size {} is not supported".format(
            cls, version, size))

    @staticmethod
    def _shape(body: bytes) -> Tuple[int, ...]:
        version, rank = body[0], body[1]
        at = 8 if version == 1 else 4
        return struct.unpack_from("<%dQ" % rank, body, at) if rank else ()

    def _dataset(self, kinds) -> np.ndarray:
        dtype = self._dtype(kinds[0x0003])
        shape = self._shape(kinds[0x0001])
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        layout = kinds[0x0008]
        if layout[0] != 3:
            raise ValueError("data layout message version {} is not supported".format(layout[0]))
        cls = layout[1]
        if cls == 0:  # compact
            size = struct.unpack_from("<H", layout, 2)[0]
            raw = layout[4:4 + size]
        elif cls == 1:  # contiguous
            addr, size = struct.unpack_from("<QQ", layout, 2)
            raw = b"" if addr == UNDEF else self.d[addr + self.base: addr + self.base + size]
        elif cls == 2:
            raw = self._chunked(layout, kinds.get(0x000B), dtype, shape)
        else:
            raise ValueError("unknown data layout class {}".format(cls))
        if len(raw) < count * dtype.itemsize:  # never allocated: fill value 0
            raw = raw + b"\x00" * (count * dtype.itemsize - len(raw))
        a = np.frombuffer(raw, dtype=dtype, count=count).reshape(shape)
        return a.astype(dtype.newbyteorder("="))

    def _chunked(self, layout: bytes, pipeline, dtype: np.dtype, shape) -> bytes:
        rank = layout[2] - 1
        btree = struct.unpack_from("<Q", layout, 3)[0]
        chunk = struct.unpack_from("<%dI" % rank, layout, 11)
        filters = []
        if pipeline is not None:
            version, nfilters = pipeline[0], pipeline[1]
            at = 8 if version == 1 else 2
            for _ in range(nfilters):
                if version == 1:
                    fid, name_len, _flags, nvalues = struct.unpack_from("<HHHH", pipeline, at)
                    at += 8 + ((name_len + 7) // 8) * 8
                else:
                    fid, _flags, nvalues = struct.unpack_from("<HHH", pipeline, at)
                    at += 6
                values = struct.unpack_from("<%dI" % nvalues, pipeline, at)
                at += 4 * nvalues + (4 if version == 1 and nvalues % 2 else 0)
                filters.append((fid, values))
        out = np.zeros(shape, dtype=dtype)
        if btree == UNDEF:
            return out.tobytes()
        d = self.d

        def walk(node):
            node += self.base
            if d[node:node + 4] != b"TREE":
                raise ValueError("chunk B-tree signature missing")
            kind, level, used = struct.unpack_from("<BBH", d, node + 4)
            if kind != 1:
                raise ValueError("expected a chunk B-tree")
            key_size = 8 + 8 * (rank + 1)
            for i in range(used):
                key = node + 24 + i * (key_size + 8)
                size, mask = struct.unpack_from("<II", d, key)
                offset = struct.unpack_from("<%dQ" % rank, d, key + 8)
                child = struct.unpack_from("<Q", d, key + key_size)[0]
                if level > 0:
                    yield from walk(child)
                else:
                    yield size, mask, offset, child

        for size, mask, offset, addr in walk(btree):
            raw = d[addr + self.base: addr + self.base + size]
            for index, (fid, values) in reversed(list(enumerate(filters))):
                if mask & (1 << index):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    width = values[0] if values else dtype.itemsize
                    n = len(raw) // width
                    raw = np.frombuffer(raw, np.uint8)[: n * width].reshape(width, n).T.tobytes()
                elif fid == 3:  # fletcher32: the checksum trails the data
                    raw = raw[:-4]
                else:
                    raise ValueError("HDF5 filter {} is not supported".format(fid))
            block = np.frombuffer(raw, dtype=dtype, count=int(np.prod(chunk))).reshape(chunk)
            where = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offset, chunk, shape))
            out[where] = block[tuple(slice(0, w.stop - w.start) for w in where)]
        return out.tobytes()


def read(filename: str) -> Tree:
    """The whole file as nested dicts of numpy arrays (rank-0 arrays for scalars)."""
    with open(filename, "rb") as f:
        data = f.read()
    reader = _Reader(data)
    return reader.read_object(reader.root_header)


def lookup(tree: Tree, path: str):
    node = tree
    for part in path.strip("/").split("/"):
        if not isinstance(node, dict) or part not in node:
            raise KeyError("no object '{}' in the file".format(path))
        node = node[part]
    return node
