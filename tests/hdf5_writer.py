"""Test helper: writes a small HDF5 file in the classic layout (superblock 0, version-1 object headers, symbol-table
groups, contiguous datasets) from {"group/sub/name": ndarray} — what h5py produces with its defaults for a Keras
``model.weights.h5``.  Written from the HDF5 File Format Specification so that ``cartpolesimulation_amd/hdf5_min.py`` can be
exercised on files other than the one archive in the reference tree (which pins the reader against a file h5py wrote).
Test infrastructure only."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K = 16            # a symbol node holds up to 2 K entries: one node per group is enough for the tests


def _pad8(b):
    return b + b"\x00" * (-len(b) % 8)


def _message(mtype, body):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), 0) + body


def _object_header(messages):
    data = b"".join(messages)
    return struct.pack("<BxHII4x", 1, len(messages), 1, len(data)) + data


def _datatype(dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        exp_bits, man_bits = {4: (8, 23), 8: (11, 52), 2: (5, 10)}[dt.itemsize]
        bias = (1 << (exp_bits - 1)) - 1
        head = struct.pack("<BBBBI", 0x11, 0x20, dt.itemsize * 8 - 1, 0, dt.itemsize)       # class 1 v1; implied-1 mantissa; sign bit
        return head + struct.pack("<HHBBBBI", 0, dt.itemsize * 8, man_bits, exp_bits, 0, man_bits, bias)
    if dt.kind in "iu":
        head = struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize)
        return head + struct.pack("<HH", 0, dt.itemsize * 8)
    raise ValueError(dt)


class Writer:
    def __init__(self):
        self.buf = bytearray(96)                 # superblock (56 bytes + the 40-byte root symbol-table entry) written last

    def _append(self, b):
        while len(self.buf) % 8:
            self.buf += b"\x00"
        addr = len(self.buf)
        self.buf += b
        return addr

    def dataset(self, arr):
        arr = np.asarray(arr)
        if arr.ndim and not arr.flags.c_contiguous:
            arr = np.ascontiguousarray(arr)          # (ascontiguousarray would turn a scalar into shape (1,))
        le = arr.astype(arr.dtype.newbyteorder("<"))
        data_addr = self._append(le.tobytes()) if arr.size else UNDEF
        space = struct.pack("<BBB5x", 1, arr.ndim, 0) + b"".join(struct.pack("<Q", d) for d in arr.shape)
        layout = struct.pack("<BBQQ", 3, 1, data_addr, le.nbytes)
        return self._append(_object_header([_message(0x0001, space), _message(0x0003, _datatype(arr.dtype)),
                                            _message(0x0008, layout)]))

    def group(self, entries):
        """entries: {name: object header address} -> the group's object header address (names sorted, as B-tree keys need)."""
        names = sorted(entries)
        assert len(names) <= 2 * LEAF_K
        heap_data = bytearray(b"\x00" * 8)       # offset 0: the empty string (the first B-tree key)
        offsets = {}
        for n in names:
            offsets[n] = len(heap_data)
            heap_data += _pad8(n.encode() + b"\x00")
        seg_addr = self._append(bytes(heap_data))
        heap_addr = self._append(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), UNDEF, seg_addr))
        snod = b"SNOD" + struct.pack("<BxH", 1, len(names))
        for n in names:
            snod += struct.pack("<QQII16x", offsets[n], entries[n], 0, 0)
        snod += b"\x00" * (40 * (2 * LEAF_K - len(names)))
        snod_addr = self._append(snod)
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF)
        tree += struct.pack("<QQQ", 0, snod_addr, offsets[names[-1]] if names else 0)
        tree_addr = self._append(tree)
        return self._append(_object_header([_message(0x0011, struct.pack("<QQ", tree_addr, heap_addr))]))

    def finish(self, root_header):
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", LEAF_K, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQII16x", 0, root_header, 0, 0)
        self.buf[:len(sb)] = sb
        return bytes(self.buf)


def write_hdf5(arrays):
    """{"a/b/c": ndarray} -> bytes of an HDF5 file."""
    w = Writer()
    tree = {}
    for path, arr in arrays.items():
        node = tree
        parts = path.split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = np.asarray(arr)

    def emit(node):
        if isinstance(node, np.ndarray):
            return w.dataset(node)
        return w.group({name: emit(child) for name, child in node.items()})
    return w.finish(emit(tree))
