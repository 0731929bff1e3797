"""Test helper: writes a TensorFlow checkpoint ("tensor bundle": <prefix>.index as a LevelDB table + one data shard) from
{tensor name: ndarray}, following the public format descriptions (LevelDB table_format.md, tensor_bundle.proto), so that
cartpolesimulation_amd/tf_bundle_min.py can be exercised on files other than the one checkpoint in the reference tree.
Test infrastructure only (block CRCs are written as zero: the reader does not verify them)."""
import numpy as np

_DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}


def _varint(x):
    out = bytearray()
    while True:
        c = x & 0x7F
        x >>= 7
        out.append(c | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _field(num, wt, payload):
    return _varint((num << 3) | wt) + payload


def _entry_proto(arr, offset):
    dims = b"".join(_field(2, 2, _varint(len(d)) + d) for d in (_field(1, 0, _varint(s)) for s in arr.shape))
    msg = _field(1, 0, _varint(_DT[arr.dtype]))
    msg += _field(2, 2, _varint(len(dims)) + dims)
    msg += _field(4, 0, _varint(offset)) + _field(5, 0, _varint(arr.nbytes)) + _field(6, 5, b"\x00\x00\x00\x00")
    return msg


def _block(entries, restart_every=4):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        if i % restart_every == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts or [0]:
        out += int(r).to_bytes(4, "little")
    out += len(restarts or [0]).to_bytes(4, "little")
    return bytes(out)


def write_tf_checkpoint(prefix, tensors, per_block=5):
    names = sorted(tensors)
    data = bytearray()
    entries = [(b"", _field(1, 0, _varint(1)) + _field(3, 2, _varint(2) + _field(1, 0, _varint(1))))]     # header: 1 shard, version {producer 1}
    for n in names:
        arr = np.ascontiguousarray(tensors[n]) if np.ndim(tensors[n]) else np.asarray(tensors[n])
        entries.append((n.encode(), _entry_proto(arr, len(data))))
        data += arr.astype(arr.dtype.newbyteorder("<")).tobytes()
    table, index = bytearray(), []
    for i in range(0, len(entries), per_block):
        chunk = entries[i:i + per_block]
        blk = _block(chunk)
        index.append((chunk[-1][0] + b"\x00", _varint(len(table)) + _varint(len(blk))))
        table += blk + b"\x00" + b"\x00\x00\x00\x00"                  # trailer: no compression, CRC (unchecked)
    meta = _block([])
    meta_handle = _varint(len(table)) + _varint(len(meta))
    table += meta + b"\x00" + b"\x00\x00\x00\x00"
    iblk = _block(index, restart_every=1)
    index_handle = _varint(len(table)) + _varint(len(iblk))
    table += iblk + b"\x00" + b"\x00\x00\x00\x00"
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + (0xDB4775248B80FB57).to_bytes(8, "little")
    open(prefix + ".index", "wb").write(bytes(table) + footer)
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
