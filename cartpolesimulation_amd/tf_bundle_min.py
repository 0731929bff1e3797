"""A minimal reader of TensorFlow checkpoints (the "tensor bundle": ``<prefix>.index`` + ``<prefix>.data-00000-of-0000N``)
— enough to take the weights out of the ``ckpt.ckpt.*`` files an SI_Toolkit model folder carries (the reference's training
script saves them next to the ``.keras`` archive: ``GymlikeCartPole/Dense-7IN-32H1-32H2-1OUT-0/Training.py:69,163``) without
TensorFlow.

Container, restated from the public format descriptions (LevelDB "table_format.md"; TensorFlow
``tensor_bundle.proto``):

* ``.index`` is a LevelDB table: data blocks of prefix-compressed (shared, unshared, value length, key delta, value)
  entries followed by a restart array; every block is followed by a 5-byte trailer (compression type + CRC); the file
  ends in a 48-byte footer = metaindex handle, index handle (varint offset, varint size), padding, 8-byte magic.  The
  index block maps separator keys to the data blocks' handles.
* key ``""`` holds a ``BundleHeaderProto`` (fields num_shards = 1, endianness = 2, version = 3); every other key is a tensor name
  whose value is a ``BundleEntryProto``: dtype = 1, shape = 2 (``TensorShapeProto``: repeated dim = 2 {size = 1}),
  shard_id = 3, offset = 4, size = 5, crc32c = 6 (fixed32), slices = 7.
* the tensor's bytes sit at ``offset`` in shard ``shard_id``'s data file, little-endian, C order.

Only uncompressed blocks and whole (unsliced) numeric tensors are read; anything else raises ``NotImplementedError``.
"""
import os

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}


def _varint(b, p):
    x = shift = 0
    while True:
        c = b[p]
        p += 1
        x |= (c & 0x7F) << shift
        if not c & 0x80:
            return x, p
        shift += 7


def _block(b, offset, size):
    """The (key, value) pairs of one table block."""
    if b[offset + size] != 0:
        raise NotImplementedError(f"compressed table block (type {b[offset + size]})")
    blk = b[offset:offset + size]
    n_restarts = int.from_bytes(blk[-4:], "little")
    end = size - 4 - 4 * n_restarts
    p, key, out = 0, b"", []
    while p < end:
        shared, p = _varint(blk, p)
        unshared, p = _varint(blk, p)
        vlen, p = _varint(blk, p)
        key = key[:shared] + bytes(blk[p:p + unshared])
        p += unshared
        out.append((key, bytes(blk[p:p + vlen])))
        p += vlen
    return out


def _proto(b):
    """A flat protobuf message -> [(field number, wire type, value)] (varints as ints, length-delimited as bytes)."""
    p, out = 0, []
    while p < len(b):
        tag, p = _varint(b, p)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, p = _varint(b, p)
        elif wt == 1:
            v, p = int.from_bytes(b[p:p + 8], "little"), p + 8
        elif wt == 2:
            n, p = _varint(b, p)
            v, p = bytes(b[p:p + n]), p + n
        elif wt == 5:
            v, p = int.from_bytes(b[p:p + 4], "little"), p + 4
        else:
            raise ValueError(f"protobuf wire type {wt}")
        out.append((field, wt, v))
    return out


def read_tf_checkpoint(prefix):
    """{tensor name: ndarray} of the checkpoint ``<prefix>.index`` / ``<prefix>.data-*`` (string tensors — the object graph —
    are skipped)."""
    idx = open(prefix + ".index", "rb").read()
    if len(idx) < 48 or int.from_bytes(idx[-8:], "little") != _MAGIC:
        raise ValueError(f"{prefix}.index: not a LevelDB table (magic missing)")
    p = len(idx) - 48
    _, p = _varint(idx, p)                       # metaindex handle
    _, p = _varint(idx, p)
    ioff, p = _varint(idx, p)                    # index handle
    isize, p = _varint(idx, p)
    entries = []
    for _, handle in _block(idx, ioff, isize):
        off, q = _varint(handle, 0)
        size, q = _varint(handle, q)
        entries.extend(_block(idx, off, size))
    shards, out = {}, {}
    num_shards = 1
    for key, value in entries:
        if key == b"":
            for f, _, v in _proto(value):                       # BundleHeaderProto: num_shards = 1, endianness = 2, version = 3
                if f == 1:
                    num_shards = v
                if f == 2 and v == 1:                           # (LITTLE = 0 is the default and is not written)
                    raise NotImplementedError("big-endian tensor bundle")
            continue
        dtype = shard = offset = size = 0
        shape, sliced = [], False
        for f, wt, v in _proto(value):
            if f == 1:
                dtype = v
            elif f == 2:
                for f2, _, v2 in _proto(v):
                    if f2 == 2:
                        dim = [x for g, _, x in _proto(v2) if g == 1]
                        shape.append(dim[0] if dim else 0)
                    elif f2 == 3 and v2:
                        raise NotImplementedError("tensor of unknown rank")
            elif f == 3:
                shard = v
            elif f == 4:
                offset = v
            elif f == 5:
                size = v
            elif f == 7:
                sliced = True
        if dtype == 7 or dtype not in _DTYPES:                  # DT_STRING (the object graph) and exotic types: not weights
            continue
        if sliced:
            raise NotImplementedError(f"{key.decode()}: partitioned (sliced) variable")
        if shard not in shards:
            path = f"{prefix}.data-{shard:05d}-of-{num_shards:05d}"
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            shards[shard] = open(path, "rb").read()
        dt = np.dtype(_DTYPES[dtype]).newbyteorder("<")
        n = int(np.prod(shape)) if shape else 1
        if size != n * dt.itemsize:
            raise ValueError(f"{key.decode()}: {size} bytes for shape {shape} of {dt}")
        out[key.decode()] = np.frombuffer(shards[shard], dtype=dt, count=n, offset=offset).reshape(shape).astype(dt.newbyteorder("="))
    return out


def read_keras_checkpoint_weights(prefix):
    """``model.get_weights()`` from a checkpoint written by ``keras.Model.save_weights(prefix)`` (object-based keys
    ``layer_with_weights-<i>/[cell/]<kernel|recurrent_kernel|bias|…>/.ATTRIBUTES/VARIABLE_VALUE``): layers in index order,
    a layer's variables in Keras' creation order (kernel, recurrent_kernel, bias).  Optimizer slots are ignored."""
    import re
    tensors = read_tf_checkpoint(prefix)
    order = {"kernel": 0, "recurrent_kernel": 1, "bias": 2, "gamma": 3, "beta": 4, "moving_mean": 5, "moving_variance": 6}
    found = []
    for name, arr in tensors.items():
        m = re.fullmatch(r"layer_with_weights-(\d+)/(?:cell/)?(\w+)/\.ATTRIBUTES/VARIABLE_VALUE", name)
        if m:
            if m.group(2) not in order:
                raise NotImplementedError(f"{name}: variable kind {m.group(2)!r}")
            found.append((int(m.group(1)), order[m.group(2)], arr))
    if not found:
        raise ValueError(f"{prefix}: no layer_with_weights-* variables (not a Keras save_weights checkpoint?)")
    found.sort(key=lambda t: (t[0], t[1]))
    return [a for _, _, a in found]
