"""A minimal HDF5 reader — enough to open the ``model.weights.h5`` inside a Keras ``.keras`` archive without h5py.

Why: SI_Toolkit model folders (``SI_Toolkit_ASF/config_predictors.yml:8-13``; the in-tree example
``GymlikeCartPole/Dense-7IN-32H1-32H2-1OUT-0/``) carry their weights as ``<name>.keras`` = a zip around
``config.json`` + ``model.weights.h5``, and this image has neither TensorFlow nor h5py.  Keras writes that file with the
library defaults, i.e. the CLASSIC on-disk format, which is small enough to read directly:

* superblock version 0 or 1 (8-byte or 4-byte offsets / lengths),
* version-1 object headers (with continuation blocks),
* groups as symbol tables: a version-1 B-tree of ``SNOD`` symbol nodes + a local heap of names,
* datasets with a simple dataspace, a fixed-point or IEEE floating-point datatype (little- or big-endian) and a
  compact, contiguous or unfiltered chunked layout.

Anything else (new-style groups, filters / compression, variable-length or compound types) raises ``NotImplementedError``
naming what was met — nothing is guessed.  Format reference: the HDF5 File Format Specification, version 1.1 / 2.0
(public; sections "Disk Format: Level 0A / 1A / 1C / 1D / 2A"), restated here from the specification, not from any
reader's source.
"""
import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = {4: 0xFFFFFFFF, 8: 0xFFFFFFFFFFFFFFFF}


class _Reader:
    def __init__(self, data):
        self.b = memoryview(bytes(data))
        if bytes(self.b[:8]) != _SIG:
            raise ValueError("not an HDF5 file (signature missing at offset 0)")
        ver = self.b[8]
        if ver not in (0, 1):
            raise NotImplementedError(f"HDF5 superblock version {ver} (only the classic versions 0 / 1 are read)")
        self.O, self.L = self.b[13], self.b[14]                 # sizes of offsets and of lengths
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise NotImplementedError(f"offset / length sizes {self.O} / {self.L}")
        p = 24 + (4 if ver == 1 else 0)                         # v1 adds indexed-storage K + 2 reserved bytes
        self.base = self._off(p)
        p += 4 * self.O                                         # base, free-space info, end of file, driver info
        # root group symbol-table entry: link name offset, object header address, cache type, reserved, scratch pad
        self.root_header = self._off(p + self.O)

    # ---- primitives ------------------------------------------------------------------------------------------------
    def _u(self, p, n):
        return int.from_bytes(self.b[p:p + n], "little")

    def _off(self, p):
        return self._u(p, self.O)

    def _len(self, p):
        return self._u(p, self.L)

    # ---- object headers --------------------------------------------------------------------------------------------
    def messages(self, addr):
        """(type, flags, payload-offset, size) of every message of the version-1 object header at ``addr``."""
        a = addr + self.base
        if self.b[a] != 1:
            raise NotImplementedError(f"object header version {self.b[a]} at {addr:#x} (a new-style file: libver='latest')")
        count, size = self._u(a + 2, 2), self._u(a + 8, 4)
        blocks = [(a + 16, size)]                               # 12 bytes of prefix, padded to 16
        out = []
        while blocks and len(out) < count:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < count:
                mtype, msize, flags = self._u(p, 2), self._u(p + 2, 2), self.b[p + 4]
                body = p + 8
                if mtype == 0x0010:                             # continuation: offset, length
                    blocks.append((self._off(body) + self.base, self._len(body + self.O)))
                out.append((mtype, flags, body, msize))
                p = body + msize
        return out

    # ---- groups ----------------------------------------------------------------------------------------------------
    def _heap_name(self, heap_addr, offset):
        h = heap_addr + self.base
        if bytes(self.b[h:h + 4]) != b"HEAP":
            raise ValueError(f"local heap signature missing at {heap_addr:#x}")
        seg = self._off(h + 8 + 2 * self.L) + self.base
        q = seg + offset
        e = q
        while self.b[e] != 0:
            e += 1
        return bytes(self.b[q:e]).decode("utf-8")

    def _tree_entries(self, tree_addr, heap_addr, out):
        t = tree_addr + self.base
        sig = bytes(self.b[t:t + 4])
        if sig == b"TREE":
            ntype, level, used = self.b[t + 4], self.b[t + 5], self._u(t + 6, 2)
            if ntype != 0:
                raise ValueError("group B-tree expected (node type 0)")
            p = t + 8 + 2 * self.O                              # past the sibling addresses
            for i in range(used):
                child = self._off(p + self.L + i * (self.L + self.O))
                self._tree_entries(child, heap_addr, out)
        elif sig == b"SNOD":
            n = self._u(t + 6, 2)
            p = t + 8
            esz = 2 * self.O + 24                               # name offset, header address, cache type, reserved, scratch
            for i in range(n):
                q = p + i * esz
                out.append((self._heap_name(heap_addr, self._off(q)), self._off(q + self.O)))
        else:
            raise ValueError(f"unexpected node {sig!r} in a group B-tree at {tree_addr:#x}")

    def children(self, header_addr):
        """[(name, object header address)] of a group, or None if the object is not a group."""
        for mtype, _, body, _ in self.messages(header_addr):
            if mtype == 0x0011:                                 # symbol table: B-tree address, local heap address
                out = []
                self._tree_entries(self._off(body), self._off(body + self.O), out)
                return out
            if mtype in (0x0002, 0x0006):
                raise NotImplementedError("new-style group (link info / link messages): file written with libver='latest'")
        return None

    # ---- datasets --------------------------------------------------------------------------------------------------
    def _dtype(self, body):
        cls, ver = self.b[body] & 0x0F, self.b[body] >> 4
        bits0 = self.b[body + 1]
        size = self._u(body + 4, 4)
        order = ">" if (bits0 & 1) else "<"
        if cls == 0:                                            # fixed point; bit 3 of the class bits: signed
            return np.dtype(f"{order}{'i' if (bits0 & 8) else 'u'}{size}")
        if cls == 1:
            if size not in (2, 4, 8):
                raise NotImplementedError(f"{size}-byte floating point")
            return np.dtype(f"{order}f{size}")
        raise NotImplementedError(f"HDF5 datatype class {cls} (version {ver}): only fixed- and floating-point arrays are read")

    def _shape(self, body):
        ver, rank = self.b[body], self.b[body + 1]
        if ver == 1:
            p = body + 8
        elif ver == 2:
            if self.b[body + 3] == 2:
                raise NotImplementedError("null dataspace")
            p = body + 4
        else:
            raise NotImplementedError(f"dataspace version {ver}")
        return tuple(self._len(p + i * self.L) for i in range(rank))

    def _chunked(self, btree, chunk, shape, dtype):
        out = np.zeros(shape, dtype)

        def walk(addr):
            t = addr + self.base
            if bytes(self.b[t:t + 4]) != b"TREE" or self.b[t + 4] != 1:
                raise ValueError("chunk B-tree expected (node type 1)")
            level, used = self.b[t + 5], self._u(t + 6, 2)
            nd = len(chunk)                                     # dimensionality + 1 (the element-size dimension)
            ksz = 8 + 8 * nd
            p = t + 8 + 2 * self.O
            for i in range(used):
                k = p + i * (ksz + self.O)
                csize, mask = self._u(k, 4), self._u(k + 4, 4)
                offs = [self._u(k + 8 + 8 * d, 8) for d in range(nd - 1)]
                child = self._off(k + ksz)
                if level > 0:
                    walk(child)
                    continue
                if mask != 0:
                    raise NotImplementedError("filtered (compressed) chunks")
                c = child + self.base
                block = np.frombuffer(self.b[c:c + csize], dtype=dtype, count=int(np.prod(chunk[:-1]))).reshape(chunk[:-1])
                sl = tuple(slice(o, min(o + cs, s)) for o, cs, s in zip(offs, chunk[:-1], shape))
                out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]
        walk(btree)
        return out

    def dataset(self, header_addr):
        """The array of a dataset object, or None if the object has no dataspace + datatype + layout."""
        shape = dtype = layout = None
        for mtype, _, body, size in self.messages(header_addr):
            if mtype == 0x0001:
                shape = self._shape(body)
            elif mtype == 0x0003:
                dtype = self._dtype(body)
            elif mtype == 0x0008:
                layout = (body, size)
            elif mtype == 0x000B:
                raise NotImplementedError("filter pipeline (compressed dataset)")
        if shape is None or dtype is None or layout is None:
            return None
        body, _ = layout
        ver = self.b[body]
        n = int(np.prod(shape)) if shape else 1
        if ver == 3:
            cls = self.b[body + 1]
            if cls == 0:                                        # compact: size, data
                raw = self.b[body + 4:body + 4 + self._u(body + 2, 2)]
            elif cls == 1:                                      # contiguous: address, size
                addr = self._off(body + 2)
                if addr == _UNDEF[self.O]:
                    return np.zeros(shape, dtype.newbyteorder("="))          # never written: the fill value (zero)
                raw = self.b[addr + self.base:addr + self.base + self._len(body + 2 + self.O)]
            elif cls == 2:                                      # chunked: dimensionality, B-tree address, chunk dims
                nd = self.b[body + 2]
                btree = self._off(body + 3)
                chunk = [self._u(body + 3 + self.O + 4 * d, 4) for d in range(nd)]
                if btree == _UNDEF[self.O]:
                    return np.zeros(shape, dtype.newbyteorder("="))
                return self._chunked(btree, chunk, shape, dtype).astype(dtype.newbyteorder("="))
            else:
                raise NotImplementedError(f"data layout class {cls}")
        elif ver in (1, 2):
            nd, cls = self.b[body + 1], self.b[body + 2]
            p = body + 8
            if cls == 1:
                addr = self._off(p)
                raw = self.b[addr + self.base:addr + self.base + n * dtype.itemsize]
            elif cls == 0:
                p += 4 * nd
                raw = self.b[p + 4:p + 4 + self._u(p, 4)]
            else:
                raise NotImplementedError("chunked layout in a version-1/2 layout message")
        else:
            raise NotImplementedError(f"data layout message version {ver}")
        return np.frombuffer(raw, dtype=dtype, count=n).reshape(shape).astype(dtype.newbyteorder("="))


def read_hdf5(data):
    """All datasets of an HDF5 file given as bytes: {"group/sub/name": ndarray}, in the file's own (alphabetical) order."""
    r = _Reader(data)
    out = {}

    def walk(addr, prefix, seen):
        if addr in seen:
            return
        seen = seen | {addr}
        kids = r.children(addr)
        if kids is None:
            arr = r.dataset(addr)
            if arr is not None:
                out[prefix] = arr
            return
        for name, child in kids:
            walk(child, f"{prefix}/{name}" if prefix else name, seen)
    walk(r.root_header, "", frozenset())
    return out


def read_keras_weights(path):
    """``model.get_weights()`` of a Keras ``.keras`` archive (Keras 2.13+ / 3 format: zip of config.json + model.weights.h5):
    the variables of every layer in the model's layer order, each layer's in its own ``vars/0, 1, …`` order.  Returns
    (list of arrays, list of (layer name, class name)) for the layers that own variables."""
    import json
    import zipfile
    with zipfile.ZipFile(path) as z:
        config = json.loads(z.read("config.json"))
        arrays = read_hdf5(z.read("model.weights.h5"))

    def vars_under(prefix):
        found = {}
        for key, arr in arrays.items():
            if key.startswith(prefix + "/") and "/vars/" in key[len(prefix):]:
                tail = key[len(prefix) + 1:]
                found[tail] = arr
        # a layer's own variables first (vars/N), then those of its cells / sub-layers, numerically within each holder
        def order(t):
            holder, _, idx = t.rpartition("vars/")
            return (holder.count("/"), holder, int(idx))
        return [found[t] for t in sorted(found, key=order)]

    layers = config["config"]["layers"] if "layers" in config.get("config", {}) else []
    # The archive names a layer's group after its CLASS in snake case with a counter (dense, dense_2, dense_4 in the
    # reference's in-tree example - the counter's step depends on the Keras version), not after its user-given name.
    # What is stable: within one class the counters grow in layer order.  So: the groups of a class, sorted by counter, are
    # matched to the model's layers of that class, in order.
    import re
    top = "_layer_checkpoint_dependencies" if any(k.startswith("_layer_checkpoint_dependencies/") for k in arrays) else "layers"
    groups = {}
    for key in arrays:
        if key.startswith(top + "/"):
            g = key[len(top) + 1:].split("/", 1)[0]
            m = re.fullmatch(r"(.*?)(?:_(\d+))?", g)
            groups.setdefault(m.group(1), set()).add((int(m.group(2) or 0), g))
    out, owners, used = [], [], {}
    for layer in layers:
        cls = layer["class_name"]
        snake = re.sub(r"(?<=[a-z0-9])([A-Z])", r"_\1", re.sub(r"([A-Z]+)([A-Z][a-z])", r"\1_\2", cls)).lower()
        ordered = sorted(groups.get(snake, ()))
        built = bool(layer.get("build_config")) or cls not in ("InputLayer", "Activation", "Dropout", "Flatten")
        if not ordered or not built:
            continue
        i = used.get(snake, 0)
        if i >= len(ordered):
            continue                                            # a layer of this class without variables comes after all that have them
        vs = vars_under(f"{top}/{ordered[i][1]}")
        used[snake] = i + 1
        if vs:
            out.extend(vs)
            owners.append((layer.get("config", {}).get("name", ordered[i][1]), cls))
    return out, owners
