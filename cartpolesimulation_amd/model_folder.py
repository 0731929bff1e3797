"""Reading SI_Toolkit model folders: net-info file, normalisation vectors, weights.

Format evidence in the reference tree: ``GymlikeCartPole/Dense-7IN-32H1-32H2-1OUT-0/`` — ``<net full name>.txt`` with
``KEY:`` lines followed by their value (INPUTS / OUTPUTS as comma-separated feature names, NET NAME, TYPE, LIBRARY,
NORMALIZE ...), ``normalization_vec_a.csv`` / ``normalization_vec_b.csv`` (one row, one value per INPUT: normalised =
a*x + b) and ``denormalization_vec_A.csv`` / ``denormalization_vec_B.csv`` (one value per OUTPUT: y = A*y_norm + B).
Model names of the neural predictor: ``SI_Toolkit_ASF/config_predictors.yml:8-13`` (``GRU-6IN-32H1-32H2-5OUT-*``).

Weights: the reference's folders carry ``.keras`` archives (a zip around ``config.json`` + ``model.weights.h5``) and / or
TensorFlow checkpoints.  The ``.keras`` archive is read directly (``hdf5_min.py``: a minimal reader of the classic HDF5
layout Keras writes — no TensorFlow, no h5py; pinned on the reference's in-tree
``GymlikeCartPole/Dense-7IN-32H1-32H2-1OUT-0/*.keras`` against that folder's own C export of the same weights).  Also
read: ``weights_keras.npz`` = ``np.savez(path, *model.get_weights())`` (Keras layout, converted by
``keras_gru_weights_to_model``), a ``torch`` state_dict (``ckpt.pt``, the layout of ``torch.nn.GRU`` + a ``Linear``
head) or ``weights.npz`` with the keys of ``MPPIEngine.set_gru``; and the TensorFlow checkpoint ``ckpt.ckpt.index`` +
``ckpt.ckpt.data-*`` that ``keras.Model.save_weights`` writes next to the archive (``tf_bundle_min.py``: LevelDB table +
``BundleEntryProto``, pinned on the same in-tree folder the same way).  Anything else raises with the reason; there is no
silent fallback.
"""
import os
import re

import numpy as np

# the kernel's fixed feature order (cpmppi_gru.hpp): inputs and outputs of GRU-6IN-32H1-32H2-5OUT
KERNEL_INPUTS = ("Q", "angleD", "angle_cos", "angle_sin", "position", "positionD")
KERNEL_OUTPUTS = ("angleD", "angle_cos", "angle_sin", "position", "positionD")


def read_net_info(folder):
    """-> dict of the net-info file: keys lower-cased with spaces -> '_'; 'inputs' / 'outputs' are lists of names."""
    folder = os.path.abspath(folder)
    name = os.path.basename(folder.rstrip(os.sep))
    path = os.path.join(folder, name + ".txt")
    if not os.path.exists(path):
        cands = [f for f in os.listdir(folder) if f.endswith(".txt") and not f.startswith("terminal")]
        if len(cands) != 1:
            raise FileNotFoundError(f"{folder}: no net-info file {name}.txt")
        path = os.path.join(folder, cands[0])
    info, key, buf = {}, None, []

    def flush():
        if key is not None:
            info[key] = "\n".join(buf).strip()

    for line in open(path):
        m = re.match(r"^([A-Z][A-Z _\[\]a-z]*):\s*$", line.rstrip("\n"))
        if m and m.group(1).upper() == m.group(1).upper() and line[0].isupper() and line.strip().endswith(":"):
            flush()
            key, buf = re.sub(r"[^a-z0-9]+", "_", m.group(1).lower()).strip("_"), []
        else:
            buf.append(line.rstrip("\n"))
    flush()
    for k in ("inputs", "outputs"):
        info[k] = [x.strip() for x in info.get(k, "").split(",") if x.strip()]
    info["path"] = path
    return info


def _row(path):
    return np.atleast_1d(np.loadtxt(path, delimiter=",", dtype=np.float64)).astype(np.float32)


def read_normalization(folder, info=None):
    """-> dict(a, b, A, B): normalised input = a*x + b per INPUT, output = A*y + B per OUTPUT (identity if files absent)."""
    info = info or read_net_info(folder)
    out = {}
    for key, fname, n in (("a", "normalization_vec_a.csv", len(info["inputs"])), ("b", "normalization_vec_b.csv", len(info["inputs"])),
                          ("A", "denormalization_vec_A.csv", len(info["outputs"])),
                          ("B", "denormalization_vec_B.csv", len(info["outputs"]))):
        p = os.path.join(folder, fname)
        if os.path.exists(p):
            v = _row(p)
            if v.size != n:
                raise ValueError(f"{p}: {v.size} values for {n} features")
        else:
            v = np.ones(n, np.float32) if key in ("a", "A") else np.zeros(n, np.float32)
        out[key] = v
    return out


def keras_gru_weights_to_model(arrays, units=32):
    """``model.get_weights()`` of the Keras network GRU(units) -> GRU(units) -> Dense(5) — the layout SI_Toolkit's
    TensorFlow nets are saved in (``config_predictors.yml:8-13`` model folders hold ``.keras`` / ``ckpt`` files) —
    into the ``torch.nn.GRU``-layout dict of ``MPPIEngine.set_gru``.

    Keras GRU layer (``reset_after=True``, the TF2 default): kernel [in, 3u], recurrent_kernel [u, 3u], bias [2, 3u]
    (row 0 input bias, row 1 recurrent bias), gate blocks ordered **z, r, h**, applied as ``x @ kernel``.
    torch: weight_ih [3u, in], weight_hh [3u, u], bias_ih, bias_hh [3u], gate blocks ordered **r, z, n**, applied as
    ``W @ x``; with reset_after the candidate is tanh(W_in x + b_in + r * (W_hn h + b_hn)) in both.  So: transpose and swap
    the first two gate blocks.  ``reset_after=False`` (bias of shape [3u]: the reset gate multiplies h BEFORE the
    recurrent product) is a different cell and is refused."""
    arrays = [np.asarray(a, dtype=np.float32) for a in arrays]
    if len(arrays) != 8:
        raise ValueError(f"expected 8 arrays (2 x [kernel, recurrent_kernel, bias] + dense kernel, bias), got {len(arrays)}")
    u = int(units)

    def zrh_to_rzn(m):                       # m [..., 3u] with blocks z, r, h  ->  blocks r, z, n
        z, r, h = m[..., :u], m[..., u:2 * u], m[..., 2 * u:]
        return np.concatenate([r, z, h], axis=-1)

    model = {}
    for l in range(2):
        k, rk, b = arrays[3 * l:3 * l + 3]
        if b.ndim != 2 or b.shape != (2, 3 * u):
            raise NotImplementedError(f"GRU layer {l}: bias shape {b.shape}; only reset_after=True layers ([2, {3 * u}]) "
                                      "have the cell torch.nn.GRU and the HIP kernel implement")
        if k.shape[1] != 3 * u or rk.shape != (u, 3 * u):
            raise ValueError(f"GRU layer {l}: kernel {k.shape}, recurrent_kernel {rk.shape} for {u} units")
        model[f"w_ih{l}"] = np.ascontiguousarray(zrh_to_rzn(k).T)
        model[f"w_hh{l}"] = np.ascontiguousarray(zrh_to_rzn(rk).T)
        model[f"b_ih{l}"], model[f"b_hh{l}"] = zrh_to_rzn(b[0]), zrh_to_rzn(b[1])
    dk, db = arrays[6], arrays[7]
    if dk.shape[0] != u or db.shape != (dk.shape[1],):
        raise ValueError(f"dense head: kernel {dk.shape}, bias {db.shape}")
    model["w_out"], model["b_out"] = np.ascontiguousarray(dk.T), db
    return model


def _weights(folder):
    npz, pt = os.path.join(folder, "weights.npz"), os.path.join(folder, "ckpt.pt")
    knpz = os.path.join(folder, "weights_keras.npz")        # np.savez(path, *model.get_weights()) — INTEGRATION.md
    if os.path.exists(npz):
        return dict(np.load(npz))
    if os.path.exists(knpz):
        z = np.load(knpz)
        return keras_gru_weights_to_model([z[f"arr_{i}"] for i in range(len(z.files))])
    if os.path.exists(pt):
        import torch
        sd = torch.load(pt, map_location="cpu", weights_only=True)
        sd = {k: v.numpy() for k, v in sd.items()}
        pick = lambda *subs: next(v for k, v in sd.items() if all(s in k for s in subs))  # noqa: E731
        w = {}
        for l in range(2):
            w[f"w_ih{l}"], w[f"w_hh{l}"] = pick(f"weight_ih_l{l}"), pick(f"weight_hh_l{l}")
            w[f"b_ih{l}"], w[f"b_hh{l}"] = pick(f"bias_ih_l{l}"), pick(f"bias_hh_l{l}")
        heads = [(k, v) for k, v in sd.items() if v.ndim == 2 and v.shape[0] == 5]
        w["w_out"] = heads[-1][1]
        w["b_out"] = next(v for k, v in sd.items() if v.ndim == 1 and v.shape[0] == 5)
        return w
    keras = sorted(f for f in os.listdir(folder) if f.endswith(".keras"))
    if keras:
        from .hdf5_min import read_keras_weights
        arrays, owners = read_keras_weights(os.path.join(folder, keras[0]))
        classes = [c for _, c in owners]
        if classes != ["GRU", "GRU", "Dense"]:
            raise NotImplementedError(f"{folder}/{keras[0]}: layers with variables are {classes}; the HIP predictor implements "
                                      "GRU -> GRU -> Dense (GRU-6IN-32H1-32H2-5OUT)")
        return keras_gru_weights_to_model(arrays)
    if os.path.exists(os.path.join(folder, "ckpt.ckpt.index")):       # keras.Model.save_weights('ckpt.ckpt'): Training.py:163
        from .tf_bundle_min import read_keras_checkpoint_weights
        return keras_gru_weights_to_model(read_keras_checkpoint_weights(os.path.join(folder, "ckpt.ckpt")))
    tf_like = [f for f in os.listdir(folder) if f.endswith((".h5", ".index", ".pb")) or ".ckpt" in f]
    if tf_like:
        raise NotImplementedError(f"{folder}: TensorFlow / Keras files found ({', '.join(sorted(tf_like)[:3])}) but neither a "
                                  "<name>.keras archive nor a ckpt.ckpt.index + data pair, the two containers read here; "
                                  "export once in the reference's environment (INTEGRATION.md: "
                                  "np.savez('weights_keras.npz', *model.get_weights()))")
    raise FileNotFoundError(f"{folder}: no weights.npz or ckpt.pt")


def load_gru_model(folder):
    """A ``GRU-6IN-32H1-32H2-5OUT-*`` folder -> the dict ``MPPIEngine.set_gru`` takes (weights permuted to the kernel's
    feature order, normalisation as in_scale/in_shift/out_scale/out_shift)."""
    info = read_net_info(folder)
    net = info.get("net_name", "")
    if not net.startswith("GRU-32H1-32H2"):
        raise NotImplementedError(f"{folder}: net {net!r}; the HIP predictor is built for GRU-32H1-32H2 (6 inputs, 5 outputs)")
    if sorted(info["inputs"]) != sorted(KERNEL_INPUTS) or sorted(info["outputs"]) != sorted(KERNEL_OUTPUTS):
        raise NotImplementedError(f"{folder}: features {info['inputs']} -> {info['outputs']}; built for {KERNEL_INPUTS} -> {KERNEL_OUTPUTS}")
    pin = [info["inputs"].index(f) for f in KERNEL_INPUTS]          # kernel column i  <- model column pin[i]
    pout = [info["outputs"].index(f) for f in KERNEL_OUTPUTS]
    w = {k: np.asarray(v, dtype=np.float32) for k, v in _weights(folder).items()}
    model = dict(w)
    model["w_ih0"] = w["w_ih0"][:, pin]
    model["w_out"], model["b_out"] = w["w_out"][pout], w["b_out"][pout]
    if str(info.get("normalize", "True")).strip().lower() != "false":
        nm = read_normalization(folder, info)
        model.update(in_scale=nm["a"][pin], in_shift=nm["b"][pin], out_scale=nm["A"][pout], out_shift=nm["B"][pout])
    for k in ("in_scale", "in_shift", "out_scale", "out_shift"):
        if k in w:                                                  # weights.npz may carry them directly (kernel order)
            model[k] = w[k]
    return model
