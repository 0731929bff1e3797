#!/usr/bin/env python3
"""Run ONCE in any TensorFlow / Keras environment (the CartPoleSimulation environment has one; the GPU image does not):
pins the Keras -> kernel GRU weight conversion (cartpolesimulation_amd/model_folder.keras_gru_weights_to_model: gate
blocks z, r, h -> r, z, n, transposes, reset_after bias pairs) to KERAS ITSELF instead of to a restatement of its
equations.

  python tools/export_keras_gru.py [MODEL.keras] [--out tests/golden/keras_gru]

Without MODEL it builds GRU(32) -> GRU(32) -> Dense(5) on 6 inputs (the layer stack of GRU-6IN-32H1-32H2-5OUT,
SI_Toolkit_ASF/config_predictors.yml:8-13) with Glorot weights from a fixed seed; with MODEL it loads that file (e.g.
.../Models/GRU-6IN-32H1-32H2-5OUT-0/GRU-6IN-32H1-32H2-5OUT-0.keras).  Writes into --out:
  weights_keras.npz   np.savez(*model.get_weights())           what INTEGRATION.md section 3 asks a user to export
  keras_io.npz        x [B, T, 6] float32 (fixed seed), y [B, T, 5] = model(x) as Keras computes it (float32),
                      keras_version, tensorflow_version, model_source
tests/test_model_folder.py::test_conversion_is_pinned_to_keras consumes the two files when they are present (commit
them: together < 100 KB) and is skipped otherwise."""
import argparse
import os

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default=None)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "keras_gru"))
    args = ap.parse_args()
    import tensorflow as tf
    keras = tf.keras
    if args.model:
        model = keras.models.load_model(args.model, compile=False)
        source = os.path.basename(args.model)
    else:
        keras.utils.set_random_seed(20260102)
        model = keras.Sequential([keras.layers.Input(shape=(None, 6)),
                                  keras.layers.GRU(32, return_sequences=True),
                                  keras.layers.GRU(32, return_sequences=True),
                                  keras.layers.Dense(5)])
        # non-zero biases: get_weights() of a fresh model has zero biases, which would not exercise the bias layout
        rng = np.random.default_rng(7)
        ws = [w if w.ndim > 1 and w.shape != (2, 96) else (0.3 * rng.standard_normal(w.shape)).astype(np.float32)
              for w in model.get_weights()]
        model.set_weights(ws)
        source = "Sequential(GRU32, GRU32, Dense5), seed 20260102, random biases"
    ws = model.get_weights()
    shapes = [w.shape for w in ws]
    assert shapes == [(6, 96), (32, 96), (2, 96), (32, 96), (32, 96), (2, 96), (32, 5), (5,)], shapes
    x = np.random.default_rng(11).standard_normal((5, 12, 6)).astype(np.float32)
    y = np.asarray(model(x, training=False), dtype=np.float32)
    os.makedirs(args.out, exist_ok=True)
    np.savez(os.path.join(args.out, "weights_keras.npz"), *ws)
    np.savez(os.path.join(args.out, "keras_io.npz"), x=x, y=y, keras_version=str(getattr(keras, "__version__", "")),
             tensorflow_version=str(tf.__version__), model_source=source)
    print("wrote", args.out, "y[0,0] =", y[0, 0])


if __name__ == "__main__":
    main()
