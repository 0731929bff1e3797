#!/usr/bin/env python3
"""Golden vector for the .keras reader (test infrastructure; run in THIS container, needs /root/reference).

The reference's in-tree model folder GymlikeCartPole/Dense-7IN-32H1-32H2-1OUT-0/ holds the same trained network twice:
as <name>.keras (zip of config.json + model.weights.h5) and as C arrays (C_implementation/network_parameters.c, written by
the reference's own export script: weightsK[] = layer K's kernel transposed, row-major [units][inputs]; biasK[]).  This
script parses the C arrays - decimal literals that round-trip float32 - into tests/golden/keras_dense_c_export.npz; the
test reads the .keras archive with cartpolesimulation_amd/hdf5_min.py and must reproduce them bit for bit.
"""
import os
import re
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src = open(os.path.join(REF, "GymlikeCartPole", "Dense-7IN-32H1-32H2-1OUT-0", "C_implementation", "network_parameters.c")).read()
arrays = {}
for m in re.finditer(r"const float (\w+)\[\] = \{([^}]*)\}", src):
    arrays[m.group(1)] = np.array([np.float32(x) for x in m.group(2).replace("\n", " ").split(",") if x.strip()], np.float32)
sizes = (7, 32, 32, 1)
out = {}
for k in range(3):
    out[f"kernel{k}"] = arrays[f"weights{k + 1}"].reshape(sizes[k + 1], sizes[k]).T.copy()      # Keras layout [inputs, units]
    out[f"bias{k}"] = arrays[f"bias{k + 1}"]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "keras_dense_c_export.npz")
np.savez(dst, **out)
print(dst, {k: v.shape for k, v in out.items()})
