"""Golden vectors for predictor_output_augmentation (SI_Toolkit_ASF/ToolkitCustomization/predictors_customization.py:
72-139), all three legs, produced by the reference's own class under the import stand-ins.
TEST INFRASTRUCTURE; usage:  cd /root/reference && python -B /root/repo/oracle/gen_golden_augmentation.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
os.chdir(ref_shims.REFERENCE_ROOT)
from SI_Toolkit_ASF.ToolkitCustomization.predictors_customization import predictor_output_augmentation  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
rng = np.random.Generator(np.random.SFC64(31))
cases = {"sincos": ["angleD", "angle_cos", "angle_sin", "position", "positionD"],
         "angle_only": ["angle", "angleD", "position", "positionD"],
         "angle_and_cos": ["angle", "angleD", "angle_cos", "position", "positionD"],
         "complete": ["angle", "angleD", "angle_cos", "angle_sin", "position", "positionD"]}
out = {}
for name, outputs in cases.items():
    aug = predictor_output_augmentation(SimpleNamespace(outputs=outputs), lib=ref_shims.NumpyLibrary(), disable_individual_compilation=True)
    x = rng.uniform(-2.5, 2.5, (4, 3, len(outputs))).astype(np.float32)
    y = aug.augment(x)
    out[f"{name}/outputs"] = np.array(outputs)
    out[f"{name}/x"], out[f"{name}/y"] = x, np.asarray(y)
    out[f"{name}/indices"] = np.array(aug.get_indices_augmentation(), dtype=np.int64)
    out[f"{name}/features"] = np.array(aug.get_features_augmentation())
    print(name, aug.get_features_augmentation(), np.asarray(y).shape, np.asarray(y).dtype)
np.savez_compressed(os.path.join(OUT, "augmentation.npz"), **out)
