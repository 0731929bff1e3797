"""Golden vectors for the GRU predictor path (BASELINE configs[4]) produced with torch.nn.GRU on the CPU.

TEST INFRASTRUCTURE.  No GRU weights exist in the reference tree, so the model is synthetic: torch.manual_seed(5),
torch's default GRU / Linear initialisation, hidden 32 x 2 layers, 6 inputs, 5 outputs (GRU-6IN-32H1-32H2-5OUT), plus
non-trivial normalisation vectors.  Expected trajectories come from torch.nn.GRU + torch.nn.Linear run step by step.
Usage: python oracle/gen_golden_gru.py
"""
import os

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
torch.manual_seed(5)
gru = torch.nn.GRU(input_size=6, hidden_size=32, num_layers=2, batch_first=True)
lin = torch.nn.Linear(32, 5)
rng = np.random.Generator(np.random.SFC64(5))
model = {
    "w_ih0": gru.weight_ih_l0, "w_hh0": gru.weight_hh_l0, "b_ih0": gru.bias_ih_l0, "b_hh0": gru.bias_hh_l0,
    "w_ih1": gru.weight_ih_l1, "w_hh1": gru.weight_hh_l1, "b_ih1": gru.bias_ih_l1, "b_hh1": gru.bias_hh_l1,
    "w_out": lin.weight, "b_out": lin.bias,
}
model = {k: v.detach().numpy().astype(np.float32).copy() for k, v in model.items()}
# normalisation: roughly the magnitudes of the cartpole variables (Q, angleD, cos, sin, position, positionD)
model["in_scale"] = np.array([1.0, 0.1, 1.0, 1.0, 5.0, 2.0], dtype=np.float32)
model["in_shift"] = np.array([0.0, 0.02, 0.0, 0.0, 0.01, -0.03], dtype=np.float32)
model["out_scale"] = (1.0 / model["in_scale"][1:]).astype(np.float32)
model["out_shift"] = (-model["in_shift"][1:] / model["in_scale"][1:]).astype(np.float32)

B, H = 48, 20
s0 = np.zeros((B, 6), dtype=np.float32)
ang = rng.uniform(-np.pi, np.pi, B)
s0[:, 0], s0[:, 1] = ang, rng.uniform(-8, 8, B)
s0[:, 2], s0[:, 3] = np.cos(ang), np.sin(ang)
s0[:, 4], s0[:, 5] = rng.uniform(-0.19, 0.19, B), rng.uniform(-0.5, 0.5, B)
Q = rng.uniform(-1, 1, (B, H)).astype(np.float32)
h0 = (0.3 * rng.standard_normal((2, B, 32))).astype(np.float32)

with torch.no_grad():
    feat = torch.tensor(np.stack([s0[:, 1], s0[:, 2], s0[:, 3], s0[:, 4], s0[:, 5]], 1)) * torch.tensor(model["in_scale"][1:]) \
        + torch.tensor(model["in_shift"][1:])
    h = torch.tensor(h0)
    traj = np.zeros((B, H + 1, 6), dtype=np.float32)
    traj[:, 0] = s0
    for k in range(H):
        qn = torch.tensor(Q[:, k]) * float(model["in_scale"][0]) + float(model["in_shift"][0])
        x = torch.cat([qn[:, None], feat], 1)[:, None, :]
        out, h = gru(x, h)
        feat = lin(out[:, 0])
        y = (feat * torch.tensor(model["out_scale"]) + torch.tensor(model["out_shift"])).numpy()
        traj[:, k + 1, 1], traj[:, k + 1, 2], traj[:, k + 1, 3], traj[:, k + 1, 4], traj[:, k + 1, 5] = y.T
        traj[:, k + 1, 0] = np.arctan2(y[:, 2], y[:, 1])
np.savez_compressed(os.path.join(OUT, "gru_c5.npz"), s0=s0, Q=Q, h0=h0, traj=traj, h_final=h.numpy(), **model)
print("gru_c5.npz written; traj range", traj.min(), traj.max())
