"""Model-folder reader (CPU): the reference's in-tree folder as format fixture, and a synthetic GRU folder round trip."""
import os

import numpy as np
import pytest

from cartpolesimulation_amd import model_folder as MF

HERE = os.path.dirname(os.path.abspath(__file__))
DENSE = os.path.join(HERE, "golden", "model_folder", "Dense-7IN-32H1-32H2-1OUT-0")


def test_reads_the_reference_net_info_and_normalization():
    info = MF.read_net_info(DENSE)
    assert info["inputs"] == ["angleD", "angle_cos", "angle_sin", "position", "positionD", "target_equilibrium", "target_position"]
    assert info["outputs"] == ["Q_calculated_offline"]
    assert info["net_name"] == "Dense-32H1-32H2" and info["net_full_name"] == "Dense-7IN-32H1-32H2-1OUT-0"
    assert info["type"] == "Dense" and info["library"] == "TF" and info["normalize"] == "True"
    nm = MF.read_normalization(DENSE, info)
    np.testing.assert_allclose(nm["a"], [0.05440483, 1, 1, 5.05050516, 0.88866770, 1, 5.05050516], rtol=1e-7)
    assert not nm["b"].any() and nm["A"].tolist() == [1.0] and nm["B"].tolist() == [0.0]
    with pytest.raises(NotImplementedError):
        MF.load_gru_model(DENSE)                       # a Dense controller net is not the GRU predictor


def _write_gru_folder(root, inputs, outputs, weights, a, b, A, B, name="GRU-6IN-32H1-32H2-5OUT-0"):
    d = os.path.join(root, name)
    os.makedirs(d)
    with open(os.path.join(d, name + ".txt"), "w") as f:
        f.write("CREATED:\n2024-01-01 at time 00:00:00\n\nLIBRARY:\nPytorch\n\nNET NAME:\nGRU-32H1-32H2\n\nNET FULL NAME:\n" + name +
                "\n\nINPUTS:\n" + ", ".join(inputs) + "\n\nOUTPUTS:\n" + ", ".join(outputs) + "\n\nTYPE:\nGRU\n\nNORMALIZE:\nTrue\n")
    for fn, v in (("normalization_vec_a.csv", a), ("normalization_vec_b.csv", b), ("denormalization_vec_A.csv", A),
                  ("denormalization_vec_B.csv", B)):
        np.savetxt(os.path.join(d, fn), np.asarray(v)[None], delimiter=",", fmt="%.8f")
    np.savez(os.path.join(d, "weights.npz"), **weights)
    return d


def test_gru_folder_round_trip_with_permuted_features(tmp_path):
    rng = np.random.Generator(np.random.SFC64(2))
    u = lambda *s: rng.uniform(-0.3, 0.3, s).astype(np.float32)  # noqa: E731
    kernel = dict(w_ih0=u(96, 6), w_hh0=u(96, 32), b_ih0=u(96), b_hh0=u(96), w_ih1=u(96, 32), w_hh1=u(96, 32), b_ih1=u(96),
                  b_hh1=u(96), w_out=u(5, 32), b_out=u(5))
    a, b = rng.uniform(0.5, 2, 6).astype(np.float32), u(6)
    A, B = rng.uniform(0.5, 2, 5).astype(np.float32), u(5)
    # the folder lists its features in another order than the kernel's
    pin, pout = [3, 0, 5, 1, 4, 2], [4, 2, 0, 1, 3]
    inputs, outputs = [MF.KERNEL_INPUTS[i] for i in pin], [MF.KERNEL_OUTPUTS[i] for i in pout]
    folder_w = dict(kernel)
    folder_w["w_ih0"] = kernel["w_ih0"][:, pin]
    folder_w["w_out"], folder_w["b_out"] = kernel["w_out"][pout], kernel["b_out"][pout]
    d = _write_gru_folder(str(tmp_path), inputs, outputs, folder_w, a[pin], b[pin], A[pout], B[pout])
    m = MF.load_gru_model(d)
    for k, v in kernel.items():
        np.testing.assert_array_equal(m[k], v, err_msg=k)
    np.testing.assert_allclose(m["in_scale"], a, rtol=1e-6)
    np.testing.assert_allclose(m["in_shift"], b, atol=1e-7)
    np.testing.assert_allclose(m["out_scale"], A, rtol=1e-6)
    np.testing.assert_allclose(m["out_shift"], B, atol=1e-7)


def test_torch_state_dict_and_tensorflow_only_folders(tmp_path):
    torch = pytest.importorskip("torch")
    gru = torch.nn.GRU(6, 32, num_layers=2, batch_first=True)
    head = torch.nn.Linear(32, 5)
    d = _write_gru_folder(str(tmp_path), MF.KERNEL_INPUTS, MF.KERNEL_OUTPUTS, {}, np.ones(6), np.zeros(6), np.ones(5), np.zeros(5))
    os.remove(os.path.join(d, "weights.npz"))
    sd = {("rnn." + k): v for k, v in gru.state_dict().items()}
    sd.update({("head." + k): v for k, v in head.state_dict().items()})
    torch.save(sd, os.path.join(d, "ckpt.pt"))
    m = MF.load_gru_model(d)
    np.testing.assert_array_equal(m["w_hh1"], gru.weight_hh_l1.detach().numpy())
    np.testing.assert_array_equal(m["w_out"], head.weight.detach().numpy())
    os.remove(os.path.join(d, "ckpt.pt"))
    open(os.path.join(d, "ckpt.ckpt.index"), "w").close()
    with pytest.raises(NotImplementedError, match="TensorFlow"):
        MF.load_gru_model(d)
