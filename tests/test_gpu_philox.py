"""GPU: the device sampler IS Philox4x32-10 with the documented keying - cpmppi_sample's knots against oracle/philox_np.py (numpy
restatement, pinned to Random123's known-answer vectors by tests/test_oracle_philox.py).  The uniforms behind a normal are 24-bit
integers: a wrong word moves the normal by O(1), so agreement to the hardware's ln / sin / cos error pins the words themselves."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import philox_np as PH  # noqa: E402

f32 = np.float32
# v_log_f32 (1 ulp) -> r = sqrt(-2 ln u) <= 5.8, v_sin_f32 / v_cos_f32 on revolutions: absolute error of a standard normal as the
# device forms it against the float64 evaluation, measured 1.9e-6 at worst over 10^7 draws (profiles/r5/philox_error.txt)
Z_ATOL = 4e-6


@pytest.mark.parametrize("seed,offset,env_offset,E,N,H", [
    (1234, 0, 0, 3, 1024, 50), (1234, 7, 4096, 2, 2048, 50), (99, 450, 17, 1, 4096, 100),
    (2 ** 40 + 5, 3, 0, 2, 256, 20),                 # a seed with a high word
    (7, 2 ** 33 + 11, 123456, 2, 512, 35),           # a step counter beyond 32 bits, large global env index
])
def test_cpmppi_sample_is_philox4x32_10(seed, offset, env_offset, E, N, H):
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H))
    kn, _ = eng.sample(seed, offset=offset, env_offset=env_offset)
    kn = kn.cpu().numpy()
    ref = PH.knots(seed, offset, env_offset, E, N, eng.P, eng.mppi.sigma)
    sigma = float(f32(eng.mppi.sigma))
    err = np.abs(kn.astype(np.float64) - ref) / sigma
    assert err.max() <= Z_ATOL, (err.max(), np.unravel_index(err.argmax(), err.shape))
    # and the stream the rollout kernel draws in-kernel is the sampler's (same device function; the FAST kernel interpolates its
    # own knots with a float32 slope, fed knots with scipy's float64 one: costs agree to rounding, not bit for bit)
    s0 = np.tile(np.array([[0.2, 0.0, np.cos(0.2), np.sin(0.2), 0.0, 0.0]], f32), (E, 1))
    ua, ub = eng.zeros(E, H), eng.zeros(E, H)
    Sa, Sb = eng.empty(E, N), eng.empty(E, N)
    eng.step(s0, ua, 0.0, 1.0, seed=seed, offset=offset, env_offset=env_offset, S_out=Sa)
    eng.step(s0, ub, 0.0, 1.0, knots=torch.as_tensor(kn, device=ua.device), S_out=Sb)
    np.testing.assert_allclose(Sa.cpu().numpy(), Sb.cpu().numpy(), rtol=2e-5)
    np.testing.assert_allclose(ua.cpu().numpy(), ub.cpu().numpy(), atol=2e-5)
    eng.close()


def test_philox_error_statistics(capsys):
    """The evidence behind Z_ATOL: distribution of |z_device - z_float64| over ~10^7 normals."""
    from cartpolesimulation_amd.engine import MPPIEngine
    from cartpolesimulation_amd.configs import MPPIConfig
    E, N, H = 64, 4096, 100                          # 11 knots
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H))
    worst, n = 0.0, 0
    hist = np.zeros(8, np.int64)
    for off in range(4):
        kn = eng.sample(5, offset=off)[0].cpu().numpy().astype(np.float64) / float(f32(eng.mppi.sigma))
        z = PH.knots(5, off, 0, E, N, eng.P, 1.0).astype(np.float64)
        # (sigma * z is one more float32 rounding on the device: compare in units of sigma, its rounding included in the bound)
        d = np.abs(kn - z)
        worst, n = max(worst, float(d.max())), n + d.size
        hist += np.histogram(d, bins=[0, 1e-7, 2.5e-7, 5e-7, 1e-6, 2e-6, 4e-6, 1e-5, 1])[0]
    with capsys.disabled():
        print(f"\n[philox] {n} normals: worst |z_dev - z_f64| = {worst:.3e}; counts by error bin "
              f"[0,1e-7,2.5e-7,5e-7,1e-6,2e-6,4e-6,1e-5,1): {hist.tolist()}")
    assert worst <= Z_ATOL
    eng.close()
