"""GPU test of the bench line itself: `python bench.py` (small shapes, seconds) prints ONE JSON object with the keys the
driver and the judge read — metric / value / unit / n_gpus / steps / warmup / ms_per_step / dtype / data / config.workload,
the `roofline` object (bound, achieved, peak, frac, traffic + where the traffic figure comes from), `roofline_valu`, and
`cpu_baseline` (value, cores, kind, sample) — and its numbers are consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract_small_shape():
    E, N, H, K, W = 64, 256, 20, 5, 2
    out = _bench("--envs", str(E), "--rollouts", str(N), "--horizon", str(H), "--steps", str(K), "--warmup", str(W),
                 "--no-extra-configs")
    assert out["unit"] == "rollouts/s" and out["n_gpus"] == 1 and out["steps"] == K and out["warmup"] == W
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["data"] == "synthetic" and out["dtype"] == "f32"
    assert out["vs_baseline"] is None and "workload" in out["config"] and "model" not in out["config"]
    # value = units processed / wall time of the K timed steps
    assert out["value"] == pytest.approx(E * N / (out["ms_per_step"] * 1e-3), rel=1e-6)
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-9)
    assert 0.0 < rf["kernel_ms"] <= out["ms_per_step"] * 1.05          # HIP events on the launch stream, inside the wall time
    assert rf["achieved"] == pytest.approx(rf["algorithmic_bytes_per_rollout"] * E * N / (rf["kernel_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert rf["traffic"] is None or "traffic_source" in rf               # a profiled figure names its source, never posed as live
    rv = out["roofline_valu"]
    assert rv["bound"] == "fp32-valu" and rv["frac"] == pytest.approx(rv["achieved"] / rv["peak"], rel=1e-9) and 0 < rv["frac"] < 1
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "rollouts/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert out["value"] > cb["value"]
    se = out["single_env"]
    assert se["rollouts_per_s"] == pytest.approx(N / (se["us_per_step"] * 1e-6), rel=1e-6)
    # the timed configuration was checked against the oracle after timing: the same kernel instantiation, envs of the launch
    # re-computed by the C oracle, every clear rollout inside the ODE_v0 rule
    for v in (out["verified"], se["verified"]):
        assert v["ok"] is True and v["same_kernel_as_timed"] is True and v["clear_off"] == 0 and v["u_off_envs"] == 0
        assert v["rule"] == "predictor_ODE_v0" and v["kernel"].startswith("rollout_cost_kernel<") and v["envs"] >= 1
    assert out["verified"]["envs"] == 8 and out["verified"]["rollouts"] == 8 * N


def test_bench_exits_nonzero_when_the_oracle_disagrees(monkeypatch):
    """A timed configuration whose results the oracle does not confirm must not pass as a bench line: with the oracle's pole
    mass changed behind the bench's back (CPMPPI_BENCH_TEST_PERTURB_ORACLE, test hook of bench.py's verify leg) the run prints
    its line, reports the miss on stderr and exits with code 3."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["CPMPPI_BENCH_TEST_PERTURB_ORACLE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--envs", "16", "--rollouts", "256", "--horizon", "20", "--steps", "3",
                        "--warmup", "1", "--no-extra-configs", "--no-cpu-baseline", "--no-single-env"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "verification FAILED" in r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["verified"]["ok"] is False and line["verified"]["clear_off"] > 0


def test_bench_gru_line_names_the_mfma_roof():
    out = _bench("--predictor", "gru", "--envs", "8", "--rollouts", "256", "--horizon", "10", "--steps", "3", "--warmup", "1",
                 "--no-cpu-baseline", "--no-single-env")
    assert out["config"]["predictor"].lower().startswith("gru") or "GRU" in json.dumps(out["config"])
    assert out["roofline"]["bound"] == "mfma" and out["roofline"]["unit"] == "TFLOP/s" and 0 < out["roofline"]["frac"] < 1
