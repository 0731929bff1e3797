"""CPU tests of bench.py's N > 1 entry point: `python bench.py --gpus N` must itself start N ranks (the driver runs it
exactly like that), rank 0's line must say n_gpus == N, and a rank count that disagrees with --gpus must fail loudly.
`--dry-run` exercises the launcher, the rendezvous and the all-gather of the control sequences on CPU (gloo); no kernel
runs, so `value` is null."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def _line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}:\n{stdout}"
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2])
def test_gpus_flag_starts_that_many_ranks(n):
    r = _run(["--gpus", str(n), "--dry-run", "--steps", "4", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _line(r.stdout)
    assert out["n_gpus"] == n and out["steps"] == 4 and out["warmup"] == 1
    assert out["value"] is None and out["dry_run"] is True
    # the line says what the collective spanned (a real run: what RCCL itself reports through ncclCommCount)
    coll = out["config"]["collective"]
    if n == 1:
        assert coll is None                              # a single process: no rendezvous, no collective
    else:
        assert coll["ranks_requested"] == n and coll["backend_ranks"] == n and coll["rank_blocks_distinct"] is True
        assert coll["rccl_ranks"] is None and "gloo" in coll["impl"]
        # the stamped blocks of the per-step gather (the grouped C4 form runs under the collective with --gpus N): the one stale
        # block - the last rank re-sending gather 1 as gather 2 - is rejected by the receiver's rule, everything else accepted
        st = coll["stamped_blocks"]
        assert st["gathers"] == 3 and st["rejected"] == [[2, n - 1]] and st["final_rows_are_gather_3"] is True
        assert "cpmppi_groups_run_gather" in coll["pipelined_under_collective"]


def test_rank_count_must_match_gpus_flag():
    # started by hand with 2 ranks' environment but --gpus 4: must refuse instead of printing a mislabelled number
    env = {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"}
    r = _run(["--gpus", "4", "--dry-run"], env=env, timeout=60)
    assert r.returncode != 0
    assert "--gpus 4" in (r.stderr + r.stdout)


def test_launcher_does_not_touch_torch_before_spawning():
    """The parent of an N > 1 run must not import torch (nor anything that initialises the GPU) before it starts the
    ranks: a process that has touched the GPU must never replace or fork the ranks (task statement)."""
    code = ("import sys, bench\n"
            "bench.subprocess.run = lambda *a, **k: type('R', (), {'returncode': 0})()\n"
            "sys.argv = ['bench.py', '--gpus', '2']\n"
            "try:\n    bench.main()\nexcept SystemExit as e:\n    assert e.code == 0\n"
            "assert 'torch' not in sys.modules, 'torch imported before the ranks were started'\n"
            "print('ok')\n")
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
