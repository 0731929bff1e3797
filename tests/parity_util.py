"""Shared parity assertions of the GPU tests: a re-export of oracle/parity.py (the rules live with the oracle, so that
bench.py's `verified` leg applies exactly what the tests apply).  See that module's docstring for the complete list of
tolerance knobs and which tests use each; in short:

  * rule ODE_V0 (default; every test of the north-star path): 1e-4 band + the oracle's A / B gap (+ the envelope of the
    oracle's further realisations where a test passes probes); flagged bucket capped at 2 % of itself / 0.5 % of all,
    `strict` (cap 0) on every reference-generated fixture; sensitive scatter NOT widened;
  * rule PREDICTOR_ODE (tests/test_gpu_ode_predictor.py, tools/dev/shape_fuzz.py --predictor-type ODE only): the same with the
    scatter of oracle-marked sensitive rollouts doubled.
"""
from oracle.parity import *  # noqa: F401,F403
from oracle.parity import _check  # noqa: F401
