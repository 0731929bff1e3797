#!/bin/bash
# Round 4: the randomised differential tests on the current library (fresh seeds where the tool takes one)
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" > $O/gpu_tests_final.txt
python tools/fuzz_parity.py --costs qbgm default legacy 2>/dev/null | tail -1 > $O/fuzz_parity.json
python tools/dev/shape_fuzz.py --n 400 --seed 41 2>/dev/null | tail -4 > $O/shape_fuzz.txt
cat $O/gpu_tests_final.txt; tail -3 $O/shape_fuzz.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4/fuzz_parity.json"))
for k,v in d["variants"].items():
    c=v["cost_rel_dev"]; u=v["control_update_abs_dev_per_env"]
    print(k, "p99 %.2e p999 %.2e below1e-4 %.5f | u max %.2e within %.3f" % (c["p99"],c["p999"],c["frac_below_1e-4"],u["max"],u["frac_within_test_allowance"]))
PY
