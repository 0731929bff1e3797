#!/bin/bash
# Round-4 GPU check: the -m gpu suite, smoke, the default bench line (with its `verified` objects).  Outputs under gpurun_out/r4/.
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/r4/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/smoke.txt 2>&1
python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err
echo "bench rc=$?" >> gpurun_out/r4/bench_default.err
tail -4 gpurun_out/r4/gpu_tests.txt; tail -2 gpurun_out/r4/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_default.json').read().strip().splitlines()[-1])
print("value %.4g  ms %.4f kernel_ms %.4f valu_frac %.3f verified %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline_valu"]["frac"], d["verified"]["ok"]))
for k,v in d["configs"].items(): print(k, "%.4g" % v["value"], "%.4f ms" % v["ms_per_step"], "kernel %.4f" % v["kernel_ms"], v["verified"]["ok"], v["verified"].get("worst_clear_excess"))
print("single", d["single_env"]["us_per_step"], d["single_env"]["rollout_kernel_us"], d["single_env"].get("host_seam",{}).get("us_per_call"))
PY
