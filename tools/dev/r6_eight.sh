#!/bin/bash
# Round 6: bench.py --gpus 8 with ALL EIGHT ranks on one device (the library's own communicator bound to the stand-in collective
# library; torch.distributed over gloo): the N = 8 code path - rendezvous, id exchange, 8-block all-gathers, stamps, BASELINE
# configs[3]'s layout (512 envs as 64 per rank) - end to end.  Not a scaling number: the ranks share the GPU.
O=gpurun_out/r6m; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_two_rank_gather.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
CPMPPI_BENCH_BACKEND=gloo CPMPPI_BENCH_ONE_DEVICE=1 CPMPPI_BENCH_COLLECTIVE=native CPMPPI_BENCH_RCCL_PATH=$PWD/tests/fake_rccl/libfake_rccl.so \
  timeout 900 python bench.py --gpus 8 --no-cpu-baseline > $O/bench_8ranks_one_device_fake_rccl.json 2> $O/bench_8ranks.err; echo "8-rank rc $?"
python - <<'PY'
import json
L=[l for l in open('gpurun_out/r6m/bench_8ranks_one_device_fake_rccl.json') if l.startswith('{"metric"')]
d=json.loads(L[-1])
print("n_gpus", d['n_gpus'], "value %.4g" % d['value'], "ms/step", d['ms_per_step'], json.dumps(d['config'].get('collective'))[:700], d['verified']['ok'])
for k,v in d.get('configs',{}).items():
    print("  ", k, {x: v.get(x) for x in ('workload','ms_per_step','without_collective_ms_per_step','collective_cost','error')}, json.dumps(v.get('collective'))[:300], (v.get('verified') or {}).get('ok'))
PY
grep -v "WARN\|^$\|iommu\|amdgpu.ids\|socket.cpp" $O/bench_8ranks.err | tail -15
