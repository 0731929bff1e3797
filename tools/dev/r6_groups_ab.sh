#!/bin/bash
# Round 6: the grouped C4 / C3 with the per-step all-gather - side-stream forms A/B (one rank on real RCCL)
O=gpurun_out/r6g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_two_rank_gather.py tests/test_gpu_boundary.py -m gpu -x -q 2>&1 | tail -3
timeout 400 python tools/dev/groups_gather_cost.py --reps 3 2>/dev/null | tee $O/groups_gather_cost_C4.txt
CPMPPI_COMM_WAITER=stream-ops timeout 400 python tools/dev/groups_gather_cost.py --reps 3 2>/dev/null | tee $O/groups_gather_cost_C4_streamops.txt
CPMPPI_COMM_SIDE_PRIORITY=normal timeout 400 python tools/dev/groups_gather_cost.py --reps 3 2>/dev/null | tee $O/groups_gather_cost_C4_normalprio.txt
timeout 400 python tools/dev/groups_gather_cost.py --reps 3 --config C3 --steps 200 2>/dev/null | tee $O/groups_gather_cost_C3.txt
