#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gpu_tests2.txt
python tools/dev/gen_time.py 256 > $O/gen_time.txt 2>&1
cat $O/gpu_tests2.txt | tail -2; cat $O/gen_time.txt | grep -v amdgpu.ids
