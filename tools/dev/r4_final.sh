#!/bin/bash
# Round 4, final tree: every piece of evidence profiles/r4/README.md quotes (profiles, A/B against the round start, randomised
# differential tests, the default bench line).  `profiles` / `checks` select one half.
mkdir -p gpurun_out/r4
if [ "${1:-all}" != "checks" ]; then
bash tools/profile.sh r4 > gpurun_out/r4/profile_sh.log 2>&1
bash tools/profile_cfg.sh r4 0 > gpurun_out/r4/profile_cfg_sh.log 2>&1
tail -2 gpurun_out/r4/profile_sh.log gpurun_out/r4/profile_cfg_sh.log
fi
if [ "${1:-all}" != "profiles" ]; then
bash tools/dev/r4_ab.sh > /dev/null 2>&1
bash tools/dev/r4_fuzz.sh 2>&1 | tail -8
bash tools/dev/r4_check.sh 2>&1 | tail -9
fi
