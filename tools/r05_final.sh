#!/bin/bash
# round-5 record run: the default bench line (as the driver runs it), the step profile, the roofline leg profile, the per-kernel counter table
set -e
python3 bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err || { tail -20 gpurun_out/r05_bench.err; exit 1; }
cat gpurun_out/r05_bench.json
tools/profile_step.sh r05 > gpurun_out/r05_profile.log 2>&1
TRACE=$(find gpurun_out/prof_r05 -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $TRACE > gpurun_out/r05_timeline.txt
python3 tools/step_gaps.py $TRACE 2 >> gpurun_out/r05_profile.log
rm -rf gpurun_out/prof_r05
tail -10 gpurun_out/r05_profile.log
tools/profile_roofline.sh r05 > gpurun_out/r05_roofline.log 2>&1 || true
tail -5 gpurun_out/r05_roofline.log
tools/step_budget_pmc.sh r05 > gpurun_out/r05_budget.log 2>&1 || true
rm -rf gpurun_out/pmc_step_r05
tail -3 gpurun_out/r05_budget.log
