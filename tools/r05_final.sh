#!/bin/bash
# round-5 record run: the default bench line (as the driver runs it), the step profile + timeline, the roofline leg profile, the per-kernel counter table
set -e
TAG=${1:-r05}
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -20 gpurun_out/${TAG}_bench.err; exit 1; }
cat gpurun_out/${TAG}_bench.json
tools/profile_step.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
TRACE=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $TRACE > gpurun_out/${TAG}_timeline.txt
python3 tools/step_gaps.py $TRACE 2 >> gpurun_out/${TAG}_profile.log
rm -rf gpurun_out/prof_$TAG
tail -10 gpurun_out/${TAG}_profile.log
tools/profile_roofline.sh $TAG > gpurun_out/${TAG}_roofline.log 2>&1 || true
tail -5 gpurun_out/${TAG}_roofline.log
tools/step_budget_pmc.sh $TAG > gpurun_out/${TAG}_budget.log 2>&1 || true
rm -rf gpurun_out/pmc_step_$TAG
tail -3 gpurun_out/${TAG}_budget.log
