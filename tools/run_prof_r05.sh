set -e
tools/profile_step.sh r05base > gpurun_out/r05base_profile.log 2>&1
TRACE=$(find gpurun_out/prof_r05base -name "*kernel_trace.csv" | head -1)
cp $TRACE gpurun_out/r05base_kernel_trace.csv
python3 tools/step_timeline.py $TRACE > gpurun_out/r05base_timeline.txt
python3 tools/step_gaps.py $TRACE 2 >> gpurun_out/r05base_profile.log
tail -5 gpurun_out/r05base_profile.log
