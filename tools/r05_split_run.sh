set -e
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "split_sweeps or attention_fwd or optimistic or large_logits" > gpurun_out/r05l_tests.log 2>&1 || { tail -40 gpurun_out/r05l_tests.log; exit 1; }
tail -2 gpurun_out/r05l_tests.log
for PASS in 1 2; do for X in 0 192; do
  SMBV_FWD_SPLIT=$X python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass $PASS fwd split $X ms_per_step %.3f' % d['ms_per_step'])"
  SMBV_FWD_SPLIT=$X python3 bench.py --vjepa --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass $PASS vjepa fwd split $X ms_per_step %.2f' % d['ms_per_step'])"
done; done
