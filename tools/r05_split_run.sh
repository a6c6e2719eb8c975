set -e
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "split_sweeps" > gpurun_out/r05k_tests.log 2>&1 || { tail -40 gpurun_out/r05k_tests.log; exit 1; }
tail -2 gpurun_out/r05k_tests.log
for PASS in 1 2; do for X in 0 192; do
  SMBV_PAIR_SPLIT=$X python3 bench.py --vit-large --fp8 --recompute --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-roofline --no-feed --no-trainer --no-extra-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass $PASS vit-large fp8 recompute split $X ms_per_step %.2f' % d['ms_per_step'])"
  SMBV_PAIR_SPLIT=$X python3 bench.py --vjepa --steps 5 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pass $PASS vjepa split $X ms_per_step %.2f' % d['ms_per_step'])"
done; done
STEPS=20 tools/pair_split_ab.sh 0 192
