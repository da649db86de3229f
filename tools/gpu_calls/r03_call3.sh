python -m pytest tests/test_parity_gpu.py -x -q -k "forward_window_fp32 or clip_fp32_matches or batch64" 2>&1 | tail -8 > gpurun_out/r03_t3.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --also '' --no-roofline"
for v in "--no-split-acts" "" "--h2-residual" "--no-split-acts" "" "--h2-residual"; do
  echo "== $v" >> gpurun_out/r03_t3.txt
  eval $B $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/r03_t3.txt
done
eval $B --no-concurrent 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial h2', d['ms_per_step'])" >> gpurun_out/r03_t3.txt
eval $B --no-concurrent --h2-residual 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial h2res', d['ms_per_step'])" >> gpurun_out/r03_t3.txt
