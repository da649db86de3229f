B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --also '' --no-roofline"
for v in "--no-split-acts" "" "--h2-variant 1" "--h2-variant 2" "--h2-variant 3" "--h2-variant 4" "--h2-variant 8" "--h2-variant 9"; do
  echo "== $v" >> gpurun_out/r03_ab1.txt
  eval $B $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/r03_ab1.txt
  eval $B $v --no-concurrent 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial', d['ms_per_step'])" >> gpurun_out/r03_ab1.txt
done
