B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --also '' --no-roofline"
for v in "--pipeline 2" "--pipeline 3" "--pipeline 4" "--pipeline 2 --no-split-acts" "--pipeline 2 --no-concurrent" "--pipeline 3 --no-concurrent"; do
  echo "== $v" >> gpurun_out/r03_t4.txt
  eval $B $v 2>gpurun_out/r03_t4.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('pipelined'))" >> gpurun_out/r03_t4.txt
done
tail -5 gpurun_out/r03_t4.err >> gpurun_out/r03_t4.txt
