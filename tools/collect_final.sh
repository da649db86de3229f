#!/bin/bash
# End-of-round evidence on the MI355X box (run from the repo root through gpurun); outputs under gpurun_out/r01f/
set -u
R=$PWD
O=$R/gpurun_out/r01f
mkdir -p $O
(timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8) > $O/pytest_gpu.txt
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3) > $O/smoke.txt
(timeout 300 python bench.py --steps 20 --warmup 3) > $O/bench.json 2> $O/bench.err
(timeout 200 python bench.py --steps 10 --warmup 3 --fused-layers --no-cpu-baseline) > $O/bench_fused_layers.json 2> $O/bench_fused.err
(timeout 200 python tools/bench_layer.py --masks 0) 2>&1 | grep -v amdgpu.ids > $O/bench_layer.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/trace.log 2>&1
cd $R
rm -f $O/trace/bench_kernel_trace.csv
tail -3 $O/pytest_gpu.txt; cat $O/smoke.txt; cut -c1-300 $O/bench.json; echo; cut -c1-300 $O/bench_fused_layers.json; echo; cat $O/bench_layer.txt
