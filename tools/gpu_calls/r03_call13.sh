#!/bin/bash
# round-3 GPU call 13: evidence on the final tree — whole GPU suite (with durations), smoke, the driver-style bench line, per-kernel
# stats of the serialized step and of the training step
set -u
R=$GRAFT_REPO_ROOT
cd "$R" || exit 1
O=$R/gpurun_out/r03g; mkdir -p $O
(timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=40 2>&1 | tail -60) > $O/pytest_gpu.txt
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4) > $O/smoke.txt
(timeout 400 python bench.py --steps 20 --warmup 3) > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_ser -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs --also "" --no-concurrent > $O/trace_ser.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_train -o train --output-format csv -- python $R/tools/bench_train_forward.py --full-step --graph --steps 3 > $O/trace_train.log 2>&1
cd $R
find $O -name "*kernel_trace*" -delete; find $O -name "*agent_info*" -delete
tail -5 $O/pytest_gpu.txt; cat $O/smoke.txt; cut -c1-400 $O/bench.json; echo; grep -v rocprofv3 $O/trace_train.log | tail -2 | cut -c1-700
