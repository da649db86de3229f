#!/bin/bash
# round-4 GPU call 16: evidence on the final tree — whole GPU suite (with durations), smoke, the driver-style bench line, per-kernel stats of the
# serialized inference step and of the training step
set -u
R=$GRAFT_REPO_ROOT
cd "$R" || exit 1
O=$R/gpurun_out/r04_c16; mkdir -p $O
(timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 2>&1 | tail -45) > $O/pytest_gpu.txt
(timeout 400 python bench.py --steps 20 --warmup 5) > $O/bench.json 2> $O/bench.err
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4) > $O/smoke.txt
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats -d $O/trace_ser -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs --also "" --no-concurrent > $O/trace_ser.log 2>&1
timeout 250 rocprofv3 --kernel-trace --stats -d $O/trace_train -o train --output-format csv -- python $R/tools/bench_train_step.py > $O/trace_train.log 2>&1
cd $R
find $O -name "*kernel_trace*" -delete; find $O -name "*agent_info*" -delete
tail -n 6 $O/pytest_gpu.txt; cat $O/smoke.txt; cut -c1-500 $O/bench.json; echo; grep -v rocprofv3 $O/trace_train.log | tail -n 1 | cut -c1-400
