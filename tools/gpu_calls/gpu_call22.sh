#!/bin/bash
# evidence set on the current tree: whole GPU suite, smoke, bench (incl. roofline + CPU baseline), serialized kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c22; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace_serial" -o bench --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --also "" --no-concurrent > "$GRAFT_REPO_ROOT/$O/trace_serial.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find $O -name "*kernel_trace*" -delete
tail -n 4 $O/pytest_gpu.txt; tail -n 1 $O/smoke.txt; cut -c1-300 $O/bench.json
