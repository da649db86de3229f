#!/bin/bash
# round-2 GPU call 3: view-based kernels (no ATen in the step), split-f16 attention, new bench line, serialized trace
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c3; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q > $O/pytest_kernels.txt 2>&1; echo "kernels rc=$?" | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -s > $O/pytest_parity.txt 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace_x3" -o bench --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --also "" > "$GRAFT_REPO_ROOT/$O/trace_x3.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace_x3_serial" -o bench --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --also "" --no-concurrent > "$GRAFT_REPO_ROOT/$O/trace_x3_serial.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find $O -name "*kernel_trace*" -delete
tail -3 $O/pytest_kernels.txt $O/pytest_parity.txt; cat $O/smoke.txt | tail -2; cut -c1-600 $O/bench.json; tail -5 $O/bench.err
