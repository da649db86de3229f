#!/bin/bash
# round-2 GPU call 12: fused block 0 with LDS-staged filters, large-N arg-min kernel, bench with calibrated event markers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c12; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "slab or block0 or vq_argmin or nan" > $O/pytest_kernels.txt 2>&1; echo "kernel tests rc=$?" | tee -a $O/summary.txt
timeout 300 python tools/bench_slab.py 2>&1 | grep -v amdgpu.ids > $O/bench_slab.txt
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "batch64 or clip_fp32 or vq_model_api" > $O/pytest_parity.txt 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace_serial" -o bench --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --also "" --no-concurrent > "$GRAFT_REPO_ROOT/$O/trace_serial.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace_conc" -o bench --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --also "" > "$GRAFT_REPO_ROOT/$O/trace_conc.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find $O -name "*kernel_trace*" -delete
tail -n 3 $O/pytest_kernels.txt $O/pytest_parity.txt; head -3 $O/bench_slab.txt; cut -c1-300 $O/bench.json
