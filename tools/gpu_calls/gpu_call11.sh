#!/bin/bash
# round-2 GPU call 11: prefetch-variant A/B, then the round's evidence set on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c11; mkdir -p $O
export TMPDIR=/tmp
for v in 2 0; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --also "" --gemm-variant $v > $O/bench_variant$v.json 2> $O/bench_variant$v.err; echo "bench variant $v rc=$?" | tee -a $O/summary.txt
done
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout 400 python tools/bench_lstm.py --steps 3 > $O/bench_lstm.json 2> $O/bench_lstm.err; echo "bench_lstm rc=$?" | tee -a $O/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace_serial" -o bench --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --also "" --no-concurrent > "$GRAFT_REPO_ROOT/$O/trace_serial.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace_conc" -o bench --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --also "" > "$GRAFT_REPO_ROOT/$O/trace_conc.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find $O -name "*kernel_trace*" -delete
tail -n 4 $O/pytest_gpu.txt; tail -n 1 $O/smoke.txt; cut -c1-300 $O/bench.json; cat $O/bench_lstm.json | cut -c1-250
for f in $O/bench_variant*.json; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; done
