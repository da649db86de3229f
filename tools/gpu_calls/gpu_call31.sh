#!/bin/bash
O=gpurun_out/c31; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "clip_runner" > $O/pytest_runner.txt 2>&1; echo "runner tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_runner.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --also "" > $O/bench_x.json 2> $O/bench_x.err; echo "bench rc=$?" | tee -a $O/summary.txt
grep -o '"ms_per_step": [0-9.]*' $O/bench_x.json | head -1
