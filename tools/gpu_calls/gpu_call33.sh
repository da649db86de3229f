#!/bin/bash
O=gpurun_out/c33; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_lstm_gpu.py -x -q -k "graph_runner or baseline_batch" > $O/pytest_lstm_runner.txt 2>&1; echo "lstm runner tests rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_lstm_runner.txt
timeout 300 python tools/bench_lstm.py --steps 3 --no-cpu > $O/bench_lstm.json 2> $O/bench_lstm.err; echo "bench_lstm rc=$?" | tee -a $O/summary.txt
cut -c1-200 $O/bench_lstm.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also "" > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python -c "
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][0]); print(d['ms_per_step'], d['roofline']['traffic'], d['roofline']['frac'])"
