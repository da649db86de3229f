#!/bin/bash
# LSTM layer with the block-exclusive exchange buffer: tests, breakdown, A/B
O=gpurun_out/c21; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -s > $O/pytest_lstm.txt 2>&1; echo "lstm tests rc=$?" | tee -a $O/summary.txt
tail -12 $O/pytest_lstm.txt
timeout 600 python tools/bench_lstm.py --steps 3 --layer-only --no-cpu > $O/bench_lstm.json 2> $O/bench_lstm.err; echo "bench_lstm rc=$?" | tee -a $O/summary.txt
cat $O/bench_lstm.json; tail -3 $O/bench_lstm.err
timeout 600 python tools/bench_lstm.py --steps 3 --no-cpu --no-exchange > $O/bench_lstm_noxch.json 2> $O/bench_lstm_noxch.err; echo "bench_lstm no-exchange rc=$?" | tee -a $O/summary.txt
cat $O/bench_lstm_noxch.json
