#!/bin/bash
# LSTM cell on the hardware transcendental units: all LSTM tests, breakdown, throughput
O=gpurun_out/c19; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -s > $O/pytest_lstm.txt 2>&1; echo "lstm tests rc=$?" | tee -a $O/summary.txt
tail -12 $O/pytest_lstm.txt
timeout 600 python tools/bench_lstm.py --steps 3 --layer-only > $O/bench_lstm.json 2> $O/bench_lstm.err; echo "bench_lstm rc=$?" | tee -a $O/summary.txt
cat $O/bench_lstm.json; tail -3 $O/bench_lstm.err
