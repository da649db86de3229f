#!/bin/bash
# persistent LSTM layer kernel: kernel tests (bitwise vs per-step), model tests, throughput A/B
O=gpurun_out/c14; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_lstm_gpu.py -x -q -s -k "lstm_layer" > $O/pytest_layer.txt 2>&1; echo "layer tests rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest_layer.txt
timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -s > $O/pytest_lstm.txt 2>&1; echo "lstm tests rc=$?" | tee -a $O/summary.txt
tail -8 $O/pytest_lstm.txt
timeout 600 python tools/bench_lstm.py --steps 3 --layer-only > $O/bench_lstm.json 2> $O/bench_lstm.err; echo "bench_lstm rc=$?" | tee -a $O/summary.txt
cat $O/bench_lstm.json
timeout 600 python tools/bench_lstm.py --steps 3 --per-step --no-cpu > $O/bench_lstm_per_step.json 2> $O/bench_lstm_per_step.err; echo "bench_lstm per-step rc=$?" | tee -a $O/summary.txt
cat $O/bench_lstm_per_step.json
