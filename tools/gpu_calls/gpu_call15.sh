#!/bin/bash
# persistent LSTM layer v2 (scoped accesses instead of fences): kernel tests, breakdown, throughput
O=gpurun_out/c15; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_lstm_gpu.py -x -q -s -k "lstm_layer" > $O/pytest_layer.txt 2>&1; echo "layer tests rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest_layer.txt
timeout 600 python tools/bench_lstm.py --steps 3 --layer-only --no-cpu > $O/bench_lstm.json 2> $O/bench_lstm.err; echo "bench_lstm rc=$?" | tee -a $O/summary.txt
cat $O/bench_lstm.json; tail -5 $O/bench_lstm.err
