#!/bin/bash
# round-4 GPU call 7: the persistent LSTM layer with the pre-split exchange (packed fp16 hi | lo words in a separate scratch): bitwise
# tests, then the per-step breakdown with the round-3 float32 exchange beside it, and the two models' throughput
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c7; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_lstm_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest_lstm.txt 2>&1; echo "pytest rc=$?" | tee $O/summary.txt
timeout 400 python tools/bench_lstm.py > $O/bench_lstm.json 2> $O/bench_lstm.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_lstm.txt; cat $O/bench_lstm.json | cut -c1-3000; tail -3 $O/bench_lstm.err
