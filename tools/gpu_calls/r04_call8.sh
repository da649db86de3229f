#!/bin/bash
# round-4 GPU call 8: per-step breakdown of the persistent LSTM layer: pre-split exchange vs round 3's float32 exchange on one box
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c8; mkdir -p $O
cd $R
timeout 400 python tools/bench_lstm.py --layer-only --no-cpu > $O/bench_lstm.json 2> $O/bench_lstm.err; echo "bench rc=$?" | tee $O/summary.txt
cat $O/bench_lstm.json | cut -c1-3000
