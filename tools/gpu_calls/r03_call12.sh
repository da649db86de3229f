#!/bin/bash
# round-3 GPU call 12: fast LSTM cell non-linearities (f16x3) — LSTM parity tests + timing; probe of the fp32 fallback; WavEncoder block test
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
(timeout 600 python -m pytest tests/test_lstm_gpu.py -x -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -14) > $O/r03_t12.txt
(timeout 400 python tools/bench_lstm.py --steps 3 --layer-only 2>&1 | grep -v amdgpu.ids) > $O/r03_bench_lstm12.txt
(timeout 300 python tools/diag/fallback_probe.py 2>&1 | grep -v amdgpu.ids | tail -12) > $O/r03_probe12.txt
(timeout 300 python -m pytest tests/test_train_forward_gpu.py -x -q -s -p no:cacheprovider -k "block_by_block" 2>&1 | grep -v "^$" | tail -8) >> $O/r03_t12.txt
cat $O/r03_t12.txt; cut -c1-2000 $O/r03_bench_lstm12.txt; cat $O/r03_probe12.txt
