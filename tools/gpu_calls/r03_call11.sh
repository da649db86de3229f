#!/bin/bash
# round-3 GPU call 11: the data-as-flag hand-over of emage_lstm_layer (parity tests, DisCo / CaMN timing with per-phase ablations and
# round 2's counter protocol beside it), the parallel finalize kernels + new tests of the training step, the fp32 fallback of ClipRunner
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
(timeout 600 python -m pytest tests/test_lstm_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -8) > $O/r03_t11.txt
(timeout 400 python tools/bench_lstm.py --steps 3 --no-cpu --layer-only 2>&1 | grep -v amdgpu.ids) > $O/r03_bench_lstm11.txt
(timeout 200 python -m pytest tests/test_parity_gpu.py -x -q -p no:cacheprovider -k "overflow or graph" 2>&1 | tail -5) >> $O/r03_t11.txt
(timeout 900 python -m pytest tests/test_train_forward_gpu.py -x -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -12) >> $O/r03_t11.txt
(timeout 300 python tools/bench_train_forward.py --full-step --graph --steps 3 2>/dev/null) > $O/r03_train11.json
cat $O/r03_t11.txt; cut -c1-1800 $O/r03_bench_lstm11.txt; cut -c1-600 $O/r03_train11.json
