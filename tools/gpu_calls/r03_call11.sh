#!/bin/bash
# round-3 GPU call 11: the data-as-flag hand-over of emage_lstm_layer — kernel / model parity tests, then DisCo / CaMN timing with
# the per-phase ablations and round 2's counter protocol beside it (tools library)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
(timeout 600 python -m pytest tests/test_lstm_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -8) > $O/r03_t11.txt
(timeout 400 python tools/bench_lstm.py --steps 3 --no-cpu --layer-only 2>&1 | grep -v amdgpu.ids) > $O/r03_bench_lstm11.txt
cat $O/r03_t11.txt; cat $O/r03_bench_lstm11.txt | cut -c1-1500
