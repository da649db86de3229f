#!/bin/bash
# round-5 GPU call 15: the LSTM GPU suite and smoke on the final tree (the EMAGE_H2 input-projection switch off by default again)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c15; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_lstm_gpu.py tests/test_abi.py -m "gpu or not gpu" -q -p no:cacheprovider 2>&1 | tail -n 5) > $O/pytest_lstm.txt; tail -n 3 $O/pytest_lstm.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 2) > $O/smoke.txt; cat $O/smoke.txt
