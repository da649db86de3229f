#!/bin/bash
# round-5 GPU call 20: the whole GPU suite and smoke on the FINAL tree (after the dispatch-order tile walk for kv_all shipped)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c20; mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 2>&1 | grep -v amdgpu.ids | tail -n 30) > $O/pytest_gpu.txt; tail -n 6 $O/pytest_gpu.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 3) > $O/smoke.txt; cat $O/smoke.txt
