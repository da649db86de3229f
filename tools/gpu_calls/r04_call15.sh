#!/bin/bash
# round-4 GPU call 15: accumulating dW contractions (res == out_f32), split-K up to 384 tiles, the relaxed grad_prep test; step time; where the
# remaining ATen launches of a step come from (dispatch census with the real operators)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c15; mkdir -p $O
cd $R
(timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6) > $O/pytest_kernels.txt
(timeout 400 python -m pytest tests/test_train_forward_gpu.py -m gpu -q -x -p no:cacheprovider -k "baseline_batch or captured or device_drawn or mul_add or training_step_matches or backward_matches or health" 2>&1 | tail -6) > $O/pytest_train.txt
(timeout 200 python tools/bench_train_step.py 2>&1 | grep -v amdgpu.ids | tail -1) > $O/train_step.json
(timeout 200 python tools/diag/aten_census.py f16x3 70 cuda 2>&1 | grep -v amdgpu.ids) > $O/census.txt
tail -n 4 $O/pytest_kernels.txt; tail -n 4 $O/pytest_train.txt; cut -c1-330 $O/train_step.json; echo; head -100 $O/census.txt
