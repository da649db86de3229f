#!/bin/bash
# round-4 GPU call 21: the final defaults once more — kernel tests, the training GPU suite, the training-step bench object
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c21; mkdir -p $O
cd $R
(timeout 400 python -m pytest tests/test_train_forward_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 4) > $O/pytest_train.txt
(timeout 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 3) > $O/pytest_kernels.txt
(timeout 150 python tools/bench_train_step.py 2>&1 | grep -v amdgpu.ids | tail -n 1) > $O/train_step.json
tail -n 2 $O/pytest_train.txt; tail -n 2 $O/pytest_kernels.txt; cut -c1-420 $O/train_step.json
