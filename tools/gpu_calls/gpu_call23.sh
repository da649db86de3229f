#!/bin/bash
# train-mode forward: kernels, forward vs oracle, three forwards of a step vs the reference's losses
O=gpurun_out/c23; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_train_forward_gpu.py -x -q -s > $O/pytest_train.txt 2>&1; echo "train tests rc=$?" | tee -a $O/summary.txt
tail -30 $O/pytest_train.txt
