#!/bin/bash
O=gpurun_out/c26; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_train_forward_gpu.py -x -q -s > $O/pytest_train.txt 2>&1; echo "train tests rc=$?" | tee -a $O/summary.txt
tail -30 $O/pytest_train.txt
