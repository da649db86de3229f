#!/bin/bash
O=gpurun_out/c28; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_train_forward_gpu.py -x -q -s -k "training_step" > $O/pytest_train_step.txt 2>&1; echo "train step test rc=$?" | tee -a $O/summary.txt
tail -30 $O/pytest_train_step.txt
