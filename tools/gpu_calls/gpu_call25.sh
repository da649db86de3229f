#!/bin/bash
O=gpurun_out/c25; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_train_forward_gpu.py -x -q -s -k "backward or gradients" > $O/pytest_train_bwd.txt 2>&1; echo "train bwd tests rc=$?" | tee -a $O/summary.txt
tail -30 $O/pytest_train_bwd.txt
