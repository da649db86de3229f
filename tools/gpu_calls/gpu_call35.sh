#!/bin/bash
O=gpurun_out/c35; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 500 python -m pytest tests/test_train_forward_gpu.py -x -q -s -k "captured or loss_kernels" > $O/pytest_capture.txt 2>&1; echo "capture test rc=$?" | tee -a $O/summary.txt
tail -25 $O/pytest_capture.txt | cut -c1-300
