#!/bin/bash
# round 6, call 13: torch.profiler origins of the small ATen kernels of one eager training step
O=gpurun_out/r06_call13; mkdir -p $O
timeout 600 python tools/diag/train_kernel_origins.py > $O/train_kernel_origins.txt 2>&1; echo "rc=$?"; tail -60 $O/train_kernel_origins.txt
