#!/bin/bash
# round 6, call 12: ATen census of one eager training step on the device (which host-side torch calls become the ~2 000 small ATen launches of the captured step)
O=gpurun_out/r06_call12; mkdir -p $O
timeout 600 python tools/diag/train_aten_census.py --second > $O/train_aten_census.txt 2>&1; echo "census rc=$?"; tail -75 $O/train_aten_census.txt
