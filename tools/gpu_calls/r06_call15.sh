#!/bin/bash
# round 6, call 15: sub_batches = 1 / 2 / 4 at the BASELINE batch on the final tree (LayerNorm fold, grouped launches)
O=gpurun_out/r06_call15; mkdir -p $O
timeout 600 python tools/bench_sub_batches.py > $O/sub_batches.json 2> $O/sub_batches.err; echo "rc=$?"; cat $O/sub_batches.json; tail -3 $O/sub_batches.err
