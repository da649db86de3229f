#!/bin/bash
O=gpurun_out/c24; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_train_forward_gpu.py -x -q -s > $O/pytest_train.txt 2>&1; echo "train tests rc=$?" | tee -a $O/summary.txt
tail -14 $O/pytest_train.txt
timeout 600 python tools/bench_train_forward.py --steps 5 > $O/bench_train_forward.json 2> $O/bench_train_forward.err; echo "bench train fwd rc=$?" | tee -a $O/summary.txt
cat $O/bench_train_forward.json; tail -3 $O/bench_train_forward.err
