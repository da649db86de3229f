#!/bin/bash
O=gpurun_out/c29; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for b in 8 56; do
  timeout 900 python tools/bench_train_forward.py --steps 2 --batch $b --full-step > $O/bench_train_step_b$b.json 2> $O/bench_train_step_b$b.err; echo "bench train step b=$b rc=$?" | tee -a $O/summary.txt
  cat $O/bench_train_step_b$b.json; tail -3 $O/bench_train_step_b$b.err
done
