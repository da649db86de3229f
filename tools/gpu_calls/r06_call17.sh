#!/bin/bash
# round 6, call 17: kernel stats of the captured training step with the direct convolution input gradients
O=$GRAFT_REPO_ROOT/gpurun_out/r06_call17; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/train -o train --output-format csv -- python $R/tools/bench_train_step.py --quick > $O/train.log 2>&1; echo "train rc=$?"
cp $(find $O/train -name '*kernel_stats.csv' | head -1) $O/train_kernel_stats.csv; rm -rf $O/train
grep "^{" $O/train.log | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
