#!/bin/bash
# round 6, call 9: kernel stats of the captured training step with the forward's Linear contractions on EMAGE_H2 operands (1) and on EMAGE_F16X3 (0)
O=$GRAFT_REPO_ROOT/gpurun_out/r06_call9; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/train$v -o train --output-format csv -- python $R/tools/bench_train_step.py --quick --h2-forward $v > $O/train$v.log 2>&1; echo "train$v rc=$?"
  cp $(find $O/train$v -name '*kernel_stats.csv' | head -1) $O/train_h2f${v}_kernel_stats.csv
  rm -rf $O/train$v
done
