#!/bin/bash
# round-4 GPU call 14: where emage_grad_prep's transposed image differs from h2_cast(transpose) (diagnostic), kernel stats of the leaner training step
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c14; mkdir -p $O
cd $R
(timeout 120 python tools/diag/grad_prep_probe.py 2>&1 | grep -v amdgpu.ids) > $O/probe.txt
cat $O/probe.txt
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o train --output-format csv -- python $R/tools/bench_train_step.py > $O/train_prof.log 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/train_kernel_stats.csv
rm -rf $O/prof
head -30 $O/train_kernel_stats.csv | cut -c1-170
