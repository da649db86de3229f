cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -o train -- python $GRAFT_REPO_ROOT/tools/bench_train_forward.py --full-step --graph --precision f16x3 --steps 3 > $GRAFT_REPO_ROOT/gpurun_out/r03_train_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_train_prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_train_kernel_stats.csv
find gpurun_out/prof_train -type f -size +2M -delete
head -40 gpurun_out/r03_train_kernel_stats.csv | cut -c1-200
