#!/bin/bash
# round-4 GPU call 10: the 128x192 configuration for many-tile wide launches (kernel tests, index-exact parity, A/B of the step with /
# without it), the piece-wise convolution backward (config-3 golden at 56 clips, step time and peak memory)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c10; mkdir -p $O
cd $R
(timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5) > $O/pytest_kernels.txt
(timeout 500 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k "grouped or batch64 or window or clip" 2>&1 | tail -5) > $O/pytest_parity.txt
(timeout 500 python -m pytest tests/test_train_forward_gpu.py -m gpu -q -x -p no:cacheprovider -k "baseline_batch or captured_step or training_step_matches" --durations=5 2>&1 | tail -12) > $O/pytest_train.txt
BQ="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --also ''"
for v in 512 0 512 0; do
  eval timeout 200 python bench.py $BQ --h2-variant $v --gemm-dbg 64 > $O/bench_v$v.json 2> $O/bench_v$v.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_v$v.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("variant $v: ms_per_step %.3f serialized %.3f gemm %.3f launches %d avg_us %.2f" % (d["ms_per_step"], r["serialized_kernel_ms"], r["kernel_time_ms_by_family"]["emage_gemm"], r["launches_per_step"], r["avg_launch_us"]))
except Exception as e:
    print("variant $v: failed", e)
PY
done | tee $O/ab.txt
(timeout 300 python tools/bench_train_step.py 2>&1 | grep -v amdgpu.ids | tail -3) > $O/train_step.txt
tail -3 $O/pytest_kernels.txt $O/pytest_parity.txt; tail -8 $O/pytest_train.txt; cut -c1-700 $O/train_step.txt
