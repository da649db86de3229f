#!/bin/bash
# round-4 GPU call 20: fused LayerNorm backward with 4 rows per block against 16 and against the separate launches (A/B of the captured step on one
# box), its kernel test; then the training GPU suite on the final defaults
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c20; mkdir -p $O
cd $R
(timeout 200 python -m pytest tests/test_train_forward_gpu.py -m gpu -q -x -p no:cacheprovider -k "backward_kernels" 2>&1 | tail -n 4) > $O/pytest_ln.txt
for arm in 0 4 16 0 4; do
  (timeout 150 python tools/bench_train_step.py --quick --ln-fused $arm 2>&1 | grep -v amdgpu.ids | tail -n 1) > $O/arm_$arm.json
  python - <<PY
import json
try:
    d = json.loads(open("$O/arm_$arm.json").read().strip().splitlines()[-1])
    print("ln_fused $arm: ms_per_step %.2f peak %.2f GB loss %.6f" % (d["ms_per_step"], d["peak_memory_gb"], d["loss_all_after_replays"]))
except Exception as e:
    print("arm $arm failed", e)
PY
done | tee $O/ab.txt
tail -n 3 $O/pytest_ln.txt
