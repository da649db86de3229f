#!/bin/bash
# round-5 GPU call 4: batched finalize launches (kernel + step test, A/B of the captured step), the 8-wave dispatch for long 768-wide
# contractions (A/B: emage_set_tuning key 5 bit 65536 = off), the reference-step tests on the new tree
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c4; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_train_forward_gpu.py -m gpu -q -s -p no:cacheprovider -k "batched_finalize or test_training_step_matches_the_reference or captured_f16x3_step or backward_kernels" 2>&1 | grep -v amdgpu.ids | tail -n 25) > $O/pytest_train.txt; tail -n 6 $O/pytest_train.txt
for arm in "--h2-variant 131072" "--h2-variant 131072 --defer-finalize 0" "--h2-variant 65536" "--h2-variant 131072" "--h2-variant 131072 --defer-finalize 0" "--h2-variant 65536"; do
  (timeout 120 python tools/bench_train_step.py --quick $arm 2>&1 | grep -v amdgpu.ids | tail -n 1) > $O/arm.json
  python - "$arm" <<PY
import json, sys
try:
    d = json.loads(open("$O/arm.json").read().strip().splitlines()[-1])
    print("%-45s ms_per_step %.2f peak %.2f GB loss %.6f" % (sys.argv[1], d["ms_per_step"], d["peak_memory_gb"], d["loss_all_after_replays"]))
except Exception as e:
    print("arm", sys.argv[1], "failed", e)
PY
done | tee $O/train_ab.txt
