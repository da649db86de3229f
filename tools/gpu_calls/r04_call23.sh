#!/bin/bash
# round-4 GPU call 23 (the last 2 GPU-minutes): split-K at all?  The backward sweep (profiles/r04_gemm_h2_sweep_backward_shapes.txt) has the 144-tile
# split-K weight gradient at 80 us while the 288-tile one, twice the work and NOT split, takes 47.5 us.  A/B of the captured step, tools library:
# key 5 bit 8192 = neutral (191 tiles, as shipped THEN), 2048 = never split, 4096 = split below 100 tiles only (as run; the tree now ships 100 and bit 4096 selects 191)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c23; mkdir -p $O
cd $R
for v in 8192 2048 4096 8192 2048; do
  (timeout 60 python tools/bench_train_step.py --quick --h2-variant $v 2>&1 | grep -v amdgpu.ids | tail -n 1) > $O/arm_$v.json
  python - <<PY
import json
try:
    d = json.loads(open("$O/arm_$v.json").read().strip().splitlines()[-1])
    print("h2_variant $v: ms_per_step %.2f peak %.2f GB loss %.6f" % (d["ms_per_step"], d["peak_memory_gb"], d["loss_all_after_replays"]))
except Exception as e:
    print("arm $v failed", e)
PY
done | tee $O/ab.txt
