#!/bin/bash
# round-4 GPU call 17: A/B on ONE box of the two backward-contraction changes — dW accumulated by the contraction (vs a temporary + queued add) and
# split-K up to 384 tiles (vs round 3's 191; tools library, emage_set_tuning key 5 bit 1024; bit 2048 is unused and only selects the tools library)
# (as run: key 5 bit 1024 selected a limit of 191 tiles and the default was 384; the tree now ships 100, bit 1024 selects 384 and bit 4096 selects 191)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c17; mkdir -p $O
cd $R
for arm in "1 2048" "1 1024" "0 2048" "0 1024" "1 2048" "1 1024"; do
  set -- $arm
  (timeout 150 python tools/bench_train_step.py --quick --accumulate-dw $1 --h2-variant $2 2>&1 | grep -v amdgpu.ids | tail -n 1) > $O/arm_$1_$2.json
  python - <<PY
import json
try:
    d = json.loads(open("$O/arm_$1_$2.json").read().strip().splitlines()[-1])
    print("accumulate_dw $1 split_k_tiles %s: ms_per_step %.2f peak %.2f GB loss %.6f" % ("191" if $2 == 1024 else "384", d["ms_per_step"], d["peak_memory_gb"], d["loss_all_after_replays"]))
except Exception as e:
    print("arm $1 $2 failed", e)
PY
done | tee $O/ab.txt
