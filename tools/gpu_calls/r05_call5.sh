#!/bin/bash
# round-5 GPU call 5: A/B of the captured training step on one box — batched finalize launches on / off, the 8-wave dispatch for long 768-wide
# contractions on / off (emage_set_tuning key 5: 131072 = neutral, 65536 = the dispatch rule off); tools library in every arm
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c5; mkdir -p $O
cd $R
for arm in "--h2-variant 131072" "--h2-variant 131072 --defer-finalize 0" "--h2-variant 65536" "--h2-variant 131072" "--h2-variant 131072 --defer-finalize 0" "--h2-variant 65536"; do
  (timeout 200 python tools/bench_train_step.py --quick $arm 2>&1 | grep -v amdgpu.ids | tail -n 1) > $O/arm.json
  python - "$arm" <<PY
import json, sys
try:
    d = json.loads(open("$O/arm.json").read().strip().splitlines()[-1])
    print("%-45s ms_per_step %.2f peak %.2f GB loss %.6f" % (sys.argv[1], d["ms_per_step"], d["peak_memory_gb"], d["loss_all_after_replays"]))
except Exception as e:
    print("arm", sys.argv[1], "failed", e, open("$O/arm.json").read()[-300:])
PY
done | tee $O/train_ab.txt
