#!/bin/bash
# round-6 GPU call 5: where the LayerNorm fold's time goes — the per-family split of the serialized step, fold on / off
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_c5; mkdir -p $O
cd $R
BQ="--no-cpu-baseline --no-other-configs --also ''"
for arm in fold nofold; do
  flag=""; [ $arm = nofold ] && flag="--no-fold-ln"
  eval timeout 300 python bench.py --steps 20 --warmup 5 $flag $BQ 2>/dev/null > $O/bench_$arm.json
  python - <<PY
import json
d = json.loads(open("$O/bench_$arm.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$arm", "ms_per_step", round(d["ms_per_step"], 3), "serialized", round(r["serialized_kernel_ms"], 3), "gemm launches", r["launches_per_step"], "avg us", round(r["avg_launch_us"], 2), "frac", round(r["frac"], 4))
print("   ", r["kernel_time_ms_by_family"])
print("   ", r["launches_by_family"])
PY
done
