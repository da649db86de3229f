#!/bin/bash
# round-6 GPU call 6: the LayerNorm fold v3 (coalesced statistics in front of the DMA prologue) — interleaved step A/B + per-family split
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_c6; mkdir -p $O
cd $R
BQ="--no-cpu-baseline --no-other-configs --also '' --no-roofline"
for arm in fold nofold fold nofold fold nofold; do
  flag=""; [ $arm = nofold ] && flag="--no-fold-ln"
  eval timeout 200 python bench.py --steps 20 --warmup 5 $flag $BQ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('arm $arm ms_per_step', round(d['ms_per_step'],3))" | tee -a $O/step_ab_fold.txt
done
BQ="--no-cpu-baseline --no-other-configs --also ''"
for arm in fold nofold; do
  flag=""; [ $arm = nofold ] && flag="--no-fold-ln"
  eval timeout 300 python bench.py --steps 20 --warmup 5 $flag $BQ 2>/dev/null > $O/bench_$arm.json
  python - <<PY
import json
d = json.loads(open("$O/bench_$arm.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$arm", "ms_per_step", round(d["ms_per_step"], 3), "serialized", round(r["serialized_kernel_ms"], 3), "gemm launches", r["launches_per_step"], "avg us", round(r["avg_launch_us"], 2), "frac", round(r["frac"], 4))
print("   ", {k: v for k, v in list(r["kernel_time_ms_by_family"].items())[:5]})
PY
done
