#!/bin/bash
# round-4 GPU call 19: staggered waveform features (model.stagger_audio) — bit-identity test, A/B of the timed step on one box
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c19; mkdir -p $O
cd $R
(timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k "staggered" 2>&1 | tail -n 8) > $O/pytest_stagger.txt
BQ="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --also ''"
for v in 0 1 0 1; do
  eval timeout 200 python bench.py $BQ --stagger-audio $v > $O/bench_st$v.json 2> $O/bench_st$v.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_st$v.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("stagger_audio $v: ms_per_step %.3f pcie_inclusive %.3f serialized %.3f gemm %.3f launches %d frac %.4f" % (d["ms_per_step"], d["pcie_inclusive"]["ms_per_step"], r["serialized_kernel_ms"], r["kernel_time_ms_by_family"]["emage_gemm"], r["launches_per_step"], r["frac"]))
except Exception as e:
    print("stagger_audio $v: failed", e)
PY
done | tee $O/ab_stagger.txt
tail -n 6 $O/pytest_stagger.txt
