#!/bin/bash
# round-4 GPU call 9: fewer-waves / larger-wave-tile EMAGE_H2 configurations (160-171, tools build) against the shipped ones, per shape,
# then the whole step under the heuristic variants that route the 768-wide launches to them (A/B on one box)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c9; mkdir -p $O
cd $R
(timeout 300 python tools/bench_gemm_h2.py --configs 120,100,124,160,161,162,163,164,165,166,167,168,169 --shapes "out_proj,ffn2,head 768,mlp 256,fc 512,ragged,conv3 256->256 +res" 2>&1 | grep -v amdgpu.ids) > $O/sweep_narrow.txt
(timeout 300 python tools/bench_gemm_h2.py --configs 100,119,113,116,168,169,170,171 --shapes "qkv 768->2304,ffn1,kv_all,kv_part" 2>&1 | grep -v amdgpu.ids) > $O/sweep_wide.txt
BQ="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --also ''"
for v in 0 16 32 64 128 256 0; do
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
cat $O/sweep_narrow.txt $O/sweep_wide.txt
