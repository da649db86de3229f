#!/bin/bash
# round-6 GPU call 1: the ANTIPHASE tiles (gemm_h2_pp.hip, configs 300-331) — validation against float64 + bit identity with config 120,
# timing sweep at the window shapes (twice: a race screen), the phase trace of config 350; this round's baseline bench line (new objects:
# other_precisions.bf16.roofline / code_agreement, cpu_baseline.b64); a step A/B with the 768-wide launches on config 300
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_c1; mkdir -p $O
cd $R
for rep in 1 2; do
timeout 400 python tools/bench_gemm_h2.py --configs 120,100,300,301,302,303,320,321,331 --shapes "out_proj,ffn2,mlp 256,fc 512,ragged" > $O/pp_sweep_narrow_$rep.txt 2>&1; echo "narrow$rep rc=$?"
done
timeout 400 python tools/bench_gemm_h2.py --configs 100,170,310,311,312,320,321 --shapes "qkv 768,ffn1,kv_part" > $O/pp_sweep_wide.txt 2>&1; echo "wide rc=$?"
timeout 300 python tools/bench_gemm_h2.py --configs 120,330,331 --shapes "head 768,conv3 256->256,cls fc,conv3 337" > $O/pp_sweep_heads.txt 2>&1; echo "heads rc=$?"
timeout 120 python tools/trace_gemm_h2.py 350,201 > $O/pp_phase_trace.txt 2>&1; echo "trace rc=$?"
tail -n 8 $O/pp_sweep_narrow_1.txt $O/pp_sweep_wide.txt $O/pp_sweep_heads.txt
BQ="--no-cpu-baseline --no-other-configs --also '' --no-roofline"
for arm in 0 300 0 300 302 303; do
eval timeout 200 python bench.py --steps 20 --warmup 5 --tools-lib --h2-pp $arm $BQ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('arm $arm ms_per_step', round(d['ms_per_step'],3))" | tee -a $O/step_ab_pp.txt
done
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r06_c1")
d = json.loads(open(os.path.join(O, "bench.json")).read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "frac", d.get("roofline", {}).get("frac"), "avg_launch_us", d.get("roofline", {}).get("avg_launch_us"))
print("bf16", json.dumps(d.get("other_precisions", {}).get("bf16", {}))[:1500])
print("code_agreement", d.get("code_agreement"))
print("cpu", json.dumps(d.get("cpu_baseline"))[:600])
print("train", d.get("train_step", {}).get("ms_per_step"), "lstm", {k: v.get("ms_per_step") for k, v in d.get("lstm_models", {}).items() if isinstance(v, dict)})
PY
