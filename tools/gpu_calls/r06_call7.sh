#!/bin/bash
# round-6 GPU call 7: the whole GPU suite + smoke on the tree with the LayerNorm fold (default) and the one-launch health check; bench line
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_c7; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/pytest_gpu.txt; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r06_c7")
d = json.loads(open(os.path.join(O, "bench.json")).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", r.get("frac"), "avg_launch_us", r.get("avg_launch_us"), "serialized", r.get("serialized_kernel_ms"))
print(r.get("kernel_time_ms_by_family")); print(r.get("launches_by_family"))
print("tb", r.get("transformer_blocks", {}).get("frac_of_mfma_peak"), "code_agreement", d.get("code_agreement"))
print("train", d.get("train_step", {}).get("ms_per_step"), "lstm", {k: v.get("ms_per_step") for k, v in d.get("lstm_models", {}).items() if isinstance(v, dict)}, "config1", {k: v.get("ms") for k, v in d.get("config1", {}).items() if isinstance(v, dict) and "ms" in v})
PY
