#!/bin/bash
# round-5 GPU call 7: the whole GPU suite with durations, smoke, and the driver-style bench line on the round's tree (first full verification;
# the profiles of the final tree come from r05_call8.sh)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c7; mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 2>&1 | grep -v amdgpu.ids | tail -n 45) > $O/pytest_gpu.txt; echo "pytest done"; tail -n 8 $O/pytest_gpu.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 3) > $O/smoke.txt; cat $O/smoke.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -n 4 $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r05_c7/bench.json")).read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d.get("roofline", {}).get("frac"), "pcie", d.get("pcie_inclusive", {}).get("ms_per_step"))
ts = d.get("train_step", {})
print("train_step", ts.get("ms_per_step"), ts.get("eager"), "exchange", json.dumps(ts.get("exchange"))[:700])
print("lstm", {k: (v.get("ms_per_step"), v.get("roofline", {}).get("us_per_time_step")) for k, v in d.get("lstm_models", {}).items() if isinstance(v, dict)})
print("config1", {k: v.get("ms") for k, v in d.get("config1", {}).items() if isinstance(v, dict) and "ms" in v}, "sweep", {k: round(v.get("ms_per_step", 0), 2) for k, v in d.get("batch_sweep", {}).get("by_batch", {}).items()})
PY
