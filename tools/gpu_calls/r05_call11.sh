#!/bin/bash
# round-5 GPU call 11: the LSTM layer with 32 clips per block and TWO blocks per CU for batches up to 256 clips (CaMN):
# the LSTM GPU suite (bit-identity with the per-step kernels, goldens, lost-block report), us per time step with / without, model step times
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c11; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_lstm_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 8) > $O/pytest_lstm.txt; tail -n 4 $O/pytest_lstm.txt
(timeout 500 python tools/bench_lstm.py --steps 3 --no-cpu --layer-only 2>&1 | grep -v amdgpu.ids | grep "^{") > $O/lstm_layer.jsonl
python - <<'PY'
import json, os
for ln in open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r05_c11/lstm_layer.jsonl")):
    d = json.loads(ln)
    print(d.get("workload", d.get("model")), "ms/step", round(d.get("ms_per_step", 0), 2))
    print("  ", json.dumps(d.get("lstm_layer_us_per_step")))
    print("  ", json.dumps(d.get("same_bits_as_shipped")))
PY
