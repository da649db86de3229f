#!/bin/bash
# round-5 GPU call 6: the LSTM layer with PIPELINED hand-over phases (staging overlapped with the MFMA phase, K slice by
# K slice): us per time step at DisCo's and CaMN's sizes against the shipped kernel, bit-identity, timing ablations
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c6; mkdir -p $O
cd $R
(timeout 500 python tools/bench_lstm.py --steps 3 --no-cpu --layer-only 2>&1 | grep -v amdgpu.ids | grep "^{") > $O/lstm_layer.jsonl
python - <<'PY'
import json, os
for ln in open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r05_c6/lstm_layer.jsonl")):
    d = json.loads(ln)
    print(d.get("workload", d.get("model")), "ms/step", round(d.get("ms_per_step", 0), 2))
    print("  ", json.dumps(d.get("lstm_layer_us_per_step")))
    print("  ", json.dumps(d.get("same_bits_as_shipped")))
PY
