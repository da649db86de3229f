#!/bin/bash
# round-5 GPU call 13: the LSTM models' input projections on pre-split EMAGE_H2 operands: the LSTM GPU suite (goldens, oracle parity), A/B of the
# DisCo / CaMN forwards with and without
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c13; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_lstm_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 8) > $O/pytest_lstm.txt; tail -n 4 $O/pytest_lstm.txt
for arm in "" "--no-h2-proj" "" "--no-h2-proj"; do
  (timeout 300 python tools/bench_lstm.py --steps 3 $arm 2>&1 | grep -v amdgpu.ids | grep "^{") > $O/arm.jsonl
  python - "$arm" <<'PY'
import json, os, sys
for ln in open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r05_c13/arm.jsonl")):
    d = json.loads(ln)
    print("%-14s %-8s ms/step %.2f  input projections: %s  max err vs oracle %.2e" % (sys.argv[1] or "(shipped)", d.get("model", "?"), d["ms_per_step"], d.get("input_projections"), d.get("max_err_vs_oracle_clip0_1", float("nan"))))
PY
done | tee $O/lstm_ab.txt
