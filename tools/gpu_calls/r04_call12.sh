#!/bin/bash
# round-4 GPU call 12: the tape frees gradient buffers / closures as the backward walks — memory report, the training GPU tests, step time
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c12; mkdir -p $O
cd $R
(timeout 180 python tools/train_memory_report.py 2>&1 | grep -v amdgpu.ids) > $O/mem_share.json
python - <<PY
import json
d = json.load(open("$O/mem_share.json"))
for m in d["marks"]:
    print("%-48s live %7.3f  peak %7.3f" % (m["at"], m["live_gb"], m["peak_since_last_gb"]))
PY
(timeout 600 python -m pytest tests/test_train_forward_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4) > $O/pytest_train.txt
(timeout 300 python tools/bench_train_step.py 2>&1 | grep -v amdgpu.ids | tail -1) > $O/train_step.json
cat $O/pytest_train.txt; cut -c1-400 $O/train_step.json
