#!/bin/bash
# round 6, call 8: the training forward's Linear contractions on EMAGE_H2 operands (TrainForward.h2_forward): training GPU tests + interleaved A/B of the captured step
O=gpurun_out/r06_call8; mkdir -p $O
timeout 900 python -m pytest tests/test_train_forward_gpu.py -x -q > $O/pytest_train.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_train.txt
for i in 1 2; do for v in 0 1; do
  timeout 300 python tools/bench_train_step.py --quick --h2-forward $v > $O/train_h2f${v}_$i.json 2> $O/train_h2f${v}_$i.err; echo "h2f=$v run $i rc=$?"
  python - <<PY
import json
d = json.loads(open("$O/train_h2f${v}_$i.json").read().strip().splitlines()[-1])
print("h2_forward=$v", round(d["ms_per_step"], 2), "ms", round(d["peak_memory_gb"], 2), "GB", "loss", d["loss_all_after_replays"])
PY
done; done
