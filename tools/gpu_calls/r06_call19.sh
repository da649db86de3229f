#!/bin/bash
# round 6, call 19: the training forward's stride-1 WavEncoder convolutions through emage_conv_slab (same bits as the implicit GEMM): training GPU tests, three runs of the captured step
O=gpurun_out/r06_call19; mkdir -p $O
timeout 900 python -m pytest tests/test_train_forward_gpu.py -x -q > $O/pytest_train.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_train.txt
for i in 1 2 3; do
  timeout 300 python tools/bench_train_step.py --quick > $O/train_$i.json 2> $O/train_$i.err; echo "run $i rc=$?"
  python -c "import json; d = json.loads(open('$O/train_$i.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 2), 'ms', round(d['peak_memory_gb'], 2), 'GB', 'loss', d['loss_all_after_replays'])"
done
