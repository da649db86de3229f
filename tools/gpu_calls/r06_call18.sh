#!/bin/bash
# round 6, call 18: a Linear's input gradient added to its input's collected gradient by the contraction's epilogue (TrainForward.fuse_grad_adds): training GPU tests, interleaved A/B
O=gpurun_out/r06_call18; mkdir -p $O
timeout 900 python -m pytest tests/test_train_forward_gpu.py -x -q > $O/pytest_train.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_train.txt
for i in 1 2; do for v in 0 1; do
  timeout 300 python tools/bench_train_step.py --quick --fuse-grad-adds $v > $O/train_fga${v}_$i.json 2> $O/train_fga${v}_$i.err; echo "fga=$v run $i rc=$?"
  python -c "import json; d = json.loads(open('$O/train_fga${v}_$i.json').read().strip().splitlines()[-1]); print('fuse_grad_adds=$v', round(d['ms_per_step'], 2), 'ms', round(d['peak_memory_gb'], 2), 'GB', 'loss', d['loss_all_after_replays'])"
done; done
