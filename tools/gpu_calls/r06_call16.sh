#!/bin/bash
# round 6, call 16: stride-1 convolutions' input gradient as one implicit-GEMM convolution of dY (TrainForward.direct_conv_dx): training GPU tests, interleaved A/B of the captured step
O=gpurun_out/r06_call16; mkdir -p $O
timeout 900 python -m pytest tests/test_train_forward_gpu.py -x -q > $O/pytest_train.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_train.txt
for i in 1 2; do for v in 0 1; do
  timeout 300 python tools/bench_train_step.py --quick --direct-conv-dx $v > $O/train_dcx${v}_$i.json 2> $O/train_dcx${v}_$i.err; echo "dcx=$v run $i rc=$?"
  python -c "import json; d = json.loads(open('$O/train_dcx${v}_$i.json').read().strip().splitlines()[-1]); print('direct_conv_dx=$v', round(d['ms_per_step'], 2), 'ms', round(d['peak_memory_gb'], 2), 'GB', 'loss', d['loss_all_after_replays'])"
done; done
