#!/bin/bash
# round 6, call 20: interleaved A/B of the captured training step with the forward's stride-1 WavEncoder convolutions as implicit GEMMs (0) / through emage_conv_slab (1)
O=gpurun_out/r06_call20; mkdir -p $O
for i in 1 2 3; do for v in 0 1; do
  timeout 300 python tools/bench_train_step.py --quick --slab-convs $v > $O/train_slab${v}_$i.json 2> $O/train_slab${v}_$i.err; echo "slab=$v run $i rc=$?"
  python -c "import json; d = json.loads(open('$O/train_slab${v}_$i.json').read().strip().splitlines()[-1]); print('slab_convs=$v', round(d['ms_per_step'], 2), 'ms', 'loss', d['loss_all_after_replays'])"
done; done
