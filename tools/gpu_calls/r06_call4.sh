#!/bin/bash
# round-6 GPU call 4: the LayerNorm fold — kernel test, the parity suite on the folded default, interleaved step A/B
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_c4; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "fold or grouped or gemm_h2" 2>&1 | tail -15 | tee $O/pytest_kernels.txt
BQ="--no-cpu-baseline --no-other-configs --also '' --no-roofline"
for arm in fold nofold fold nofold fold nofold; do
  flag=""; [ $arm = nofold ] && flag="--no-fold-ln"
  eval timeout 200 python bench.py --steps 20 --warmup 5 $flag $BQ 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('arm $arm ms_per_step', round(d['ms_per_step'],3))" | tee -a $O/step_ab_fold.txt
done
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -15 | tee $O/pytest_parity.txt
