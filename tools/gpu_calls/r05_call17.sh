#!/bin/bash
# round-5 GPU call 17: grids of at most one 64 x 64 tile per CU on 8 waves (config 189: 2 LDS-DMA instructions per wave and K-tile instead of 4):
# kernel test (fp32-grade, same bits as 120), A/B of one clip / 8 clips / the 64-clip batch
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c17; mkdir -p $O
cd $R
(timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "two_ktiles or (every_tile_configuration and (189 or 188))" 2>&1 | tail -n 4) > $O/pytest_kernels.txt; tail -n 2 $O/pytest_kernels.txt
(timeout 900 python tools/bench_small_grids.py 2>&1 | grep "^{") | tee $O/small_grids_ab.jsonl
