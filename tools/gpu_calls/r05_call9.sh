#!/bin/bash
# round-5 GPU call 9: the ring-of-8 tile for grids of at most one tile per CU (config 188): kernel tests (fp32-grade, same bits as 120), the clip
# goldens (B = 1 / 2 / 64), A/B of one clip / 8 clips / the 64-clip batch with and without it
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c9; mkdir -p $O
cd $R
(timeout 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "two_ktiles or (every_tile_configuration and (188 or 120)) or gemm_h2 or grouped" 2>&1 | tail -n 6) > $O/pytest_kernels.txt; tail -n 3 $O/pytest_kernels.txt
(timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "clip_fp32_matches_reference or batch64 or grouped_launches or lockstep" 2>&1 | tail -n 6) > $O/pytest_parity.txt; tail -n 3 $O/pytest_parity.txt
(timeout 600 python tools/bench_small_grids.py 2>&1 | grep "^{") | tee $O/small_grids_ab.jsonl
