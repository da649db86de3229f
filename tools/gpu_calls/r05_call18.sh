#!/bin/bash
# round-5 GPU call 18: lone-block grids (ONE clip: every contraction; the N = 256 launches of larger batches) on the 64 x 64 configurations that
# overlap INSIDE a wave — register-pipelined fragments (121), DMA issue interleaved with the MFMAs (151), two K-tiles per barrier (183) — against the
# shipped tile (120): at 64 clips co-resident blocks hide a block's serial phases, a lone block has nobody to hide them
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c18; mkdir -p $O
cd $R
(timeout 900 python tools/bench_small_grids.py 121,151,183,0 2>&1 | grep "^{") | tee $O/small_grids_ab.jsonl
