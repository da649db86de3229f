#!/bin/bash
# round-6 GPU call 2: antiphase tiles v2 — where the DMA pieces are issued (MODE 1: in front of the fragment reads; MODE 2: between the MFMAs of
# the C phase), sweep + phase traces
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_c2; mkdir -p $O
cd $R
timeout 400 python tools/bench_gemm_h2.py --configs 120,100,300,340,341,342,343,344 --shapes "out_proj,ffn2,mlp 256,fc 512" > $O/pp_sweep_narrow.txt 2>&1; echo "narrow rc=$?"
timeout 400 python tools/bench_gemm_h2.py --configs 100,170,311,345,346,344 --shapes "qkv 768,ffn1,kv_part" > $O/pp_sweep_wide.txt 2>&1; echo "wide rc=$?"
timeout 120 python tools/trace_gemm_h2.py 350,351,352 > $O/pp_phase_trace.txt 2>&1; echo "trace rc=$?"
tail -n 6 $O/pp_sweep_narrow.txt $O/pp_sweep_wide.txt
grep -A2 "wave  0\|wave  4" $O/pp_phase_trace.txt | cut -c1-420
