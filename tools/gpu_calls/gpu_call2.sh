#!/bin/bash
# round-2 GPU call 2: L2 -> LDS stream micro-benchmark, split-f16 / bf16 tile sweeps, kernel-stats trace of the f16x3 step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c2; mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/micro/l2_stream 2048 200 > $O/l2_stream_2m.txt 2>&1
timeout 120 tools/micro/l2_stream 512 200 > $O/l2_stream_512k.txt 2>&1
timeout 300 python tools/bench_gemm.py --dtype f16x3 --configs 25,18,27,32,33,34,36,37,38,39,40,41,42 2>&1 | grep -v amdgpu.ids > $O/gemm_sweep_f16x3.txt
timeout 300 python tools/bench_gemm.py --dtype bf16 --configs 25,32,33,34,36,37,38,39,40,41,42 2>&1 | grep -v amdgpu.ids > $O/gemm_sweep_bf16.txt
timeout 200 python tools/bench_gemm.py --dtype f16x3 --ablate --configs 25,32 2>&1 | grep -v amdgpu.ids > $O/gemm_ablate_f16x3.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace" -o bench --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --precision f16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find $O/trace -name "*kernel_trace*" -delete
head -30 $O/l2_stream_2m.txt; cat $O/gemm_sweep_f16x3.txt
