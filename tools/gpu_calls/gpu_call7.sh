#!/bin/bash
# round-2 GPU call 7: LDS-resident-slab convolutions + fused WavEncoder block 0 (bitwise vs the unfused path), result-store
# policy experiment, end-to-end A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c7; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "slab or block0 or gemm" > $O/pytest_slab.txt 2>&1; echo "slab tests rc=$?" | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_lstm_gpu.py -x -q -k "batch64 or clip_fp32 or golden or bf16_agreement or graph" > $O/pytest_parity.txt 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt
for tag in "" "--no-slab-convs" "--gemm-dbg 8" "--gemm-dbg 16"; do
  name=$(echo "x$tag" | tr -d ' -')
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline $tag > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench [$tag] rc=$?" | tee -a $O/summary.txt
done
timeout 200 python tools/bench_gemm.py --dtype f16x3 --configs 25,33,34,36 --dbg 8 2>&1 | grep -v amdgpu.ids > $O/gemm_sweep_f16x3_sc1.txt
timeout 200 python tools/bench_gemm.py --dtype f16x3 --configs 25,33,34,36 2>&1 | grep -v amdgpu.ids > $O/gemm_sweep_f16x3_plain.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err
tail -n 3 $O/pytest_slab.txt $O/pytest_parity.txt
for f in $O/bench_x*.json; do echo $f; cut -c1-260 $f | grep -o '"ms_per_step": [0-9.]*\|"other_precisions".*'; done
