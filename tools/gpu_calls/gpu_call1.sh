#!/bin/bash
# round-2 GPU call 1: split-f16 GEMM kernel tests, f16x3 parity, first bench lines of the three precisions
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > gpurun_out/c1/pytest_gemm.txt 2>&1
echo "gemm tests rc=$?" | tee -a gpurun_out/c1/summary.txt
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -s -k "f16x3 or batch64 or vq_model_api or bf16_agreement" > gpurun_out/c1/pytest_parity.txt 2>&1
echo "parity tests rc=$?" | tee -a gpurun_out/c1/summary.txt
for p in f16x3 bf16 fp32; do
  timeout 300 python bench.py --precision $p --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c1/bench_$p.json 2> gpurun_out/c1/bench_$p.err
  echo "bench $p rc=$?" | tee -a gpurun_out/c1/summary.txt
done
timeout 300 python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --fused-layers > gpurun_out/c1/bench_bf16_fused.json 2> gpurun_out/c1/bench_bf16_fused.err
echo "bench fused rc=$?" | tee -a gpurun_out/c1/summary.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/c1/prof_x3" -o x3 -- python "$GRAFT_REPO_ROOT/bench.py" --precision f16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/c1/prof_x3.log" 2>&1
echo "rocprof rc=$?" | tee -a "$GRAFT_REPO_ROOT/gpurun_out/c1/summary.txt"
cd "$GRAFT_REPO_ROOT"
find gpurun_out/c1/prof_x3 -name "*.db" -size +20M -delete 2>/dev/null
find gpurun_out/c1/prof_x3 -name "*kernel_trace*" -delete 2>/dev/null
tail -3 gpurun_out/c1/pytest_gemm.txt gpurun_out/c1/pytest_parity.txt
cat gpurun_out/c1/bench_*.json | cut -c1-400
