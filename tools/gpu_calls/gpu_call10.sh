#!/bin/bash
# round-2 GPU call 10: two K-tiles per ring slot (configs 45-48) — tests, sweep, end-to-end A/B; slab swizzle fix re-measured
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c10; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or slab or block0" > $O/pytest_gemm.txt 2>&1; echo "gemm tests rc=$?" | tee -a $O/summary.txt
timeout 300 python tools/bench_gemm.py --dtype f16x3 --configs 25,47,33,45,34,46,36,48 2>&1 | grep -v amdgpu.ids > $O/gemm_sweep_f16x3_kps.txt
timeout 300 python tools/bench_slab.py 2>&1 | grep -v amdgpu.ids > $O/bench_slab.txt
for v in 1 0; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --also "" --gemm-variant $v > $O/bench_variant$v.json 2> $O/bench_variant$v.err; echo "bench variant $v rc=$?" | tee -a $O/summary.txt
done
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "batch64 or clip_fp32" > $O/pytest_parity.txt 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt
tail -n 3 $O/pytest_gemm.txt $O/pytest_parity.txt; cat $O/gemm_sweep_f16x3_kps.txt | cut -c1-110; head -3 $O/bench_slab.txt
for f in $O/bench_variant*.json; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; done
