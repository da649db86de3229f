#!/bin/bash
# round-2 GPU call 8: slab kernels after the fill / two-K-tiles-per-barrier rework (bitwise tests, per-kernel A/B, end to end),
# deeper rings for small-grid GEMMs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c8; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "slab or block0" > $O/pytest_slab.txt 2>&1; echo "slab tests rc=$?" | tee -a $O/summary.txt
timeout 300 python tools/bench_slab.py 2>&1 | grep -v amdgpu.ids > $O/bench_slab.txt; echo "bench_slab rc=$?" | tee -a $O/summary.txt
timeout 300 python tools/bench_gemm.py --dtype f16x3 --configs 25,44,43,33,32 2>&1 | grep -v amdgpu.ids > $O/gemm_sweep_f16x3_rings.txt
timeout 300 python tools/bench_gemm.py --dtype bf16 --configs 25,44,43,33,32 2>&1 | grep -v amdgpu.ids > $O/gemm_sweep_bf16_rings.txt
for tag in "" "--no-slab-convs"; do
  name=$(echo "x$tag" | tr -d ' -')
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline $tag > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench [$tag] rc=$?" | tee -a $O/summary.txt
done
tail -n 3 $O/pytest_slab.txt; cat $O/bench_slab.txt; cat $O/gemm_sweep_f16x3_rings.txt
for f in $O/bench_x*.json; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f; done
