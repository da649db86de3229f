#!/bin/bash
# round-4 GPU call 1: full GPU suite on the round's first tree (training health, grouped GEMM, H2 residual default), the driver-style bench line,
# A/B of the two new defaults against the round-3 forms on ONE box, and the vendor-GEMM calibration (tools only)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c1; mkdir -p $O
cd $R
timeout 1100 python -m pytest tests -m gpu -q --maxfail=12 --durations=12 -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee $O/summary.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
B="timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --also ''"
for v in "" "--no-group-gemms" "--f32-residual" "--no-group-gemms --f32-residual"; do
  eval $B $v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d.get('roofline', {})
print('AB [%s] ms_per_step %.3f serialized %.3f gemm_launches %s gemm_ms %s ln_ms %s' % ('$v', d['ms_per_step'], r.get('serialized_kernel_ms', -1), r.get('launches_per_step'), r.get('kernel_time_ms_by_family', {}).get('emage_gemm'), r.get('kernel_time_ms_by_family', {}).get('emage_layernorm')))" >> $O/ab.txt 2>&1
done
timeout 300 python tools/bench_gemm_vs_hipblaslt.py > $O/gemm_vs_hipblaslt.txt 2> $O/gemm_vs_hipblaslt.err; echo "calib rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.txt; cat $O/ab.txt; cat $O/gemm_vs_hipblaslt.txt
