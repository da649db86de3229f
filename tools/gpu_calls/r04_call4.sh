#!/bin/bash
# round-4 GPU call 4: EMAGE_H2W third version (loads drained before the loop's register homes are taken, epilogue loads ahead of the stores)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c4; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "h2w or gemm_h2" > $O/pytest_h2w.txt 2>&1; echo "pytest h2w rc=$?" | tee $O/summary.txt
timeout 300 python tools/bench_gemm_h2.py --configs 100,120,160,161,162,163 --shapes "qkv 768->2304 +vt,out_proj,ffn1,ffn2,kv_part,ragged,mlp 256->768,head 768->256" > $O/sweep.txt 2>&1; echo "sweep rc=$?" | tee -a $O/summary.txt
timeout 120 python tools/trace_gemm_h2.py 260 > $O/trace.txt 2>&1; echo "trace rc=$?" | tee -a $O/summary.txt
B="timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --also ''"
for v in "--w-direct 0" "--w-direct 1"; do
  eval $B $v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d.get('roofline', {})
print('AB [%s] ms_per_step %.3f serialized %.3f gemm_launches %s gemm_ms %s' % ('$v', d['ms_per_step'], r.get('serialized_kernel_ms', -1), r.get('launches_per_step'), r.get('kernel_time_ms_by_family', {}).get('emage_gemm')))" >> $O/ab.txt 2>&1
done
tail -8 $O/pytest_h2w.txt; cat $O/sweep.txt; cat $O/ab.txt; head -4 $O/trace.txt | cut -c1-500
