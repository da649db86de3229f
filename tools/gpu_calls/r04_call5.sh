#!/bin/bash
# round-4 GPU call 5: epilogue loads batched ahead of the stores (h2_tile_epilogue) — kernel tests, then A/B in ONE library (tools build:
# emage_set_tuning key 1 bit 5 = the chained form; bit 6 is unused and only selects the tools library for the other arm)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c5; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "gemm" > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee $O/summary.txt
B="timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --also ''"
for v in "--gemm-dbg 64" "--gemm-dbg 32" "--gemm-dbg 64" "--gemm-dbg 32" ""; do
  eval $B $v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d.get('roofline', {})
print('AB [%s] ms_per_step %.3f serialized %.3f gemm_launches %s gemm_ms %s' % ('$v', d['ms_per_step'], r.get('serialized_kernel_ms', -1), r.get('launches_per_step'), r.get('kernel_time_ms_by_family', {}).get('emage_gemm')))" >> $O/ab.txt 2>&1
done
timeout 200 python tools/bench_gemm_h2.py --configs 100,120,119 --shapes "qkv 768->2304 +vt,out_proj,ffn1,ffn2,kv_all,head 768->256,conv3 256->256 +res" > $O/sweep.txt 2>&1
tail -4 $O/pytest.txt; cat $O/ab.txt; cat $O/sweep.txt
