#!/bin/bash
# round-3 GPU call 16: split-f16 attention with K / V^T staged in LDS — bitwise test against the register path, the attention kernel
# tests, the graph parity tests, then the step A/B (tools library: --attn-variant 1 = register path) in one call
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
(timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider -k "attention" 2>&1 | tail -8) > $O/r03_t16.txt
if grep -q "failed\|error" $O/r03_t16.txt; then cat $O/r03_t16.txt; exit 0; fi
(timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -p no:cacheprovider -k "batch64 or tail_windows" 2>&1 | tail -5) >> $O/r03_t16.txt
for v in 0 1 0; do
  echo "== --attn-variant $v" >> $O/r03_ab16.txt
  (timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --also "" --attn-variant $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['serialized_kernel_ms'], d['roofline']['kernel_time_ms_by_family'].get('emage_attention'))") >> $O/r03_ab16.txt 2>&1
done
cat $O/r03_t16.txt; cat $O/r03_ab16.txt
