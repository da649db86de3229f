#!/bin/bash
# round-5 GPU call 19: the dispatch-order tile walk for projections whose weights exceed the L2s (kv_all of the cross-attention layers) —
# kernel tests, then the 64-clip step A/B (shipped rule vs XCD-aware runs everywhere), then bench.py
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c19; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "gemm_h2 or abi" 2>&1 | grep -v amdgpu.ids | tail -n 8) > $O/pytest_kernels.txt; tail -n 4 $O/pytest_kernels.txt
(timeout 300 python tools/bench_step_variants.py 131072,4325376 3 2>&1 | grep -v amdgpu.ids) > $O/tile_order_step_ab.txt; cat $O/tile_order_step_ab.txt
(timeout 400 python bench.py 2> $O/bench.err | tail -n 1) > $O/bench.json; python - <<'PY'
import json
d = json.loads(open("/root/repo/gpurun_out/r05_c19/bench.json").read())
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"])
PY
