#!/bin/bash
# round-5 GPU call 3: (a) emage_gemm_ws kernel test (fixed); (b) sweeps: fat tiles (256 x 256 / 256 x 192 / 256 x 128, one block per CU) on the
# wide K / V projection, the 64 x 64 tile with its residual fetched ahead of the K-loop, the backward shapes through the two-pass split-K
# (workspace) incl. the K / V projections' gradients; (c) the driver-style bench line with config 1 and the batch sweep
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c3; mkdir -p $O
cd $R
(timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "gemm_ws" 2>&1 | tail -n 12) > $O/pytest_kernels.txt; tail -n 3 $O/pytest_kernels.txt
(timeout 300 python tools/bench_gemm_h2.py --configs 170,119,171,172,173,174,175,100 --shapes "kv_all,qkv 768->2304 +vt,ffn1" 2>&1 | grep -v amdgpu.ids) > $O/sweep_fat_tiles.txt; cat $O/sweep_fat_tiles.txt
(timeout 300 python tools/bench_gemm_h2.py --configs 120,187,100,128 --shapes "out_proj,ffn2,ragged,conv3 256->256 +res" 2>&1 | grep -v amdgpu.ids) > $O/sweep_residual_prefetch.txt; cat $O/sweep_residual_prefetch.txt
(timeout 400 python tools/bench_gemm_h2.py --configs 120,124,100,113,170,119,172,174 --shapes "bwd" --workspace-mb 256 2>&1 | grep -v amdgpu.ids) > $O/sweep_backward_two_pass.txt; cat $O/sweep_backward_two_pass.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -n 5 $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r05_c3/bench.json")).read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d.get("roofline", {}).get("frac"))
print("config1", json.dumps(d.get("config1"))[:900])
print("batch_sweep", json.dumps(d.get("batch_sweep"))[:900])
ts = d.get("train_step", {})
print("train_step", ts.get("ms_per_step"), ts.get("eager"), ts.get("roofline", {}).get("frac"))
print("lstm", {k: v.get("ms_per_step") for k, v in d.get("lstm_models", {}).items() if isinstance(v, dict)})
PY
