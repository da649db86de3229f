#!/bin/bash
# round-5 GPU call 2: (a) the captured step WITH its RCCL collectives (tests/test_rccl_gpu.py, second try); (b) emage_gemm_ws — the two-pass
# split-K weight gradients: kernel test, the reference-step tests at 2 and 56 clips (bit-equal parameters across two captures), A/B of the
# captured training step against the fp32-atomic form; (c) "BK = 64": two K-tiles per ring slot and barrier (KPB = 2) — bit-identity with
# the shipped configurations and a sweep over the window shapes
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c2; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_rccl_gpu.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -n 30) > $O/pytest_rccl.txt; echo "rccl done"
(timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "gemm_ws or two_ktiles or (every_tile_configuration and (180 or 181 or 183 or 184 or 185 or 186 or 100 or 120 or 170))" 2>&1 | tail -n 15) > $O/pytest_kernels.txt; echo "kernels done"; tail -n 3 $O/pytest_kernels.txt
(timeout 300 python tools/bench_gemm_h2.py --configs 100,180,170,181,120,183,186,113,184,185 --shapes "qkv 768->2304 +vt,out_proj,ffn1,ffn2,kv_all,head,mlp 256,fc 512,conv3 256->256 +res,bwd dX 768<-768" 2>&1 | grep -v amdgpu.ids) > $O/sweep_kpb.txt; cat $O/sweep_kpb.txt
for v in 32768 16384 8192 32768 16384; do
  (timeout 120 python tools/bench_train_step.py --quick --h2-variant $v 2>&1 | grep -v amdgpu.ids | tail -n 1) > $O/arm_$v.json
  python - <<PY
import json
try:
    d = json.loads(open("$O/arm_$v.json").read().strip().splitlines()[-1])
    print("h2_variant $v: ms_per_step %.2f peak %.2f GB loss %.6f" % (d["ms_per_step"], d["peak_memory_gb"], d["loss_all_after_replays"]))
except Exception as e:
    print("arm $v failed", e)
PY
done | tee $O/train_ab.txt
(timeout 900 python -m pytest tests/test_train_forward_gpu.py -m gpu -q -s -p no:cacheprovider -k "baseline_batch_step or test_training_step_matches_the_reference or captured_f16x3_step" 2>&1 | grep -v amdgpu.ids | tail -n 25) > $O/pytest_train.txt
tail -n 12 $O/pytest_train.txt; tail -n 25 $O/pytest_rccl.txt
