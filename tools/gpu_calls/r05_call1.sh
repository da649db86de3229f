#!/bin/bash
# round-5 GPU call 1: (a) the training exchange through RCCL on one GPU, eager and captured (tests/test_rccl_gpu.py); (b) FETCH_SIZE / WRITE_SIZE
# calibrated on launches with known byte counts and on the EMAGE_H2 GEMM shapes of the step with three tile orders (tools/prof_traffic_calib.py);
# (c) the wide K/V projection with an XCD's run walking M first, timed; (d) this box's inference step (quick bench line)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c1; mkdir -p $O
cd $R
(timeout 700 python -m pytest tests/test_rccl_gpu.py -m gpu -q -x -s -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -n 40) > $O/pytest_rccl.txt; echo "rccl done"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o calib --output-format csv -- python $R/tools/prof_traffic_calib.py --plan $O/plan.json > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o calib --output-format csv -- python $R/tools/prof_traffic_calib.py > $O/pmc_write.log 2>&1; echo "write rc=$?"
cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -n 1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -n 1)
python tools/prof_traffic_calib.py --summarize "$F" "$W" $O/plan.json --out $O/traffic_calibration.json 2>&1 | tee $O/traffic_calibration.txt
rm -rf $O/pmc_fetch $O/pmc_write
(timeout 200 python tools/bench_gemm_h2.py --shapes kv_all,qkv,out_proj --configs 170,100,120 2>&1 | grep -v amdgpu.ids) > $O/sweep_default_order.txt
(timeout 200 python tools/bench_gemm_h2.py --shapes kv_all,qkv,out_proj --configs 170,100,120 --dbg 64 2>&1 | grep -v amdgpu.ids) > $O/sweep_m_first.txt
(timeout 200 python tools/bench_gemm_h2.py --shapes kv_all,qkv,out_proj --configs 170,100,120 --dbg 32 2>&1 | grep -v amdgpu.ids) > $O/sweep_no_remap.txt
tail -n 4 $O/sweep_default_order.txt $O/sweep_m_first.txt $O/sweep_no_remap.txt
(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --also '' 2> $O/bench.err) > $O/bench_quick.json; cut -c1-600 $O/bench_quick.json
tail -n 30 $O/pytest_rccl.txt
