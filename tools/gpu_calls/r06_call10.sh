#!/bin/bash
# round 6, call 10: the per-model activation shift (EMAGE_H2_SHIFT): kernel test, range cases on the rescaled twin, inference parity files, the shift study
O=gpurun_out/r06_call10; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "activation_shift or layernorm_fold or test_gemm" > $O/pytest_kernels.txt 2>&1; echo "kernels rc=$?"; tail -4 $O/pytest_kernels.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -s -k "range or rescaled" > $O/pytest_range.txt 2>&1; echo "range rc=$?"; grep "range case" $O/pytest_range.txt; tail -4 $O/pytest_range.txt
timeout 600 python tools/bench_activation_shift.py > $O/shift_study.json 2> $O/shift_study.err; echo "study rc=$?"; cat $O/shift_study.json | head -c 3000
