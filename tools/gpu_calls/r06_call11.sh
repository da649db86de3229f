#!/bin/bash
# round 6, call 11: activation shift incl. the attention kernel's own Q / K / V split: kernel tests (NaN-strict _cmp), range cases, the shift study, the whole parity file
O=gpurun_out/r06_call11; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q > $O/pytest_kernels.txt 2>&1; echo "kernels rc=$?"; tail -4 $O/pytest_kernels.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -k "range or rescaled" > $O/pytest_range.txt 2>&1; echo "range rc=$?"; grep "range case" $O/pytest_range.txt; tail -4 $O/pytest_range.txt
timeout 600 python tools/bench_activation_shift.py > $O/shift_study.json 2> $O/shift_study.err; echo "study rc=$?"; cat $O/shift_study.json | head -c 3000
