#!/bin/bash
# round-3 GPU call 17: the final tree — smoke, the driver-style bench line, the parity / kernel / training tests the attention change touches
set -u
R=$GRAFT_REPO_ROOT
cd "$R" || exit 1
O=$R/gpurun_out/r03h; mkdir -p $O
(timeout 150 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4) > $O/smoke.txt
(timeout 300 python bench.py --steps 20 --warmup 3) > $O/bench.json 2> $O/bench.err
(timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -q -p no:cacheprovider 2>&1 | tail -6) > $O/pytest_kernels_parity.txt
(timeout 200 python -m pytest tests/test_train_forward_gpu.py -q -p no:cacheprovider -k "forward_matches_oracle or training_step_matches or mul_add" 2>&1 | tail -4) > $O/pytest_train_subset.txt
cat $O/smoke.txt; cut -c1-300 $O/bench.json; echo; cat $O/pytest_kernels_parity.txt $O/pytest_train_subset.txt
