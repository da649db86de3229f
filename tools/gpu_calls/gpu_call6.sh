#!/bin/bash
# round-2 GPU call 6: the whole GPU suite through the torch.library ops / nn.Module classes, bench with the x3 tile heuristics
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c6; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/pytest_gpu.txt; tail -n 2 $O/smoke.txt; cut -c1-400 $O/bench.json
