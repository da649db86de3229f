#!/bin/bash
# final: the whole GPU suite + smoke on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c36; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
tail -n 5 $O/pytest_gpu.txt; tail -n 1 $O/smoke.txt
