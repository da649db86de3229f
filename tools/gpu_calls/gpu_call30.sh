#!/bin/bash
# evidence set on the final tree: whole GPU suite, smoke, bench (roofline + CPU baseline), LSTM models, training step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c30; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout 400 python tools/bench_lstm.py --steps 3 > $O/bench_lstm.json 2> $O/bench_lstm.err; echo "bench_lstm rc=$?" | tee -a $O/summary.txt
timeout 600 python tools/bench_train_forward.py --steps 2 --batch 56 --full-step --precision fp32 > $O/bench_train_step_fp32.json 2> $O/bench_train_step_fp32.err; echo "bench train step fp32 rc=$?" | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_train_forward_gpu.py -q -s > $O/pytest_train.txt 2>&1; echo "train tests rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/pytest_gpu.txt; tail -n 1 $O/smoke.txt; cut -c1-300 $O/bench.json; cut -c1-250 $O/bench_lstm.json; cat $O/bench_train_step_fp32.json; grep -h "training step on\|backward on device\|passed" $O/pytest_train.txt
