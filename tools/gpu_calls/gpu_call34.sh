#!/bin/bash
O=gpurun_out/c34; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 700 python -m pytest tests/test_train_forward_gpu.py -x -q -s -k "captured" > $O/pytest_capture.txt 2>&1; echo "capture test rc=$?" | tee -a $O/summary.txt
tail -25 $O/pytest_capture.txt | cut -c1-300
timeout 500 python tools/bench_train_forward.py --steps 3 --batch 56 --full-step --graph --precision fp32 > $O/bench_train_step_graph.json 2> $O/bench_train_step_graph.err; echo "bench graph rc=$?" | tee -a $O/summary.txt
cat $O/bench_train_step_graph.json; tail -5 $O/bench_train_step_graph.err | cut -c1-300
