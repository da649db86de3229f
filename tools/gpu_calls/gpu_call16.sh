#!/bin/bash
# persistent LSTM layer v3 (gx prefetch, two-half staging): kernel tests, breakdown, kernel traces of the two LSTM models
O=gpurun_out/c16; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_lstm_gpu.py -x -q -s -k "lstm_layer" > $O/pytest_layer.txt 2>&1; echo "layer tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_layer.txt
timeout 600 python tools/bench_lstm.py --steps 3 --layer-only --no-cpu > $O/bench_lstm.json 2> $O/bench_lstm.err; echo "bench_lstm rc=$?" | tee -a $O/summary.txt
cat $O/bench_lstm.json; tail -3 $O/bench_lstm.err
for m in camn disco; do
  timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace_$m" -o bench --output-format csv -- python "$GRAFT_REPO_ROOT/tools/bench_lstm.py" --models $m --steps 2 --no-cpu > $O/trace_$m.log 2>&1; echo "trace $m rc=$?" | tee -a $O/summary.txt
  find $O/trace_$m -name "*kernel_trace.csv" -delete
done
ls -R $O | head -30
