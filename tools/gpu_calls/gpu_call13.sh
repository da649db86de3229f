#!/bin/bash
# round-2 GPU call 13: paired LSTM step (both directions per launch), stream-priority experiment, bench with the roofline tied
# to the single-stream graph's wall time
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c13; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -s > $O/pytest_lstm.txt 2>&1; echo "lstm tests rc=$?" | tee -a $O/summary.txt
timeout 400 python tools/bench_lstm.py --steps 3 > $O/bench_lstm.json 2> $O/bench_lstm.err; echo "bench_lstm rc=$?" | tee -a $O/summary.txt
for tag in "--main-priority" ""; do
  name=$(echo "x$tag" | tr -d ' -')
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --also "" $tag > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench [$tag] rc=$?" | tee -a $O/summary.txt
done
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -n 3 $O/pytest_lstm.txt; cat $O/bench_lstm.json | cut -c1-200
for f in $O/bench_x*.json; do echo $f; grep -o '"ms_per_step": [0-9.]*' $f | head -1; done
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c13/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["ms_per_step"], {k: r[k] for k in ("achieved", "frac", "avg_launch_us", "serialized_kernel_ms", "event_bracket_sum_ms")})
PY
