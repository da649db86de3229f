#!/bin/bash
O=gpurun_out/c18; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --also "" > $O/bench_x.json 2> $O/bench_x.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout 600 python tools/bench_no_layernorm.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --also "" > $O/bench_noln.json 2> $O/bench_noln.err; echo "bench noln rc=$?" | tee -a $O/summary.txt
grep -o '"ms_per_step": [0-9.]*' $O/bench_x.json $O/bench_noln.json
tail -3 $O/bench_noln.err
