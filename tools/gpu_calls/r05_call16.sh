#!/bin/bash
# round-5 GPU call 16: ONE clip (BASELINE configs[0]) under rocprofv3 — per-kernel stats of the B = 1 step (serialized graph), what it is bound by
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
eval timeout 300 rocprofv3 --kernel-trace --stats -d $O/b1 -o b1 --output-format csv -- python $R/bench.py --batch 1 --steps 20 --warmup 3 --no-concurrent --no-cpu-baseline --no-other-configs --also "''" > $O/b1.json 2> $O/b1.err; echo "rc=$?"
find $O/b1 -name "*kernel_trace.csv" -delete
cd $R
python - <<'PY'
import csv, glob, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05_c16")
d = json.loads(open(os.path.join(O, "b1.json")).read().strip().splitlines()[-1])
r = d["roofline"]
print("B=1 step (single stream)", d["ms_per_step"], "ms; serialized kernel ms", r["serialized_kernel_ms"], "launches", sum(r["launches_by_family"].values()))
print("by family", r["kernel_time_ms_by_family"], "gemm launches", r["launches_per_step"], "avg us", r["avg_launch_us"])
f = glob.glob(os.path.join(O, "b1", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for x in rows[:14]:
    print("   %-95s %6s %8.2f ms %7.1f us" % (x["Name"][:95], x["Calls"], float(x["TotalDurationNs"]) / 1e6, float(x["AverageNs"]) / 1e3))
PY
