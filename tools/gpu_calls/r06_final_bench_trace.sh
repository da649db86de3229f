#!/bin/bash
# round-6: the driver-style bench line and the rocprofv3 kernel stats of the serialized step ONLY (after profiles/hbm_traffic.json was refreshed by r06_final.sh: the line then
# carries traffic_is_stale = false); the two averages (roofline.avg_launch_us live, rocprof's per-kernel average) must agree
set -u
R=$GRAFT_REPO_ROOT
TAG=${1:-r06_final_bt}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
BQ="--no-cpu-baseline --no-other-configs --also ''"
eval timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-concurrent $BQ > $O/trace.log 2>&1; echo "trace rc=$?"
eval timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace2 -o bench --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-concurrent $BQ > $O/trace2.log 2>&1; echo "trace2 rc=$?"
cd $R
TAG=$TAG python - <<'PY'
import csv, glob, json, os
R = os.environ["GRAFT_REPO_ROOT"]; O = os.path.join(R, "gpurun_out", os.environ["TAG"])
d = json.loads(open(os.path.join(O, "bench.json")).read().strip().splitlines()[-1])
r = d["roofline"]
print("ms_per_step", d["ms_per_step"], "frac", r["frac"], "avg_launch_us", r["avg_launch_us"], "serialized", r["serialized_kernel_ms"], "stale", r.get("traffic_is_stale"))
for t in ("trace", "trace2"):
    for f in glob.glob(os.path.join(O, t, "**", "*kernel_trace.csv"), recursive=True):
        os.remove(f)
    for f in glob.glob(os.path.join(O, t, "**", "*kernel_stats.csv"), recursive=True):
        rows = [x for x in csv.DictReader(open(f)) if "gemm_h2" in x["Name"] or "gemm_pipe_kernel<float" in x["Name"]]      # not the bf16 8192^3 calibration GEMM of the roofline object
        c = sum(int(x["Calls"]) for x in rows); ns = sum(float(x["TotalDurationNs"]) for x in rows)
        print(t, "rocprof gemm launches", c, "avg us", ns / c / 1e3)
        for l in open(os.path.join(O, t + ".log")):
            if l.startswith("{"):
                dd = json.loads(l); print(t, "live in that run: avg_launch_us", dd["roofline"]["avg_launch_us"], "ms_per_step", dd["ms_per_step"])
PY
