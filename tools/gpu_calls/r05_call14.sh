#!/bin/bash
# round-5 GPU call 14: per-kernel stats of the CaMN and DisCo forwards on the final tree (where their 72 / 8 ms go)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c14; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in camn disco; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$m -o $m --output-format csv -- python $R/tools/bench_lstm.py --steps 3 --no-cpu --models $m > $O/$m.log 2>&1; echo "$m rc=$?"
  find $O/$m -name "*kernel_trace.csv" -delete
done
cd $R
python - <<'PY'
import csv, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05_c14")
for m in ("camn", "disco"):
    f = glob.glob(os.path.join(O, m, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(m, "total kernel ms", round(tot / 1e6, 1))
    for r in rows[:12]:
        print("   %-100s %6s %8.1f ms %8.1f us" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
