#!/bin/bash
# round-2 GPU call 4: DisCo / CaMN kernels + models on the device, their throughput, and where the copyBuffer launches of the EMAGE step come from
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c4; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -s > $O/pytest_lstm.txt 2>&1; echo "lstm tests rc=$?" | tee -a $O/summary.txt
timeout 600 python tools/bench_lstm.py --steps 3 > $O/bench_lstm.json 2> $O/bench_lstm.err; echo "bench_lstm rc=$?" | tee -a $O/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o bench --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --also "" --no-concurrent > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/c4/trace/*kernel_trace.csv")
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last wav_conv_in marks the start of the last step
starts = [i for i, n in enumerate(names) if "wav_conv_in" in n]
seg = rows[starts[-1]:]
cnt = collections.Counter()
with open("gpurun_out/c4/last_step_kernels.txt", "w") as out:
    for r in seg:
        short = r["Kernel_Name"].split("(")[0][-70:]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        out.write(f"{short:72s} {dur:8.1f} us grid {r.get('Grid_Size_X','?')} wg {r.get('Workgroup_Size_X','?')}\n")
        cnt[short] += 1
print(cnt.most_common(12))
PY
find $O/trace -name "*kernel_trace*" -delete
tail -5 $O/pytest_lstm.txt; cat $O/bench_lstm.json
