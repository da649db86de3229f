#!/bin/bash
# round-2 GPU call 5: register-pipelined K-loop (all tile configs, three precisions), LSTM models at the BASELINE sizes,
# tile sweeps to compare with c2, bench, and the names behind the copyBuffer launches of the EMAGE step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c5; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or attention" > $O/pytest_gemm.txt 2>&1; echo "gemm tests rc=$?" | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -s > $O/pytest_lstm.txt 2>&1; echo "lstm tests rc=$?" | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "batch64 or clip_fp32 or bf16_agreement" > $O/pytest_parity.txt 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt
timeout 300 python tools/bench_gemm.py --dtype f16x3 --configs 25,18,27,32,33,34,36,37,38,40,42 2>&1 | grep -v amdgpu.ids > $O/gemm_sweep_f16x3.txt
timeout 300 python tools/bench_gemm.py --dtype bf16 --configs 25,18,27,32,33,34,36,37,38,40,42 2>&1 | grep -v amdgpu.ids > $O/gemm_sweep_bf16.txt
timeout 200 python tools/bench_gemm.py --dtype f16x3 --ablate --configs 25,32 2>&1 | grep -v amdgpu.ids > $O/gemm_ablate_f16x3.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout 600 python tools/bench_lstm.py --steps 3 --models camn > $O/bench_lstm.json 2> $O/bench_lstm.err; echo "bench_lstm rc=$?" | tee -a $O/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/trace" -o bench --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --also "" --no-concurrent > "$GRAFT_REPO_ROOT/$O/trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/c5/trace/*kernel_trace.csv")
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n)[:80]
starts = [i for i, r in enumerate(rows) if "wav_conv_in" in r["Kernel_Name"]]
seg = rows[starts[-1]:]
with open("gpurun_out/c5/last_step_kernels.txt", "w") as out:
    prev_end = None
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        out.write(f"{short(r['Kernel_Name']):82s} {(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}\n")
        prev_end = e
print(collections.Counter(short(r["Kernel_Name"]) for r in seg).most_common(14))
PY
find $O/trace -name "*kernel_trace*" -delete
tail -3 $O/pytest_gemm.txt $O/pytest_lstm.txt $O/pytest_parity.txt; cut -c1-300 $O/bench.json; cat $O/bench_lstm.json
