#!/bin/bash
# round-4 GPU call 13: the leaner training step (queued multi-tensor gradient adds, emage_grad_prep, device weight packing, Philox-key dropout
# masks, per-step weight images, train-only operand set) — kernel tests, the whole training GPU suite, step time + peak memory; and the A/B of the
# face / body lock-step walk in inference (one box).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c13; mkdir -p $O
cd $R
(timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6) > $O/pytest_kernels.txt
(timeout 200 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k "grouped_launches" 2>&1 | tail -6) > $O/pytest_parity_grouped.txt
(timeout 200 python tools/bench_train_step.py 2>&1 | grep -v amdgpu.ids | tail -1) > $O/train_step.json
(timeout 500 python -m pytest tests/test_train_forward_gpu.py -m gpu -q -x -p no:cacheprovider --durations=6 2>&1 | tail -14) > $O/pytest_train.txt
BQ="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --also ''"
for v in 0 1 0 1; do
  eval timeout 200 python bench.py $BQ --group-face-body $v > $O/bench_fb$v.json 2> $O/bench_fb$v.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_fb$v.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("group_face_body $v: ms_per_step %.3f serialized %.3f gemm %.3f launches %d avg_us %.2f frac %.4f" % (d["ms_per_step"], r["serialized_kernel_ms"], r["kernel_time_ms_by_family"]["emage_gemm"], r["launches_per_step"], r["avg_launch_us"], r["frac"]))
except Exception as e:
    print("group_face_body $v: failed", e)
PY
done | tee $O/ab_face_body.txt
(timeout 150 python tools/train_memory_report.py 2>&1 | grep -v amdgpu.ids) > $O/mem.json
tail -3 $O/pytest_kernels.txt $O/pytest_parity_grouped.txt; tail -12 $O/pytest_train.txt; cut -c1-900 $O/train_step.json; cat $O/ab_face_body.txt
python - <<PY
import json
try:
    d = json.load(open("$O/mem.json"))
    for m in d["marks"]:
        print("%-48s live %7.3f  peak %7.3f" % (m["at"], m["live_gb"], m["peak_since_last_gb"]))
except Exception as e:
    print("memory report failed", e)
PY
