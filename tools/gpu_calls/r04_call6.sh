#!/bin/bash
# round-4 GPU call 6: the round's evidence on one box — full GPU suite, smoke, the driver's bench line, rocprofv3 kernel stats of the serialized
# step (agrees with roofline.avg_launch_us), HBM-side bytes per launch (separate FETCH_SIZE / WRITE_SIZE passes), kernel stats of the training step
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c6; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 2>&1 | tail -40) > $O/pytest_gpu.txt; echo "pytest done" | tee $O/summary.txt
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3) > $O/smoke.txt
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
BQ="--no-cpu-baseline --no-other-configs --also ''"
eval timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-concurrent $BQ > $O/trace.log 2>&1; echo "trace rc=$?" | tee -a $O/summary.txt
eval timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-roofline --no-graph $BQ > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?" | tee -a $O/summary.txt
eval timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-roofline --no-graph $BQ > $O/pmc_write.log 2>&1; echo "write rc=$?" | tee -a $O/summary.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/train -o train --output-format csv -- python $R/tools/bench_train_step.py > $O/train.log 2>&1; echo "train rc=$?" | tee -a $O/summary.txt
cd $R
python - <<'PY'
import csv, collections, glob, json, os, shutil
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r04_c6")
def fam(n):
    for key, tag in (("gemm_pipe_kernel", "emage_gemm"), ("gemm_h2_kernel", "emage_gemm"), ("gemm_h2_group_kernel", "emage_gemm"), ("conv_slab_kernel", "emage_conv_slab"),
                     ("wav_block0", "emage_conv_slab"), ("attn_", "emage_attention"), ("layernorm_kernel", "emage_layernorm")):
        if key in n:
            return tag
    return None
out = {}
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    fs = glob.glob(os.path.join(O, tag, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        k = fam(r["Kernel_Name"])
        if k and r["Counter_Name"] == ctr:
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out[ctr] = {k: {"launches": c, "kb_per_launch": v / c} for k, (c, v) in agg.items()}
    shutil.rmtree(os.path.join(O, tag))
json.dump(out, open(os.path.join(O, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
for d in ("trace", "train"):                       # keep the stats, drop the per-dispatch traces (tens of MB)
    for f in glob.glob(os.path.join(O, d, "**", "*kernel_trace.csv"), recursive=True):
        os.remove(f)
PY
tail -6 $O/pytest_gpu.txt; cat $O/smoke.txt; cut -c1-300 $O/bench.json; find $O -name "*stats*" | head
