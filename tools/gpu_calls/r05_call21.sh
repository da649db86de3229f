#!/bin/bash
# round-5 GPU call 21 (FINAL tree, after the dispatch-order walk for kv_all shipped; the same collection as call 8): rocprofv3 kernel stats of the serialized inference step (agrees with roofline.avg_launch_us) and of the captured
# training step, memory-side bytes per launch behind roofline.traffic (separate FETCH_SIZE / WRITE_SIZE passes), MfmaUtil of the EMAGE_H2 GEMMs at
# the window shapes (north_star: "MFMA utilisation for the transformer blocks"), the driver-style bench line
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c21; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
BQ="--no-cpu-baseline --no-other-configs --also ''"
eval timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-concurrent $BQ > $O/trace.log 2>&1; echo "trace rc=$?"
eval timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-roofline --no-graph $BQ > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
eval timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-roofline --no-graph $BQ > $O/pmc_write.log 2>&1; echo "write rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/train -o train --output-format csv -- python $R/tools/bench_train_step.py --quick > $O/train.log 2>&1; echo "train rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc MfmaUtil VALUBusy -d $O/pmc_mfma -o pmc --output-format csv -- python $R/tools/prof_gemm_pmc.py h2 > $O/pmc_mfma.log 2>&1; echo "mfma rc=$?"
cd $R
python - <<'PY'
import csv, collections, glob, json, os, re, shutil
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05_c21")
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
json.dump(out, open(os.path.join(O, "pmc_fetch_write_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
mf = {}
for f in glob.glob(os.path.join(O, "pmc_mfma", "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:80]
        if "gemm" in n:
            agg[n + " grid" + r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        mf.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
json.dump(mf, open(os.path.join(O, "pmc_mfma_util_gemm_h2.json"), "w"), indent=1)
print(json.dumps(mf, indent=1)[:1500])
shutil.rmtree(os.path.join(O, "pmc_mfma"), ignore_errors=True)
for d in ("trace", "train"):
    for f in glob.glob(os.path.join(O, d, "**", "*kernel_trace.csv"), recursive=True):
        os.remove(f)
d = json.loads(open(os.path.join(O, "bench.json")).read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d.get("roofline", {}).get("frac"), "avg_launch_us", d.get("roofline", {}).get("avg_launch_us"))
PY
find $O -name "*stats*" | head
