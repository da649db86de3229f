#!/bin/bash
# round-6 FINAL-tree collection (the same set as round 5's call 21): the driver-style bench line; rocprofv3 kernel stats of the serialized inference step
# (must agree with roofline.avg_launch_us) and of the captured training step; memory-side bytes per launch behind roofline.traffic (separate FETCH_SIZE /
# WRITE_SIZE passes) -> profiles/hbm_traffic.json WITH the tree it was collected on (_commit, _kernel_sources_sha, _gpu_call); MfmaUtil of the H2 GEMMs
set -u
R=$GRAFT_REPO_ROOT
TAG=${1:-r06_final}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
BQ="--no-cpu-baseline --no-other-configs --also ''"
eval timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-concurrent $BQ > $O/trace.log 2>&1; echo "trace rc=$?"
eval timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-roofline --no-graph $BQ > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
eval timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-roofline --no-graph $BQ > $O/pmc_write.log 2>&1; echo "write rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/train -o train --output-format csv -- python $R/tools/bench_train_step.py --quick > $O/train.log 2>&1; echo "train rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc MfmaUtil VALUBusy -d $O/pmc_mfma -o pmc --output-format csv -- python $R/tools/prof_gemm_pmc.py h2 > $O/pmc_mfma.log 2>&1; echo "mfma rc=$?"
cd $R
TAG=$TAG python - <<'PY'
import csv, collections, glob, json, os, re, shutil, sys
R = os.environ["GRAFT_REPO_ROOT"]
TAG = os.environ["TAG"]
O = os.path.join(R, "gpurun_out", TAG)
sys.path.insert(0, R)
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
# bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md, calibrated in round 5) -> the table bench.py reads, stamped with the tree
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    import bench
    tj = {"gemm_bf16": 24005639.314285718}
    for k in out["FETCH_SIZE"]:
        if k in out["WRITE_SIZE"]:
            tj[f"{k}:f16x3"] = (2 * out["FETCH_SIZE"][k]["kb_per_launch"] + out["WRITE_SIZE"][k]["kb_per_launch"]) * 1024
    commit = open(os.path.join(R, "profiles", "_collect_stamp.txt")).read().strip() if os.path.exists(os.path.join(R, "profiles", "_collect_stamp.txt")) else None
    tj["_commit"], tj["_kernel_sources_sha"], tj["_gpu_call"] = commit, bench.kernel_sources_sha(), TAG
    tj["_source"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py --steps 2 --warmup 1 --no-graph (tools/gpu_calls/r06_final.sh), "
                     "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per the gfx950 note of MI355X_MICROARCH.md; the factors were calibrated on this code's own access patterns in round 5 "
                     "(profiles/r05_traffic_calibration.txt: LayerNorm 2 x FETCH / bytes 1.00, WRITE / bytes 1.00)")
    tj["_note"] = ("bytes per launch, averaged over every launch of the kernel family in the f16x3 step.  Memory-side (fabric) request counters: Infinity-Cache hits are counted.  "
                   "The GEMM family's excess over its algorithmic bytes is the eight XCD L2s each fetching every W panel (DESIGN.md 4.1)")
    json.dump(tj, open(os.path.join(O, "hbm_traffic.json"), "w"), indent=1)
    print({k: v for k, v in tj.items() if not k.startswith("_")})
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
r = d.get("roofline", {})
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", r.get("frac"), "avg_launch_us", r.get("avg_launch_us"), "serialized", r.get("serialized_kernel_ms"))
print(r.get("kernel_time_ms_by_family"))
print("tb", r.get("transformer_blocks", {}).get("frac_of_mfma_peak"), "bf16", {k: (v if not isinstance(v, dict) else v.get("frac")) for k, v in d.get("other_precisions", {}).get("bf16", {}).items() if k in ("ms_per_step", "roofline")})
print("train", d.get("train_step", {}).get("ms_per_step"), "lstm", {k: v.get("ms_per_step") for k, v in d.get("lstm_models", {}).items() if isinstance(v, dict)}, "config1", {k: v.get("ms") for k, v in d.get("config1", {}).items() if isinstance(v, dict) and "ms" in v})
PY
find $O -name "*stats*" | head
