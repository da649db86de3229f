#!/bin/bash
# round-3 GPU call 15: HBM-side bytes per launch of the f16x3 step with the EMAGE_H2 kernels (roofline.traffic): FETCH_SIZE and WRITE_SIZE in separate PMC passes
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_c15; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --no-other-configs --also "" > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?" | tee -a $O/summary.txt
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --no-other-configs --also "" > $O/pmc_write.log 2>&1; echo "write rc=$?" | tee -a $O/summary.txt
cd $R
python - <<'PY'
import csv, collections, json, os, re
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03_c15")
def fam(n):
    for key, tag in (("gemm_pipe_kernel", "emage_gemm"), ("gemm_h2_kernel", "emage_gemm"), ("conv_slab_kernel", "emage_conv_slab"), ("attn_kernel", "emage_attention"), ("layernorm_kernel", "emage_layernorm")):
        if key in n:
            return tag
    return None
out = {}
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = os.path.join(O, tag, "bench_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = fam(r["Kernel_Name"])
        if k and r["Counter_Name"] == ctr:
            agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out[ctr] = {k: {"launches": c, "kb_per_launch": v / c} for k, (c, v) in agg.items()}
    os.remove(f)
    t = os.path.join(O, tag, "bench_kernel_trace.csv")
    if os.path.exists(t):
        os.remove(t)
json.dump(out, open(os.path.join(O, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
