#!/bin/bash
# Collect the round's evidence on the MI355X box (run from the repo root through gpurun); outputs under gpurun_out/r01/
set -u
R=$PWD
O=$R/gpurun_out/r01
mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15) > $O/pytest_gpu.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3) > $O/smoke.txt
(timeout 500 python bench.py --steps 20 --warmup 3) > $O/bench.json 2> $O/bench.err
(timeout 200 python tools/bench_gemm.py --ablate --configs 25,20) 2>&1 | grep -v amdgpu.ids > $O/gemm_ablation.txt
(timeout 200 python tools/bench_gemm.py --configs 0,10,11,13,19,20,25,26) 2>&1 | grep -v amdgpu.ids > $O/gemm_sweep.txt
(timeout 100 python tools/bf16_agreement.py) 2>&1 | grep -v amdgpu.ids > $O/bf16_agreement.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o bench --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $O/pmc_write.log 2>&1
cd $R
# keep the merged payload small: the full traces can be tens of MB
rm -f $O/trace/bench_kernel_trace.csv
python - <<'PY'
import csv, collections, json, os, re
O = os.path.join(os.getcwd(), "gpurun_out", "r01")
def short(n):
    m = re.search(r'(gemm_pipe_kernel<[^>]*>|gemm_kernel<[^>]*>|attn_kernel<[^>]*>|\w+_kernel|vq_argmin\w*|argmax\w*|gather_rows)', n)
    return m.group(1) if m else n[:60]
out = {}
for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = os.path.join(O, tag, "bench_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            a = agg[short(r["Kernel_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    out[ctr] = {k: {"launches": c, "kb_per_launch": v / c} for k, (c, v) in agg.items() if "gemm" in k or "attn" in k or "layernorm" in k}
    os.remove(f)
    t = os.path.join(O, tag, "bench_kernel_trace.csv")
    if os.path.exists(t):
        os.remove(t)
json.dump(out, open(os.path.join(O, "pmc_summary.json"), "w"), indent=1)
PY
ls -R $O | head -40
tail -4 $O/pytest_gpu.txt; cat $O/smoke.txt; cut -c1-400 $O/bench.json; tail -2 $O/bench.err
