#!/bin/bash
# round-6 GPU call 3: WHAT bounds the K-loop — timing ablations (emage_set_tuning key 1: 1 = no operand DMA, 2 = no fragment reads, 4 = no MFMAs) of the
# shipped 64 x 64 tile (120), the lone 8-wave 64 x 192 tile (100) and the antiphase tiles (300 / 342) on out_proj and ffn2; LDS bank conflicts by PMC
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_c3; mkdir -p $O
cd $R
for dbg in 0 1 2 4 3 5 6 7; do
  echo "== dbg $dbg" >> $O/ablations.txt
  timeout 200 python tools/bench_gemm_h2.py --configs 120,100,300,342 --shapes "out_proj,ffn2" --dbg $dbg 2>&1 | grep -v amdgpu.ids >> $O/ablations.txt
done
cat $O/ablations.txt
cd /tmp && export TMPDIR=/tmp
for cfg in 120 300 342; do
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $O/pmc_$cfg -o pmc --output-format csv -- python $R/tools/bench_gemm_h2.py --configs $cfg --shapes "out_proj" --loop 4 > $O/pmc_$cfg.log 2>&1; echo "pmc $cfg rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r06_c3")
for cfg in (120, 300, 342):
    for f in glob.glob(os.path.join(O, f"pmc_{cfg}", "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "gemm_h2" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            print(cfg, k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
