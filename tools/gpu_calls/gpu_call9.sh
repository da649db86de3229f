#!/bin/bash
# round-2 GPU call 9: PMC passes over the dominant GEMM shapes (MFMA busy, VALU busy, LDS conflicts, wait cycles)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c9; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail > "$GRAFT_REPO_ROOT/$O/list_avail.txt" 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "MfmaUtil VALUBusy" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  tag=$(echo "$set" | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d "$GRAFT_REPO_ROOT/$O/pmc_$tag" -o pmc --output-format csv -- python "$GRAFT_REPO_ROOT/tools/prof_gemm_pmc.py" f16x3 > "$GRAFT_REPO_ROOT/$O/pmc_$tag.log" 2>&1
  echo "pmc [$set] rc=$?" | tee -a "$GRAFT_REPO_ROOT/$O/summary.txt"
done
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, collections, re, json
out = {}
for f in glob.glob("gpurun_out/c9/pmc_*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
        if "gemm" in n or "slab" in n:
            agg[n + " grid" + r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
json.dump(out, open("gpurun_out/c9/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
find $O -name "*counter_collection.csv" -size +2M -delete; find $O -name "*kernel_trace*" -delete
grep -i "mfma\|lds_bank\|VALUBusy" $O/list_avail.txt | head -30
