#!/bin/bash
# round-3 GPU call 10: evidence on the current tree — whole GPU suite, smoke, the driver-style bench line, per-kernel stats of the
# serialized step and of the training step, PMC passes (own runs, --kernel-trace only) over the H2 and the round-2 GEMM kernels.
set -u
R=$GRAFT_REPO_ROOT
cd "$R" || exit 1
O=$R/gpurun_out/r03f; mkdir -p $O
(timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12) > $O/pytest_gpu.txt
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4) > $O/smoke.txt
(timeout 400 python bench.py --steps 20 --warmup 3) > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_ser -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs --also "" --no-concurrent > $O/trace_ser.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_train -o train --output-format csv -- python $R/tools/bench_train_forward.py --full-step --graph --steps 3 > $O/trace_train.log 2>&1
for mode in h2 f16x3; do
  for set in "MfmaUtil VALUBusy" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    tag=${mode}_$(echo "$set" | cut -d' ' -f1)
    timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_$tag -o pmc --output-format csv -- python $R/tools/prof_gemm_pmc.py $mode > $O/pmc_$tag.log 2>&1
    echo "pmc $mode [$set] rc=$?" >> $O/summary.txt
  done
done
cd $R
python - <<'PY'
import csv, glob, collections, re, json
out = {}
for f in sorted(glob.glob("gpurun_out/r03f/pmc_*/*counter_collection.csv")):
    mode = f.split("/pmc_")[1].split("_")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:80]
        if "gemm" in n:
            agg[mode + ": " + n + " grid" + r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
json.dump(out, open("gpurun_out/r03f/pmc_gemm_summary.json", "w"), indent=1)
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace*" -delete; find $O -name "*agent_info*" -delete
tail -4 $O/pytest_gpu.txt; cat $O/smoke.txt; cut -c1-400 $O/bench.json; echo; cat $O/summary.txt; tail -2 $O/trace_train.log
