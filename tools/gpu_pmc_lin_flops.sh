#!/bin/bash
# fp64 VALU instruction counts of the rigid-body linearisation (values pre-pass + tangent walk) at the bench size, for the
# flop-based roof bench.py reports: flops = 64 lanes x (ADD_F64 + MUL_F64 + 2 FMA_F64) wave-instructions, every lane slot
# counted (masked-off lanes included: an upper bound of the useful work, the denominator of "instruction-bound").
# Writes gpurun_out/summary/<tag>_linearize_flops.json, stamped with the kernel-source hash bench.py checks.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${1:-r05}
BATCH=${2:-1024}
mkdir -p $OUT/summary
export TMPDIR=/tmp
cd /tmp
D="python $R/tools/linearize_bench.py $BATCH"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/prof_linf -o linf -- $D > $OUT/prof_linf.log 2>&1
cd $R
python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "$R")
from bench import kernel_source_hash
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/prof_linf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "linearize_contact_dynamics_kernel" in k or "rbd_values_kernel" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
kern = {}
tot = collections.defaultdict(float)
for k, c in acc.items():
    # every dispatch of the tool is the same launch; with / without the multiplier terms alternate: keep the plain mean
    kern[k] = {n: sum(v) / len(v) for n, v in c.items()}
    kern[k]["dispatches"] = len(next(iter(c.values())))
    for n, v in c.items():
        tot[n] += sum(v) / len(v)
points = $BATCH * 46
flops = 64.0 * (tot["SQ_INSTS_VALU_ADD_F64"] + tot["SQ_INSTS_VALU_MUL_F64"] + 2.0 * tot["SQ_INSTS_VALU_FMA_F64"])
out = {"grid_points": points, "batch": $BATCH, "per_launch": dict(tot), "kernels": kern, "flops_per_launch": flops,
       "flops_per_grid_point": flops / points if points else None,
       "definition": "64 x (SQ_INSTS_VALU_ADD_F64 + SQ_INSTS_VALU_MUL_F64 + 2 SQ_INSTS_VALU_FMA_F64), summed over rbd_values_kernel and "
                     "linearize_contact_dynamics_kernel of one rtoc_linearize_contact_dynamics call (tools/linearize_bench.py)",
       "_kernel_source_hash": kernel_source_hash()}
json.dump(out, open("$OUT/summary/${TAG}_linearize_flops.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("grid_points", "flops_per_launch", "flops_per_grid_point")}))
PY
tail -2 $OUT/prof_linf.log
rm -rf $OUT/prof_linf
