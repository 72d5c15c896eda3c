#!/bin/bash
# usage (on the GPU box): scripts/pmc_t16.sh TAG — HBM traffic and SQ counters of the 16-token-tile kernel (Q4_K, 11008 x 4096,
# batch 8 / 16 / 32), separate --pmc passes as MI355X_MICROARCH.md prescribes.  Summary -> gpurun_out/pmc_TAG_t16.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
cd /tmp && export TMPDIR=/tmp
for B in 8 16 32; do
  for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
    D=$R/gpurun_out/pmc_${TAG}_t16_b${B}_$(echo $C | tr ' ' '_' | cut -c1-24)
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 $R/scripts/run_kernel.py t16 12 $B 8 > $D.log 2>&1 || echo "$C batch $B failed"
  done
done
python3 - <<PY
import csv, glob, collections
R = "$R"; TAG = "$TAG"
out = []
for B in (8, 16, 32):
    vals = collections.defaultdict(list)
    for f in glob.glob(R + "/gpurun_out/pmc_%s_t16_b%d_*/**/*counter_collection.csv" % (TAG, B), recursive=True):
        for r in csv.DictReader(open(f)):
            if "mmq_t16_kernel" in r["Kernel_Name"]: vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in vals.items()}
    alg = 11008 * 4096 // 256 * 144 + B * 4096 * 2 + B * 11008 * 2
    line = "batch %2d: " % B + "  ".join("%s %.4g" % (k, m[k]) for k in sorted(m))
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        hbm = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
        line += "  | HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB units) %.0f = %.2f x algorithmic %d" % (hbm, hbm / alg, alg)
    if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m: line += "  | SQ_WAIT_ANY / SQ_WAVE_CYCLES %.2f" % (m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"])
    out.append(line); print(line)
open(R + "/gpurun_out/pmc_%s_t16.txt" % TAG, "w").write("\n".join(out) + "\n")
PY
