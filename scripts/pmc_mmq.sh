#!/bin/bash
# usage (on the GPU box): [WHAT=x64] scripts/pmc_mmq.sh TAG [type] [batch]   (env N, K pass through; WHAT = the run_kernel.py mode, default mmq)
# three SQ counter passes (8 counters each, never combined with other trace domains) over scripts/run_kernel.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; T=${2:-12}; B=${3:-128}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_WAVE32_LDS"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/scripts/run_kernel.py ${WHAT:-mmq} $T $B 6 > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
for i in (1,2,3):
    fs = glob.glob("$R/gpurun_out/pmc_${TAG}_%d/**/*counter_collection.csv" % i, recursive=True)
    acc = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if "mmq" in r["Kernel_Name"]: acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%-62s %-28s n=%d mean %.4g" % (k, c, len(v), sum(v) / len(v)))
PY
