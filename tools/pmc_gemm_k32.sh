#!/bin/bash
# PMC passes over the bf16 fc.0 GEMM, 64-k K-tiles / one workgroup per CU against 32-k K-tiles / two per CU (tools/ab_bf16_k32.py)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=${1:-8192}
for K32 in 0 1; do
  OUT=gpurun_out/r5_gemm_k32_$K32; mkdir -p $OUT
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA" \
             "TCC_HIT TCC_MISS TCC_REQ TCC_READ" "TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -o p -- python tools/time_gemm_bf16.py $K32 $N > $OUT/pmc$i.log 2>&1
  done
done
python - <<'PY'
import csv, glob, collections
for k32 in (0, 1):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r5_gemm_k32_{k32}/pmc*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fc_gemm_phased_kernel<true, true, 2, 2" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("bf16_k32 =", k32)
    for c, v in sorted(agg.items()):
        print(f"   {c:32s} {sum(v) / len(v):14.5g}  (n={len(v)})")
PY
