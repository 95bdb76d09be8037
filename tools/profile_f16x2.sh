#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes for the bench step in the DCE_FP32_F16X2 precision.
# Usage: tools/profile_f16x2.sh <tag> [extra bench args]      outputs under gpurun_out/<tag>_*
set -u
TAG=${1:-r5q}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
python -c "from deep_contact_estimator_amd import build; print(build.built_hash())" > $OUT/${TAG}_source_hash.txt
B="python bench.py --precision fp32_f16x2 --steps 20 --warmup 3 --settle-s 0.2 --no-cpu-baseline --no-extras --no-kernel-timing $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o ${TAG} -- python bench.py --precision fp32_f16x2 --steps 300 --warmup 50 --no-cpu-baseline --no-extras --no-kernel-timing $* > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o ${TAG} -- $B > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o ${TAG} -- $B > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_sq -o ${TAG} -- $B > $OUT/${TAG}_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_lds -o ${TAG} -- $B > $OUT/${TAG}_pmc_lds.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_REQ TCC_READ --output-format csv -d $OUT/${TAG}_pmc_tcc -o ${TAG} -- $B > $OUT/${TAG}_pmc_tcc.log 2>&1
{
  echo "# $TAG: bench step in DCE_FP32_F16X2 ($*), source hash $(cat $OUT/${TAG}_source_hash.txt)"
  echo "## kernel-trace --stats (300 steps)"; f=$(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-70s calls %6s avg %9.1f us  %6s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:6]))
PY
  echo "## counters (mean per launch)"
  for d in fetch write sq lds tcc; do f=$(find $OUT/${TAG}_pmc_$d -name "*counter_collection.csv" | head -1); echo "-- $d"; python tools/pmc_summary.py "$f"; done
} > $OUT/${TAG}_summary.txt 2>&1
cat $OUT/${TAG}_summary.txt | cut -c1-260
