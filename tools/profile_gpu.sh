#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes for the bench workload.
# Usage: tools/profile_gpu.sh <tag>      outputs under gpurun_out/<tag>_*
set -u
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
python -c "from deep_contact_estimator_amd import build; print(build.built_hash())" > $OUT/${TAG}_source_hash.txt
B="python bench.py --steps 20 --warmup 3 --settle-s 0.2 --no-cpu-baseline --no-extras --no-kernel-timing"
# long enough for the clocks to settle: the averages then agree with bench.py's HIP-event times to <1 %
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o ${TAG} -- python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o ${TAG} -- $B > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o ${TAG} -- $B > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/${TAG}_pmc_sq -o ${TAG} -- $B > $OUT/${TAG}_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_lds -o ${TAG} -- $B > $OUT/${TAG}_pmc_lds.log 2>&1
# L2 (TCC) view of the same step: hit rate, and the read requests it sends on to the fabric (Infinity Fabric -> MALL -> HBM) by size
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_REQ TCC_READ --output-format csv -d $OUT/${TAG}_pmc_tcc1 -o ${TAG} -- $B > $OUT/${TAG}_pmc_tcc1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_DRAM --output-format csv -d $OUT/${TAG}_pmc_tcc2 -o ${TAG} -- $B > $OUT/${TAG}_pmc_tcc2.log 2>&1
# bf16-FC mode (BASELINE configs[4]): what binds its GEMMs -- matrix pipe, LDS, L2
BB="$B --precision bf16_fc"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_bf16sq -o ${TAG} -- $BB > $OUT/${TAG}_pmc_bf16sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_bf16lds -o ${TAG} -- $BB > $OUT/${TAG}_pmc_bf16lds.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_REQ TCC_READ --output-format csv -d $OUT/${TAG}_pmc_bf16tcc -o ${TAG} -- $BB > $OUT/${TAG}_pmc_bf16tcc.log 2>&1
# ... and what they (and the mode's conv stack) move through HBM
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_bf16fetch -o ${TAG} -- $BB > $OUT/${TAG}_pmc_bf16fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_bf16write -o ${TAG} -- $BB > $OUT/${TAG}_pmc_bf16write.log 2>&1
find $OUT -name "${TAG}*" -type f | head -40
for f in $OUT/${TAG}_*.log; do echo "== $f"; grep -E '"metric"|rror' $f | cut -c1-300; done
