#!/bin/bash
# GPU box: rocprofv3 kernel trace + PMC passes of the headline bench; summaries land in gpurun_out/prof_<tag>/.
# usage: [PROFILE_LAST=N] tools/profile.sh <tag> [bench args...]     (PROFILE_LAST: summarise only the last N dispatches of each kernel,
#        e.g. 200 = the three timed windows + the back-to-back replay of `--steps 50`, without the warm-up that led to the window)
set -u
TAG=${1:-r1}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 50 --no-cpu-baseline --no-roofline --no-steady --no-configs $*"  # (default --warmup 20; pass --warmup 300 for the steady window)
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES -d $OUT/pmc_mfma -o pmc -- $BENCH > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
cd - >/dev/null
python tools/summarize_profile.py $OUT $OUT/summary.json > /dev/null 2>&1
rm -rf $OUT/trace $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write  # keep only the small summary (gpurun_out is capped at 64 MiB)
ls -la $OUT
