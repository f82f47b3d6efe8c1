#!/bin/bash
# rocprofv3 evidence for one round: kernel-trace stats + PMC passes (separate runs, as the guide prescribes).
# usage (on the GPU box, from the repo root): bash tools/profile.sh <tag> [bench args]
TAG=${1:-r02}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
BENCH="python $ROOTDIR/bench.py --steps 40 --warmup 5 --cpu-iters 0 --no-roofline-pass --pmc 0 $@"
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $ROOTDIR/$OUT/stats -o stats -- $BENCH > $ROOTDIR/$OUT/stats.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $ROOTDIR/$OUT/pmc_sq -o pmc -- $BENCH > $ROOTDIR/$OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $ROOTDIR/$OUT/pmc_sq2 -o pmc -- $BENCH > $ROOTDIR/$OUT/pmc_sq2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $ROOTDIR/$OUT/pmc_fetch -o pmc -- $BENCH > $ROOTDIR/$OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $ROOTDIR/$OUT/pmc_write -o pmc -- $BENCH > $ROOTDIR/$OUT/pmc_write.log 2>&1
cd $ROOTDIR
python tools/summarize_prof.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md
# keep only the small files: the summaries and the per-kernel statistics (gpurun copies at most 64 MiB back)
find $OUT -name "*.db" -delete 2>/dev/null
cp $(ls $OUT/stats/*/*kernel_stats.csv $OUT/stats/*kernel_stats.csv 2>/dev/null | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/stats $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_fetch $OUT/pmc_write
du -sh $OUT
