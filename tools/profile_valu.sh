#!/bin/bash
# VALU instruction-mix PMC passes for the two hot kernels
OUT=gpurun_out/prof_valu; mkdir -p $OUT; export TMPDIR=/tmp; ROOTDIR=$(pwd)
BENCH="python $ROOTDIR/bench.py --steps 30 --warmup 5 --cpu-iters 0 --no-roofline-pass --pmc 0"
cd /tmp
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $ROOTDIR/$OUT/p1 -o pmc -- $BENCH > $ROOTDIR/$OUT/p1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES -d $ROOTDIR/$OUT/p2 -o pmc -- $BENCH > $ROOTDIR/$OUT/p2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $ROOTDIR/$OUT/p3 -o pmc -- $BENCH > $ROOTDIR/$OUT/p3.log 2>&1
cd $ROOTDIR
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for path in glob.glob("gpurun_out/prof_valu/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k=row["Kernel_Name"]; 
        if "col_tile" in k: k="col_tile"
        elif "row_kernel<float, 4096, 2" in k: k="row2"
        else: continue
        a=agg[k][row["Counter_Name"]]; a[0]+=float(row["Counter_Value"]); a[1]+=1
for k in agg:
    print(k)
    for c in sorted(agg[k]): print("  %-26s %.4g"%(c, agg[k][c][0]/agg[k][c][1]))
PY
