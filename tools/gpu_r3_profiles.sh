#!/bin/bash
# round-3 evidence: rocprofv3 summaries of the configurations, the other bench lines, the cfg 5 sweep, end-to-end timing
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/profile.sh r03_cfg2 > gpurun_out/prof_cfg2.log 2>&1
bash tools/profile.sh r03_cfg3 --workload cfg3 > gpurun_out/prof_cfg3.log 2>&1
bash tools/profile.sh r03_cfg5pad --workload cfg5pad > gpurun_out/prof_cfg5pad.log 2>&1
bash tools/profile.sh r03_cfg5mraf --workload cfg5mraf > gpurun_out/prof_cfg5mraf.log 2>&1
bash tools/profile.sh r03_cfg5mraf_f64 --workload cfg5mraf --dtype f64 > gpurun_out/prof_cfg5mraf_f64.log 2>&1
bash tools/profile.sh r03_hd --workload hd > gpurun_out/prof_hd.log 2>&1
bash tools/profile.sh r03_cfg2dense --workload cfg2dense > gpurun_out/prof_cfg2dense.log 2>&1
bash tools/profile.sh r03_cfg4 --workload cfg4 --steps 20 > gpurun_out/prof_cfg4.log 2>&1
bash tools/profile.sh r03_cfg4zern --workload cfg4zern --steps 20 > gpurun_out/prof_cfg4zern.log 2>&1
bash tools/profile.sh r03_cfg1 --workload cfg1 --steps 200 > gpurun_out/prof_cfg1.log 2>&1
bash tools/gpu_configs.sh > gpurun_out/configs.log 2>&1; tail -16 gpurun_out/configs.log
timeout 900 python bench.py --workload refbench 2>/dev/null | grep '^{' >> gpurun_out/configs.jsonl
python tools/e2e_timing.py gpurun_out/e2e_timing.json > gpurun_out/e2e.log 2>&1
[ -n "$WITH_SWEEP" ] && { timeout 1500 python tools/cfg5_sweep.py gpurun_out/cfg5_sweep.json > gpurun_out/cfg5_sweep.log 2>&1; tail -3 gpurun_out/cfg5_sweep.log; }
timeout 600 python bench.py > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log | cut -c1-300
