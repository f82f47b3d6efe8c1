#!/bin/bash
# One GPU visit: parity tests, smoke, bench (headline + the reference's own benchmark).  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/device.txt 2>&1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> gpurun_out/device.txt
rm -f gpurun_out/parity_report.jsonl
timeout ${PYTEST_TIMEOUT:-2400} python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -70 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
timeout 300 python tools/torch_order_check.py > gpurun_out/torch_order.log 2>&1; cat gpurun_out/torch_order.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
tail -5 gpurun_out/bench.log
timeout 900 python bench.py --workload refbench > gpurun_out/bench_refbench.log 2>&1; tail -3 gpurun_out/bench_refbench.log
