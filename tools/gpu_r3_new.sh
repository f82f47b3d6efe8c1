export TMPDIR=/tmp
python -m pytest tests/test_gpu_round3.py tests/test_full_configs.py -m gpu -q --tb=short -p no:cacheprovider --durations=8 > gpurun_out/r3_new.log 2>&1; tail -60 gpurun_out/r3_new.log
