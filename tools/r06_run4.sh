export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06/gpu_tests_4.log 2>&1; echo "gpu tests rc=$?"; tail -6 gpurun_out/r06/gpu_tests_4.log
timeout 900 python bench.py > gpurun_out/r06/bench_full.log 2> gpurun_out/r06/bench_full.err; echo "bench rc=$?"; grep '"metric"' gpurun_out/r06/bench_full.log | cut -c1-200
timeout 1500 bash tools/ab_r06.sh > gpurun_out/r06/ab_r06.txt 2>&1; echo "ab rc=$?"; tail -60 gpurun_out/r06/ab_r06.txt
timeout 900 bash tools/pmc_verify.sh > gpurun_out/r06/pmc_verify.log 2>&1; echo "pmc rc=$?"; cat gpurun_out/r06_pmc_families_verify.txt
