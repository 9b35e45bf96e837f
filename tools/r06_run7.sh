export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06/gpu_tests_7.log 2>&1; echo "gpu tests rc=$?"; tail -6 gpurun_out/r06/gpu_tests_7.log
timeout 1800 bash tools/repro_profiles.sh > gpurun_out/r06/repro.log 2>&1; echo "repro rc=$?"; tail -30 gpurun_out/r06/repro.log
