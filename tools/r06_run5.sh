export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06/gpu_tests_5.log 2>&1; echo "gpu tests rc=$?"; tail -15 gpurun_out/r06/gpu_tests_5.log
