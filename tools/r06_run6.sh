export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_verify.py -q -m gpu -k "p256" > gpurun_out/r06/gpu_tests_6.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r06/gpu_tests_6.log
timeout 1500 bash tools/ab_handover.sh > gpurun_out/r06/ab_handover.txt 2>&1; echo "handover rc=$?"; cat gpurun_out/r06/ab_handover.txt
