export GPU_MAX_HW_QUEUES=8
ROOT=$PWD
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06/gpu_tests_3.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/r06/gpu_tests_3.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/r06/b1prof -o r -- python $ROOT/tools/b1_timeline.py run > $ROOT/gpurun_out/r06/b1_run.log 2>&1; echo "b1 rc=$?"
cd $ROOT
grep "^call" gpurun_out/r06/b1_run.log
python tools/b1_timeline.py parse gpurun_out/r06/b1prof/r_results.db > gpurun_out/r06/b1_timeline_coop1.txt 2>&1
rm -rf gpurun_out/r06/b1prof
timeout 600 python bench.py --host-io 0 --json-sample 0 --no-cpu-baseline > gpurun_out/r06/bench_coop1.log 2>&1; echo "bench rc=$?"; grep '"metric"' gpurun_out/r06/bench_coop1.log | cut -c1-300
ZKATTEST_ONE_LANE_CHAINS=1 timeout 600 python bench.py --host-io 0 --json-sample 0 --no-cpu-baseline --latency 0 > gpurun_out/r06/bench_coop1_onelane.log 2>&1; echo "bench rc=$?"; grep '"metric"' gpurun_out/r06/bench_coop1_onelane.log | cut -c1-300
