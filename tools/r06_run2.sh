export GPU_MAX_HW_QUEUES=8
ROOT=$PWD
mkdir -p gpurun_out/r06
timeout 300 tools/coop_bench > gpurun_out/r06/coop_bench_v2.txt 2>&1; echo "coop_bench rc=$?"
tail -12 gpurun_out/r06/coop_bench_v2.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06/gpu_tests_2.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/r06/gpu_tests_2.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/r06/b1prof -o r -- python $ROOT/tools/b1_timeline.py run > $ROOT/gpurun_out/r06/b1_run.log 2>&1; echo "b1 rc=$?"
cd $ROOT
grep "^call" gpurun_out/r06/b1_run.log
python tools/b1_timeline.py parse gpurun_out/r06/b1prof/r_results.db > gpurun_out/r06/b1_timeline_divsteps.txt 2>&1
rm -rf gpurun_out/r06/b1prof
timeout 600 python bench.py --host-io 0 --json-sample 0 --no-cpu-baseline > gpurun_out/r06/bench_divsteps.log 2>&1; echo "bench rc=$?"; grep '"metric"' gpurun_out/r06/bench_divsteps.log | cut -c1-400
