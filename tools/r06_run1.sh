export GPU_MAX_HW_QUEUES=8
ROOT=$PWD
mkdir -p gpurun_out/r06
timeout 300 tools/coop_bench > gpurun_out/r06/coop_bench_v1.txt 2>&1; echo "coop_bench rc=$?"
cat gpurun_out/r06/coop_bench_v1.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/r06/b1prof -o r -- python $ROOT/tools/b1_timeline.py run > $ROOT/gpurun_out/r06/b1_run.log 2>&1; echo "b1 rc=$?"
cd $ROOT
tail -8 gpurun_out/r06/b1_run.log
python tools/b1_timeline.py parse gpurun_out/r06/b1prof/r_results.db > gpurun_out/r06/b1_timeline_before.txt 2>&1
rm -rf gpurun_out/r06/b1prof
head -5 gpurun_out/r06/b1_timeline_before.txt
