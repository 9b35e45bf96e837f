#!/bin/bash
# Round 5: lane-scheduling A/Bs of the prover on one box + kernel traces of the baseline and of the heavy-queue schedule.
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/r05
ROOT=$PWD
F="ZKATTEST_HEAVY_FIFO=1,ZKATTEST_PHASE_MAJOR=1"
timeout 900 python tools/exp_overlap.py --reps 2 \
  base: \
  prio:ZKATTEST_LANE_PRIO=-1/0/1 \
  lds84:ZKATTEST_HEAVY_LDS_KB=84 \
  fifo:ZKATTEST_HEAVY_FIFO=1 \
  fifo_pm:$F \
  fifo_pm_gk:$F,ZKATTEST_GK_BESIDE=1 \
  fifo_pm_gk_hi:$F,ZKATTEST_GK_BESIDE=1,ZKATTEST_LANE_PRIO=-1/-1/-1/-1 \
  fifo_pm_gk_eq:$F,ZKATTEST_GK_BESIDE=1,ZKATTEST_HEAVY_PRIO=0 \
  fifo_pm_gk_lds:$F,ZKATTEST_GK_BESIDE=1,ZKATTEST_HEAVY_LDS_KB=84 \
  fifo2_gk:ZKATTEST_HEAVY_FIFO=2,ZKATTEST_GK_BESIDE=1 \
  fifo_pm_gk_l4@lanes=4,chunk=16384:$F,ZKATTEST_GK_BESIDE=1 \
  base_l4@lanes=4,chunk=16384: \
  > gpurun_out/r05/exp1.log 2> gpurun_out/r05/exp1.err
echo "exp rc=$?"
tail -3 gpurun_out/r05/exp1.err
# bytes under the new schedule: the prover's parity tests with the heavy queue on
ZKATTEST_HEAVY_FIFO=1 ZKATTEST_PHASE_MAJOR=1 ZKATTEST_GK_BESIDE=1 timeout 600 python -m pytest tests/test_gpu_prove.py tests/test_gpu_small_batches.py tests/test_gpu_stream.py -x -q -m gpu > gpurun_out/r05/tests_fifo.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r05/tests_fifo.log
cd /tmp && export TMPDIR=/tmp
BARGS="--no-cpu-baseline --host-io 0 --latency 0 --json-sample 0 --verify-steps 0 --roofline-steps 0 --steps 2 --warmup 1"
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/r05/trace_base -o r -- python $ROOT/bench.py $BARGS > $ROOT/gpurun_out/r05/trace_base.log 2>&1
ZKATTEST_HEAVY_FIFO=1 ZKATTEST_PHASE_MAJOR=1 ZKATTEST_GK_BESIDE=1 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/r05/trace_fifo -o r -- python $ROOT/bench.py $BARGS > $ROOT/gpurun_out/r05/trace_fifo.log 2>&1
cd $ROOT
ls -la gpurun_out/r05/trace_base gpurun_out/r05/trace_fifo
cat gpurun_out/r05/exp1.log | cut -c1-260
