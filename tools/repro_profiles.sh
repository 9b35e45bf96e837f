#!/bin/bash
# Regenerates profiles/r06_* on an MI355X box (run from the repo root, e.g. through gpurun; outputs under gpurun_out/, to be copied into
# profiles/).  ROUND=r06 by default.  Counters are collected in their own runs, never together with tracing domains other than --kernel-trace.
# About 13 minutes of GPU time.
set -e
R=${ROUND:-r06}
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
ROOT=$PWD
python bench.py > gpurun_out/b_default.log 2> gpurun_out/b_default.err
grep '"metric"' gpurun_out/b_default.log > gpurun_out/${R}_bench_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_default -o r -- python $ROOT/bench.py --host-io 0 --no-cpu-baseline --json-sample 0 --latency 0 > $ROOT/gpurun_out/b_prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_lanes1 -o r -- python $ROOT/bench.py --lanes 1 --verify-lanes 1 --host-io 0 --no-cpu-baseline --json-sample 0 --latency 0 > $ROOT/gpurun_out/b_prof1.log 2>&1
# PMC at the bench's own batch size (65 536 proofs, 3 chunks of 22 016), prover only: three passes
PMCARGS="--steps 1 --warmup 0 --no-cpu-baseline --verify-steps 0 --roofline-steps 0 --host-io 0 --json-sample 0 --latency 0 --lanes 1"
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
    n=$(echo $c | cut -d" " -f1)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_$n -- python $ROOT/bench.py $PMCARGS > $ROOT/gpurun_out/pmc_$n.log 2>&1 || echo "pmc pass $n failed" >> $ROOT/gpurun_out/pmc_fail.log
done
cd $ROOT
python tools/rocpd_stats.py gpurun_out/prof_default/r_results.db > gpurun_out/${R}_rocprofv3_kernel_stats.csv
python tools/rocpd_stats.py gpurun_out/prof_lanes1/r_results.db > gpurun_out/${R}_rocprofv3_kernel_stats_lanes1.csv
python tools/pmc_families.py 65536 gpurun_out/pmc_SQ_WAVES gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/${R}_pmc_families.txt 2>&1 || true
python tools/pmc_summary.py gpurun_out/pmc_SQ_WAVES gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/${R}_pmc_summary_body.txt 2>&1 || true
python tools/kernel_meta.py --csv > gpurun_out/${R}_kernel_resources.csv
# the reference's own bench ring (100 001 keys -> 2^17) and BASELINE configs[4]'s per-GPU shard
python bench.py --ring 100001 --host-io 0 --json-sample 0 2>/dev/null | grep '"metric"' > gpurun_out/${R}_bench_line_ring100001.json
python bench.py --mode verify --batch 131072 --ring 1048576 --steps 1 --warmup 1 2>/dev/null | grep '"metric"' > gpurun_out/${R}_bench_line_verify_2e20.json
python bench.py --pool --gpus 1 --steps 2 --warmup 1 2>/dev/null | grep '"metric"' > gpurun_out/${R}_bench_line_pool.json
# the build without secret-dependent control flow (INTEGRATION.md "side channels"): what it costs, same box
ZKATTEST_LIB=$ROOT/zkp-ecdsa_amd/lib/libzkattest_hip_uniform.so python bench.py --host-io 0 --json-sample 0 --latency 0 --no-cpu-baseline 2>/dev/null | grep '"metric"' > gpurun_out/${R}_bench_line_uniform.json
python bench.py --host-io 0 --json-sample 0 --latency 0 --no-cpu-baseline 2>/dev/null | grep '"metric"' > gpurun_out/${R}_bench_line_default_again.json
make -s -C bindings/napi OUT=/tmp/zk.node && (cd bindings/napi && ZKATTEST_NODE=/tmp/zk.node node latency.js 7 > $ROOT/gpurun_out/${R}_facade_latency.json 2>/dev/null) || true
rm -rf gpurun_out/prof_default gpurun_out/prof_lanes1 gpurun_out/pmc_SQ_WAVES gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
ls -la gpurun_out/${R}_*
