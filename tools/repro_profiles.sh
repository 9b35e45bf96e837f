#!/bin/bash
# Regenerates profiles/r03_* on an MI355X box (run from the repo root, e.g. through gpurun; outputs under gpurun_out/, to be copied
# into profiles/).  ROUND=r03 by default.  Counters are collected in their own runs, never together with tracing domains other
# than --kernel-trace.
set -e
R=${ROUND:-r03}
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python bench.py > gpurun_out/b_default.log 2> gpurun_out/b_default.err
grep '"metric"' gpurun_out/b_default.log > gpurun_out/${R}_bench_line.json
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_default -o r -- python $ROOT/bench.py --host-io 0 --no-cpu-baseline --json-sample 0 > $ROOT/gpurun_out/b_prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_lanes1 -o r -- python $ROOT/bench.py --lanes 1 --verify-lanes 1 --host-io 0 --no-cpu-baseline --json-sample 0 > $ROOT/gpurun_out/b_prof1.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
    n=$(echo $c | cut -d" " -f1)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_$n -- python $ROOT/bench.py --batch 16384 --chunk 16384 --steps 1 --warmup 0 \
        --no-cpu-baseline --verify-steps 1 --roofline-steps 0 --host-io 0 --json-sample 0 > $ROOT/gpurun_out/pmc_$n.log 2>&1
done
# the matrix-core kernel of the verifier's ring fold at ring 2^20 (its own counter group; tolerated to fail where a counter name is unknown)
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_MFMA -- \
    python $ROOT/bench.py --mode verify --batch 8192 --ring 1048576 --slab 8192 --verify-chunk 8192 --steps 1 --warmup 0 > $ROOT/gpurun_out/pmc_MFMA.log 2>&1 || true
cd $ROOT
python tools/rocpd_stats.py gpurun_out/prof_default/r_results.db > gpurun_out/${R}_rocprofv3_kernel_stats.csv
python tools/rocpd_stats.py gpurun_out/prof_lanes1/r_results.db > gpurun_out/${R}_rocprofv3_kernel_stats_lanes1.csv
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_WAVES gpurun_out/pmc_SQ_WAIT_ANY > gpurun_out/${R}_pmc_summary_body.txt
python tools/pmc_summary.py gpurun_out/pmc_MFMA > gpurun_out/${R}_pmc_mfma_body.txt   # another workload (verify, ring 2^20, 8192 proofs): its own summary
[ -x tools/valu_peak ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/valu_peak
tools/valu_peak > gpurun_out/${R}_valu_peak_microbench.txt
python tools/exp_stream_timeline.py 22016 3 8192 4 > gpurun_out/${R}_stream_timeline.txt 2>&1 || true
python tools/kernel_meta.py --csv > gpurun_out/${R}_kernel_resources.csv
# BASELINE configs[4]'s per-GPU shard with the ring fold on the matrix cores and on the vector ALU (same box)
for m in 1 0; do
    ZKATTEST_GK_MFMA=$m python bench.py --mode verify --batch 131072 --ring 1048576 --steps 1 --warmup 1 2>/dev/null | grep '"metric"' > gpurun_out/${R}_bench_line_verify_2e20_mfma$m.json
done
python bench.py --pool --gpus 1 --steps 2 --warmup 1 2>/dev/null | grep '"metric"' > gpurun_out/${R}_bench_line_pool.json
python tools/json_rate.py 2048 1 0 > gpurun_out/${R}_json_rates.txt
python tools/exp_set_ring.py 65536 > gpurun_out/${R}_set_ring.txt 2>/dev/null
rm -rf gpurun_out/prof_default gpurun_out/prof_lanes1 gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_WAVES gpurun_out/pmc_SQ_WAIT_ANY gpurun_out/pmc_MFMA
