#!/bin/bash
# Regenerates profiles/r02_* on an MI355X box (run from the repo root, e.g. through gpurun; outputs under gpurun_out/).
set -e
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python bench.py > gpurun_out/b_default.log 2> gpurun_out/b_default.err
grep '"metric"' gpurun_out/b_default.log > gpurun_out/r02_bench_line.json
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_default -o r -- python $ROOT/bench.py --host-io 0 --no-cpu-baseline > $ROOT/gpurun_out/b_prof.log 2>&1
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_lanes1 -o r -- python $ROOT/bench.py --lanes 1 --verify-lanes 1 --host-io 0 --no-cpu-baseline > $ROOT/gpurun_out/b_prof1.log 2>&1
# counters: one group per run, never together with tracing domains other than --kernel-trace
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
    n=$(echo $c | cut -d" " -f1)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_$n -- python $ROOT/bench.py --batch 16384 --chunk 16384 --steps 1 --warmup 0 \
        --no-cpu-baseline --verify-steps 0 --roofline-steps 0 --host-io 0 --json-sample 0 > $ROOT/gpurun_out/pmc_$n.log 2>&1
done
cd $ROOT
python tools/rocpd_stats.py gpurun_out/prof_default/r_results.db > gpurun_out/r02_rocprofv3_kernel_stats.csv
python tools/rocpd_stats.py gpurun_out/prof_lanes1/r_results.db > gpurun_out/r02_rocprofv3_kernel_stats_lanes1.csv
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_WAVES gpurun_out/pmc_SQ_WAIT_ANY > gpurun_out/r02_pmc_summary_body.txt
tools/valu_peak > gpurun_out/r02_valu_peak_microbench.txt
tools/stream_overlap > gpurun_out/r02_stream_overlap.txt
python tools/exp_io_timeline.py 65536 16384 > gpurun_out/r02_io_timeline.txt 2>&1 || true
python tools/kernel_meta.py --csv > gpurun_out/r02_kernel_resources.csv
rm -rf gpurun_out/prof_default gpurun_out/prof_lanes1 gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_WAVES gpurun_out/pmc_SQ_WAIT_ANY
