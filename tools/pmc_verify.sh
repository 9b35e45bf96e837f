#!/bin/bash
# rocprofv3 --pmc passes over ONE verify step (65 536 proofs, 2 chunks of 32 768, serial lanes), counters in their own runs with --kernel-trace only.
# Output: gpurun_out/r06_pmc_verify.txt (per kernel: VALU activity, waits, FETCH_SIZE / WRITE_SIZE per launch).  About 1.5 minutes of GPU time.
set -e
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
ROOT=$PWD
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --verify-steps 1 --verify-warmup 1 --roofline-steps 0 --host-io 0 --json-sample 0 --latency 0 --lanes 1 --verify-lanes 1"
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
    n=$(echo $c | cut -d" " -f1)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmcv_$n -- python $ROOT/bench.py $ARGS > $ROOT/gpurun_out/pmcv_$n.log 2>&1 || echo "pmc pass $n failed"
done
cd $ROOT
python tools/pmc_summary.py gpurun_out/pmcv_SQ_WAVES gpurun_out/pmcv_FETCH_SIZE gpurun_out/pmcv_WRITE_SIZE > gpurun_out/pmcv_all.txt
python - <<'EOF' > gpurun_out/r06_pmc_verify.txt
import re
keep = ('k_msm_', 'k_pm_', 'k_v_', 'k_rtab', 'k_gkm', 'k_tom_commit_list')
out, on = [], False
for line in open('gpurun_out/pmcv_all.txt'):
    if not line.startswith(' '):
        on = any(k in line for k in keep)
    if on:
        out.append(line.rstrip())
print('# rocprofv3 --pmc over one verify step (tools/pmc_verify.sh): 65 536 proofs, 2 chunks of 32 768, serial lanes; FETCH_SIZE / WRITE_SIZE in the counter\'s raw units')
print('# (KB per the guide; the guide\'s x2 correction of FETCH_SIZE on gfx950 is NOT applied here), SQ_* summed over the XCDs')
print('\n'.join(out))
EOF
# the same passes by kernel family and for the whole step (three verify steps in the run: warm-up, timed, family pass -- all single-lane here)
python tools/pmc_families.py --verify --steps 3 65536 gpurun_out/pmcv_SQ_WAVES gpurun_out/pmcv_FETCH_SIZE gpurun_out/pmcv_WRITE_SIZE > gpurun_out/r06_pmc_families_verify.txt 2>&1 || true
rm -rf gpurun_out/pmcv_SQ_WAVES gpurun_out/pmcv_FETCH_SIZE gpurun_out/pmcv_WRITE_SIZE
wc -l gpurun_out/r06_pmc_verify.txt
