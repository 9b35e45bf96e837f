# same-box A/B of the verifier's cross-proof P-256 pass (k_pmsm.hip): on (default threshold), off (ZKATTEST_P256_BATCH=0), on again; then a serial kernel trace
mkdir -p gpurun_out
ARGS="--steps 1 --warmup 1 --roofline-steps 0 --verify-steps 3 --host-io 0 --json-sample 0 --latency 0 --no-cpu-baseline"
for v in on off on2; do
  e=8192; [ $v = off ] && e=0
  ZKATTEST_P256_BATCH=$e timeout 300 python bench.py $ARGS 2>gpurun_out/p256_$v.err | grep '"metric"' > gpurun_out/p256_$v.json
  python - <<EOF
import json
d=json.loads(open("gpurun_out/p256_$v.json").read())["verify"]
print("$v", d["value"], d["ms_per_step"], json.dumps(d["gpu_ms_by_family_per_step"]))
EOF
done
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_p256 -o r -- python $ROOT/bench.py $ARGS --verify-lanes 1 > $ROOT/gpurun_out/p256_prof.log 2>&1
cd $ROOT
python tools/rocpd_stats.py gpurun_out/prof_p256/r_results.db > gpurun_out/p256_kernel_stats_lanes1.csv
grep -i "k_pm_\|k_v_p256\|k_msm_bucket<\|k_msm_reduce" gpurun_out/p256_kernel_stats_lanes1.csv
rm -rf gpurun_out/prof_p256
