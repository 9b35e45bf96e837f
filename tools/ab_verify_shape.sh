# same-box comparison of the verify half's chunk x lanes shapes: SHAPES="chunk:lanes ..." bash tools/ab_verify_shape.sh
ARGS="--steps 1 --warmup 1 --roofline-steps 0 --verify-steps 3 --host-io 0 --json-sample 0 --latency 0 --no-cpu-baseline"
for cfg in ${SHAPES:-32768:2 22016:3 16384:2 16384:3 16384:4 21846:2 32768:2}; do
  c=${cfg%%:*}; l=${cfg##*:}
  timeout 300 python bench.py $ARGS --verify-chunk $c --verify-lanes $l 2>gpurun_out/vs.err | grep '"metric"' > gpurun_out/vs.json
  python - <<EOF
import json
try:
    d=json.loads(open("gpurun_out/vs.json").read())["verify"]
    print("chunk $c lanes $l:", d["value"], d["ms_per_step"])
except Exception as e:
    print("chunk $c lanes $l: failed", e, open("gpurun_out/vs.err").read()[-300:])
EOF
done
