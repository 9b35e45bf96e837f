mkdir -p gpurun_out/abw
for v in 24 22 20 18 24; do
  timeout 300 python bench.py --no-cpu-baseline --verify-steps 0 --steps 1 --warmup 1 --comb-bits $v > gpurun_out/abw/w$v.json 2> gpurun_out/abw/w$v.err
  echo "$v rc=$?"
done
