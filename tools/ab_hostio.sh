mkdir -p gpurun_out/hio
for cfg in "16384 32768" "16384 4096" "16384 2048" "65536 8192"; do set -- $cfg
 python bench.py --no-cpu-baseline --verify-steps 0 --steps 1 --host-io $1 --host-io-chunk $2 > gpurun_out/hio/h_$1_$2.json 2> gpurun_out/hio/h_$1_$2.err; echo "$cfg rc=$?"
done
