mkdir -p gpurun_out/abs
python bench.py --no-cpu-baseline --host-io 0 --steps 1 > gpurun_out/abs/d1.json 2>/dev/null
HSA_SCRATCH_SINGLE_LIMIT=4000000000 python bench.py --no-cpu-baseline --host-io 0 --steps 1 > gpurun_out/abs/lim.json 2>/dev/null
python bench.py --no-cpu-baseline --host-io 0 --steps 1 > gpurun_out/abs/d2.json 2>/dev/null
python bench.py --no-cpu-baseline --host-io 0 --steps 1 --lanes 1 > gpurun_out/abs/l1.json 2>/dev/null
rocm-smi --showmeminfo vram 2>/dev/null | head -5
