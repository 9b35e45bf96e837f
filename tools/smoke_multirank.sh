# N ranks of bench.py on ONE GPU (gloo process group, all on cuda:0): exercises the N > 1 code path -- ring broadcast, per-rank
# seeds, max-over-ranks timing, rank-0 JSON line with cpu_baseline -- where no multi-GPU node is at hand.  Small tables so that
# the contexts fit one device.   usage: bash tools/smoke_multirank.sh [prove|verify|prove8]
#   prove8: the driver's `--gpus 8` launch line as a dry run (world size 8, 16-bit combs, 2048 proofs per rank)
mode=${1:-prove}
export ZK_BENCH_ONE_DEVICE=1
n=2
if [ "$mode" = verify ]; then
  extra="--mode verify --batch 4096 --ring 65536 --slab 1024 --verify-chunk 512 --verify-lanes 2 --comb-bits 16"
elif [ "$mode" = prove8 ]; then
  n=8
  extra="--batch 2048 --ring 4096 --chunk 1024 --lanes 2 --verify-chunk 1024 --comb-bits 16 --host-io 0 --cpu-sample 2 --json-sample 2"
else
  extra="--batch 4096 --chunk 2048 --lanes 2 --verify-chunk 2048 --comb-bits 16 --host-io 1024 --host-io-chunk 512 --host-io-verify-chunk 512 --cpu-sample 4 --json-sample 2"
fi
python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 2 --warmup 1 $extra
