#!/usr/bin/env python3
"""Does a device-to-host DMA stream slow the prover's kernels down?  zk_prove_batch_device (65 536 proofs, chunk 16384 x 2 lanes) alone, and
with a second host thread copying an unrelated 1 GiB device buffer to page-locked host memory back to back on its own stream."""
import os
import sys
import threading
import time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zkp_ecdsa_amd as Z

B, sec = 65536, 80
dev = torch.device('cuda', 0)
eng = Z.Engine(0)
eng.set_comb_bits(24)
eng.set_params(*eng.synth_params(2024), sec)
ring, msg, sig, pk, which, seeds = eng.synth_workload(2024, 65536, B)
eng.set_ring(ring, 65536)
eng.set_chunk(int(sys.argv[1]) if len(sys.argv) > 1 else 16384)
eng.set_lanes(int(sys.argv[2]) if len(sys.argv) > 2 else 2)
tb = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
d_msg, d_sig, d_pk, d_seeds = tb(msg), tb(sig), tb(pk), tb(seeds)
d_which = torch.tensor(which, dtype=torch.int32, device=dev)
cap = int(B * (304 + 336 * sec + 3392 * (sec // 2 + 4) + 384 * 20 + 32) + (64 << 20))
d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
d_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
d_st = torch.empty(B, dtype=torch.int32, device=dev)


def step():
    eng.prove_batch_device(B, d_msg.data_ptr(), d_sig.data_ptr(), d_pk.data_ptr(), d_which.data_ptr(), d_seeds.data_ptr(), d_out.data_ptr(), cap, d_off.data_ptr(), d_st.data_ptr())


def timed(n=3):
    step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3

print('alone            %.1f ms per step' % timed())
src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
dst = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
streams = [torch.cuda.Stream() for _ in range(8)]   # HIP maps streams to hardware queues round-robin: a stream that shares a queue with one of
# the engine's makes every kernel wait for a 1 GiB copy; the quietest candidate is the one on a queue of its own
for k, cs in enumerate(streams):
    stop, copied = [False], [0]

    def copier():
        torch.cuda.set_device(0)
        with torch.cuda.stream(cs):
            while not stop[0]:
                dst.copy_(src, non_blocking=True)
                cs.synchronize()
                copied[0] += 1

    th = threading.Thread(target=copier)
    th.start()
    time.sleep(0.2)
    c0, t0 = copied[0], time.time()
    ms = timed(1)
    gbps = (copied[0] - c0) * (1 << 30) / (time.time() - t0) / 1e9
    stop[0] = True
    th.join()
    print('with a D2H loop on stream %d: %.1f ms per step (%.1f GB/s copied meanwhile)' % (k, ms, gbps), flush=True)
