#!/usr/bin/env python3
"""Cost of slicing the PointAdd phase: device-resident and host-buffer prove rates per (chunk, slice)."""
import os
import sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zkp_ecdsa_amd as Z

B = 65536
dev = torch.device('cuda', 0)
eng = Z.Engine(0)
p = eng.synth_params(2024)
eng.set_comb_bits(24)
eng.set_params(*p, 80)
ring, msg, sig, pk, which, seeds = eng.synth_workload(2024, 65536, B)
eng.set_ring(ring, 65536)
tb = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
d_msg, d_sig, d_pk, d_seeds = tb(msg), tb(sig), tb(pk), tb(seeds)
d_which = torch.tensor(which, dtype=torch.int32, device=dev)
cap = int(B * (304 + 336 * 80 + 3392 * 44 + 384 * 16 + 32) + (64 << 20))
d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
d_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
d_st = torch.empty(B, dtype=torch.int32, device=dev)
pin = Z.PinnedBuffer(cap)
for chunk in (16384, 32768):
    eng.set_chunk(chunk)
    for sl in (1 << 20, 8192, 4096, 2048):
        eng.set_slice(sl)
        f = lambda: eng.prove_batch_device(B, d_msg.data_ptr(), d_sig.data_ptr(), d_pk.data_ptr(), d_which.data_ptr(), d_seeds.data_ptr(), d_out.data_ptr(), cap, d_off.data_ptr(), d_st.data_ptr())
        f()
        torch.cuda.synchronize()
        t0 = time.time()
        f(), f()
        torch.cuda.synchronize()
        dv = 2 * B / (time.time() - t0)
        best = 1e9
        for rep in range(3):
            dt = eng.prove_batch_host_raw(msg, sig, pk, which, seeds, out=pin)[0]
            if rep:
                best = min(best, dt)
        print('chunk %6d slice %8d: device %7.0f /s   host %7.0f /s' % (chunk, sl, dv, B / best), flush=True)
