#!/usr/bin/env python3
"""ZK_IO_DEBUG timeline of one zk_prove_batch call on a page-locked buffer (when does every slice become ready, when does its DMA run)."""
import os
import sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zkp_ecdsa_amd as Z

B, chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 16384
eng = Z.Engine(0)
p = eng.synth_params(2024)
eng.set_comb_bits(24)
eng.set_params(*p, 80)
ring, msg, sig, pk, which, seeds = eng.synth_workload(2024, 65536, B)
eng.set_ring(ring, 65536)
eng.set_chunk(chunk)
pin = Z.PinnedBuffer(int(B * (304 + 336 * 80 + 3392 * 44 + 384 * 16 + 32) + (64 << 20)))
eng.prove_batch_host_raw(msg, sig, pk, which, seeds, out=pin)
os.environ['ZK_IO_DEBUG'] = '1'
dt, _, off, st = eng.prove_batch_host_raw(msg, sig, pk, which, seeds, out=pin)
print('call %.1f ms, %.0f proofs/s' % (dt * 1e3, B / dt))
