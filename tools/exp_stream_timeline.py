#!/usr/bin/env python3
"""Timeline of the streamed prove calls (ZK_STREAM_DEBUG=1): N batches back to back, two in flight.
   python tools/exp_stream_timeline.py [chunk] [lanes] [slice] [batches] [batch]"""
import os
import sys
import time
os.environ['ZK_STREAM_DEBUG'] = '1'
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkp_ecdsa_amd as Z

chunk, lanes, slc, nj, B = (int(a) for a in (sys.argv[1:6] + ['22016', '3', '262144', '4', '65536'][len(sys.argv) - 1:]))
eng = Z.Engine(0)
eng.set_comb_bits(int(os.environ.get('COMB', '24')))
eng.set_params(*eng.synth_params(1), 80)
ring, msg, sig, pk, which, seeds = eng.synth_workload(1, 65536, B)
eng.set_ring(ring, 65536)
eng.set_chunk(chunk), eng.set_lanes(lanes), eng.set_slice(slc)
cap = int(B * (304 + 336 * 80 + 3392 * 44 + 384 * 16 + 32) + (64 << 20))
pins = [Z.PinnedBuffer(cap), Z.PinnedBuffer(cap)]
for rnd in range(2):
    t0 = time.time()
    tk = []
    for k in range(nj):
        tk.append(eng.prove_submit(msg, sig, pk, which, seeds, pins[k % 2]))
        if k:
            eng.prove_wait(tk[k - 1])
    eng.prove_wait(tk[-1])
    print('round %d: %.1f proofs/s' % (rnd, nj * B / (time.time() - t0)), file=sys.stderr)
