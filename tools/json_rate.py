#!/usr/bin/env python3
"""Host-only rates of the batch JSON converters (zk_proofs_to_json_batch / zk_proofs_from_json_batch) on synthetic ZKA1 proofs of
the headline shape (secLevel 80, n = 16, 40 zero bits: 169 KB binary, ~596 KB of text).  The converters do not look at curve
membership, so random field bytes do.   python tools/json_rate.py [n_proofs] [threads ...]"""
import ctypes as C
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkp_ecdsa_amd as Z


def fake_proof(rnd, sec=80, n=16, zeros=40):
    bits = [1] * sec
    for i in rnd.sample(range(sec), zeros):
        bits[i] = 0
    total = 304 + 336 * sec + 3392 * zeros + (4 * 72 + 96) * n + 32
    b = bytearray(rnd.getrandbits(8 * total).to_bytes(total, 'big'))
    b[0:4] = b'ZKA1'
    b[4:8], b[8:12], b[12:16] = total.to_bytes(4, 'big'), sec.to_bytes(4, 'big'), n.to_bytes(4, 'big')
    hb = bytearray(16)
    for i, v in enumerate(bits):
        if v:
            hb[15 - (i >> 3)] |= 1 << (i & 7)
    b[16:32] = hb
    return bytes(b)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    threads = [int(a) for a in sys.argv[2:]] or [1, 0]
    rnd = random.Random(1)
    base = [fake_proof(rnd) for _ in range(8)]
    ps = [base[i % 8] for i in range(n)]
    L = Z.lib()
    off = (C.c_uint64 * (n + 1))()
    for i, p in enumerate(ps):
        off[i + 1] = off[i] + len(p)
    blob = (C.c_uint8 * off[n]).from_buffer_copy(b''.join(ps))
    cap = int(3.7 * off[n]) + 4096 * n
    out, toff, st = (C.c_uint8 * cap)(), (C.c_uint64 * (n + 1))(), (C.c_int32 * n)()
    back, poff = (C.c_uint8 * off[n])(), (C.c_uint64 * (n + 1))()
    for th in threads:
        t0 = time.time()
        rc = L.zk_proofs_to_json_batch(n, blob, off, out, cap, toff, st, th)
        t1 = time.time()
        rc2 = L.zk_proofs_from_json_batch(n, out, toff, back, off[n], poff, st, th)
        t2 = time.time()
        assert rc == 0 and rc2 == 0 and bytes(back) == bytes(blob)
        print('threads=%-3s to_json %8.0f proofs/s  from_json %8.0f proofs/s  (%d proofs, %d B -> %d B of text each, %d cpus)'
              % (th or 'all', n / (t1 - t0), n / (t2 - t1), n, len(ps[0]), toff[1], os.cpu_count()))


if __name__ == '__main__':
    main()
