#!/usr/bin/env python3
"""Same-box experiment behind the host-pointer numbers of DESIGN.md section 5: device-resident rate per chunk size, and the
PCIe-inclusive rates of zk_prove_batch / zk_verify_batch on page-locked buffers for every (chunk cap, plan) combination.
  python tools/exp_hostio.py [--batch 65536] [--ring 65536] [--comb-bits 24]"""
import argparse
import json
import os
import sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=65536)
    ap.add_argument('--ring', type=int, default=65536)
    ap.add_argument('--comb-bits', type=int, default=24)
    ap.add_argument('--chunks', default='8192,16384')
    ap.add_argument('--lanes', default='2,3,4')
    ap.add_argument('--tapers', default='1')
    ap.add_argument('--slice', type=int, default=0, help='zk_ctx_set_slice (0 = the engine default)')
    args = ap.parse_args()
    import torch
    import zkp_ecdsa_amd as Z
    import bench
    dev = torch.device('cuda', 0)
    B, sec = args.batch, 80
    eng = Z.Engine(0)
    params = eng.synth_params(2024)
    eng.set_comb_bits(args.comb_bits)
    eng.set_params(*params, sec)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(2024, args.ring, B)
    eng.set_ring(ring, args.ring)
    tb = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_msg, d_sig, d_pk, d_seeds = tb(msg), tb(sig), tb(pk), tb(seeds)
    d_which = torch.tensor(which, dtype=torch.int32, device=dev)
    cap = int(B * (304 + 336 * sec + 3392 * (sec // 2 + 4) + 384 * 20 + 32) + (64 << 20))
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
    d_st = torch.empty(B, dtype=torch.int32, device=dev)
    d_ok = torch.empty(B, dtype=torch.uint8, device=dev)
    res = {'pcie': bench.pcie_bandwidth(dev), 'device': {}, 'host': {}}
    print(json.dumps(res['pcie']), flush=True)
    pin = Z.PinnedBuffer(cap)
    if args.slice:
        eng.set_slice(args.slice)
    for lanes in [int(x) for x in args.lanes.split(',')]:
        eng.set_lanes(lanes)
        for c, taper in [(int(x), int(y)) for x in args.chunks.split(',') for y in args.tapers.split(',')]:
            eng.set_chunk(min(c, B)), eng.set_host_taper(taper)
            r = {}
            for name, f in (('prove', lambda: eng.prove_batch_device(B, d_msg.data_ptr(), d_sig.data_ptr(), d_pk.data_ptr(), d_which.data_ptr(), d_seeds.data_ptr(),
                                                                    d_out.data_ptr(), cap, d_off.data_ptr(), d_st.data_ptr())),
                            ('verify', lambda: eng.verify_batch_device(B, d_msg.data_ptr(), d_out.data_ptr(), d_off.data_ptr(), d_seeds.data_ptr(), d_ok.data_ptr(), d_st.data_ptr()))):
                f()
                torch.cuda.synchronize()
                t0 = time.time()
                f(), f()
                torch.cuda.synchronize()
                r[name] = round(2 * B / (time.time() - t0), 0)
            best = [1e9, 1e9]
            for rep in range(3):
                pdt, _, off, st = eng.prove_batch_host_raw(msg, sig, pk, which, seeds, out=pin)
                vdt, ok, vst = eng.verify_batch_host_raw(msg, pin, off, B)
                if rep:
                    best = [min(best[0], pdt), min(best[1], vdt)]
            nbytes = int(off[B])
            r.update({'host_prove': round(B / best[0]), 'host_verify': round(B / best[1]), 'd2h_gbps': round(nbytes / best[0] / 1e9, 1), 'h2d_gbps': round(nbytes / best[1] / 1e9, 1),
                      'accepted': int(sum(ok)), 'failed': int(sum(1 for x in st if x)), 'prove_frac': round(B / best[0] / r['prove'], 3),
                      'hbm_gb': round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 2**30, 1)})
            res['device']['%dx%d/%d' % (lanes, c, taper)] = r
            print('lanes', lanes, 'chunk', c, 'stagger', taper, r, flush=True)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
