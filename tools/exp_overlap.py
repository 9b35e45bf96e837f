#!/usr/bin/env python3
"""Same-box A/B of the prover's lane scheduling knobs (round 5: heavy queue, stream priorities, LDS pad), ONE process, one context per variant.

A variant is `name[@lanes=L,chunk=C]:ENV=VALUE,ENV=VALUE,...` (the environment is read by zk_ctx_create and the launch wrappers, so a fresh
context per variant sees it).  Every variant proves the same 65 536-proof step; the sum of all output bytes is compared with the first
variant's (the bytes must not depend on the schedule).  Prints one line per variant and, with --reps 2, a second interleaved round.

    python tools/exp_overlap.py base: fifo:ZKATTEST_HEAVY_FIFO=1,ZKATTEST_PHASE_MAJOR=1 ...
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KNOBS = ['ZKATTEST_LANE_STAGGER', 'ZKATTEST_LANE_PRIO', 'ZKATTEST_HEAVY_FIFO', 'ZKATTEST_HEAVY_PRIO', 'ZKATTEST_PHASE_MAJOR', 'ZKATTEST_GK_BESIDE', 'ZKATTEST_HEAVY_LDS_KB', 'ZKATTEST_LANE_CUS', 'ZKATTEST_HEAVY_CUS']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('variants', nargs='+')
    ap.add_argument('--batch', type=int, default=65536)
    ap.add_argument('--ring', type=int, default=65536)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--reps', type=int, default=1)
    ap.add_argument('--comb-bits', type=int, default=24)
    ap.add_argument('--families', action='store_true', help='one extra serial pass per variant for the per-family milliseconds')
    args = ap.parse_args()
    import torch
    import zkp_ecdsa_amd as Z
    dev = torch.device('cuda', 0)
    B, nkeys, sec = args.batch, args.ring, 80
    tb = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    work = None
    ref_sum = None
    results = {}
    for rep in range(args.reps):
        for spec in args.variants:
            head, _, envs = spec.partition(':')
            name, _, opts = head.partition('@')
            lanes, chunk = 3, 22016
            for o in filter(None, opts.split(',')):
                k, v = o.split('=')
                if k == 'lanes':
                    lanes = int(v)
                elif k == 'chunk':
                    chunk = int(v)
            for k in KNOBS:
                os.environ.pop(k, None)
            for e in filter(None, envs.split(',')):
                k, v = e.split('=', 1)
                os.environ[k] = v.replace('/', ',')   # priorities are written -1/0/1 on the command line
            eng = Z.Engine(0)
            nh, tg, th = eng.synth_params(2024)
            eng.set_comb_bits(args.comb_bits)
            eng.set_params(nh, tg, th, sec)
            eng.set_chunk(min(chunk, B))
            eng.set_lanes(lanes)
            if work is None:
                ring, msg, sig, pk, which, seeds = eng.synth_workload(2024, nkeys, B)
                work = (tb(ring), tb(msg), tb(sig), tb(pk), tb(seeds), torch.tensor(which, dtype=torch.int32, device=dev))
                cap = int(B * (304 + 336 * sec + 3392 * (sec // 2 + 4) + (4 * 72 + 96) * 20 + 32) + (64 << 20))
                d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
                d_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
                d_st = torch.empty(B, dtype=torch.int32, device=dev)
            d_ring, d_msg, d_sig, d_pk, d_seeds, d_which = work
            eng.set_ring_device(d_ring.data_ptr(), nkeys)
            torch.cuda.synchronize()

            def step():
                eng.prove_batch_device(B, d_msg.data_ptr(), d_sig.data_ptr(), d_pk.data_ptr(), d_which.data_ptr(), d_seeds.data_ptr(),
                                       d_out.data_ptr(), cap, d_off.data_ptr(), d_st.data_ptr())
            d_out.zero_()
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt = (time.time() - t0) / args.steps
            total = int(d_off[B].item())
            osum = int(torch.sum(d_out[:total], dtype=torch.int64).item()) + 1000003 * int(torch.sum(d_off, dtype=torch.int64).item() % 1000000007)
            bad = int((d_st != 0).sum().item())
            if ref_sum is None:
                ref_sum = osum
            fam = None
            if args.families and rep == 0:
                eng.set_lanes(1)
                step()
                _, fam = eng.last_timing()
                fam = {k: round(v, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])}
            r = {'variant': name, 'rep': rep, 'lanes': lanes, 'chunk': chunk, 'ms_per_step': round(dt * 1e3, 2), 'proofs_per_s': round(B / dt, 1),
                 'bytes_equal_first_variant': osum == ref_sum, 'failed_proofs': bad, 'env': envs}
            if fam:
                r['serial_ms_by_family'] = fam
            results.setdefault(name, []).append(r['proofs_per_s'])
            print(json.dumps(r), flush=True)
            eng.close()
            del eng
    print('summary (proofs/s):', json.dumps({k: v for k, v in results.items()}))


if __name__ == '__main__':
    main()
