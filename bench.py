#!/usr/bin/env python3
"""bench.py -- proveSignatureList throughput of the MI355X engine (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one zk_prove_batch_device call over one batch of B synthetic proofs (default B = 65536, ring = 2^16:
BASELINE.json configs[2]) with every input already resident in HBM and the proofs left in HBM.  Multi-GPU is weak
scaling: proofs are independent given (params, ring), so every rank proves its own B proofs; the only collective is
the RCCL broadcast of the key ring at set-up (outside the timed region), as the north star prescribes.

The JSON line carries `roofline` for the dominant kernel (k_tom_commit: integer VALU bound)
and `cpu_baseline` (the C restatement in oracle/, timed on this box's host cores on a bounded sample).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps its streams onto 4 hardware queues by default; the engine's lanes and copy streams want their own (csrc/api.hip,
# zk_ctx_create).  Must be in the environment before the first HIP call of the process, i.e. before torch touches the device.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

# measured on MI355X with tools/valu_peak.hip (profiles/r01_valu_peak_microbench.txt, r02_...): v_mad_u64_u32 chip-wide issue
# rate at 16 independent accumulators x 8 waves/SIMD (4.2 cycles per wave-instruction; 39.3 T/s would be 16 lanes/clk at 2.4 GHz);
# 36.66 and 37.11 T/s were measured on two boxes of the pool in round 1 (37.28-37.36 T/s sustained over 2 s in round 2); the
# denominator stays 37.11 so that the fractions of the two rounds compare
VALU_MAD_PEAK_TOPS = 37.11
# the same instruction at the kernel's OWN parallelism: 4 lock-step chains per wave x 2 waves per SIMD (212 VGPRs) = 8 independent
# accumulator chains per SIMD issue at 5.7 cycles per wave-instruction (tools/valu_peak.hip "NACC= 4 waves/SIMD=2",
# profiles/r03_valu_peak_microbench.txt: 27.83 T/s; 16 chains 31.9, 64 chains 36.3, 128 chains 37.2)
VALU_MAD_8CHAIN_TOPS = 27.83
HBM_PEAK_GBPS = 8000.0
# multiplier-pipe instructions (v_mad_u64_u32 + v_mul_lo_u32) per Tom-field Montgomery product: 1224 + 72 per table addition
# of 8 products in k_tom_commit (tools/isa_blocks.py; nominal 171 = 81 + 81 + 9, the modulus limb that is zero costs nothing);
# PMC (profiles/r03_pmc_summary.txt; r02: 36 674): 36 769 VALU wave-instructions per unpaired commitment of 163 products on average = 225
# instructions per product, 230 in the paired kernel (round 1: 239)
MACS_PER_MODMUL = 162


def tom_commit_modmuls(comb_bits):
    """executed per commitment: 2 x ceil(256/W) table additions of a W-bit comb, 8 modmuls each"""
    return 2 * ((256 + comb_bits - 1) // comb_bits) * 8


TOM_COMMIT_NOMINAL = 4064      # reference: 256 dbl + 160 add (src/curves/group.ts:97-132)
TOM_COMMIT_BYTES = 2 * 36 + 3 * 36  # algorithmic HBM bytes per commitment: read (v, r), write (X, Y, Z)
# PMC passes (profiles/r03_pmc_summary.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate runs, batch 16384),
# bytes per commitment through the L2's memory-side port, keyed by comb width.  24 bits (128-byte table entries, 47 GB of
# tables): 2 x 1310 B fetched (round 2: 2 x 1251) (gfx950 tallies 16-byte-per-lane loads at half their bytes, MI355X_MICROARCH.md section HBM;
# 20.3 gathers x 128 B = 2600 B expected) + 112 B written.  16 bits (112-byte entries, 235 MB; round-1 pass): 3238 B (raw) + 111 B.
TOM_COMMIT_PMC_BYTES = {24: 2620 + 113, 16: 3238 + 111}
PMC_SOURCE = 'profiles/r03_pmc_summary.txt (separate rocprofv3 --pmc passes at batch 16384; constants of bench.py, NOT measured in this run)'
# same passes, SQ counters at 24 bits: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES per wave at 2 waves per SIMD (VALU pipe busy
# 99 % of the time), SQ_WAIT_INST_ANY 0.378, SQ_WAIT_ANY (memory) 0.117
TOM_COMMIT_VALU_ACTIVE_PER_WAVE = {24: 0.495}
DEFAULT_COMB_BITS = 24


def rank_seeds(base_seeds: bytes, rank: int) -> bytes:
    """Per-rank RNG seeds: rank 0 keeps the synthetic seeds, rank r > 0 re-keys them (distinct proofs, same statements)."""
    if rank == 0:
        return base_seeds
    out = bytearray()
    tag = b'rank' + (rank.to_bytes(4, 'big') if rank < (1 << 32) else rank.to_bytes(12, 'big'))
    for i in range(0, len(base_seeds), 32):
        out += hashlib.sha256(tag + base_seeds[i:i + 32]).digest()
    return bytes(out)


def nominal_modmuls(n_log2, z=40):
    """Reference-algorithm modular multiplications per proof (SURVEY.md section 8(d) / BASELINE.md section 2)."""
    wt = (162 + 26 * z + 4 * n_log2) * 4064 + 8 * z * 3184
    wq = (163 + z) * 4448 + 5568
    ring = 2 * (1 << n_log2) * n_log2
    return wt, wq, ring


def host_cores():
    """CPUs this process may actually use: min(affinity, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return max(1, min(n, 256))


def cpu_baseline(nh, tg, th, ring, nkeys, msg, sig, pk, which, seeds, sec, budget_proofs):
    """Oracle (C restatement, reference-faithful algorithms) on this box's host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import coracle as CO
    nthreads = host_cores()
    n = min(len(which), budget_proofs)
    octx = CO.OracleCtx(nh, tg, th, sec)
    octx.set_ring(ring, nkeys)
    t0 = time.time()
    proofs, st = octx.prove_batch(msg[:32 * n], sig[:64 * n], pk[:64 * n], which[:n], seeds=seeds[:32 * n], nthreads=nthreads)
    dt = time.time() - t0
    assert all(s == 0 for s in st)
    return {'value': n / dt, 'unit': 'proofs/s', 'cores': nthreads, 'kind': 'port',
            'sample': '%d proofs of the same workload (ring=%d keys, secLevel %d), %d threads, %.1f s wall' % (n, nkeys, sec, nthreads, dt)}, proofs


def v8_bigint_indicator():
    """Optional: the plain-JS BigInt restatement oracle/js/zkattest_ref.js proves and verifies
    one golden proof at ring = 6 keys padded to 8, secLevel 80 (the shape of BASELINE configs[0] and of the reference's own
    test) on whatever `node` the box has -- an approximation of `npm run bench`, which needs Node >= 24 and cannot run here."""
    import shutil
    import subprocess
    if shutil.which('node') is None:
        return None
    try:
        p = subprocess.run(['node', os.path.join(ROOT, 'oracle', 'js', 'zkattest_ref.js'), 'bench', os.path.join(ROOT, 'tests', 'golden', 'golden.json'),
                            'ring6_sec80'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        rec = json.loads(p.stdout.decode().strip().splitlines()[-1])
    except Exception as e:  # the indicator must never break the bench line
        return {'error': repr(e)[:200]}
    return {'prove_s': rec['prove_ms'] / 1e3, 'verify_s': rec['verify_ms'] / 1e3, 'proofs_per_s': round(1e3 / max(rec['prove_ms'], 1), 3),
            'node': rec['node'], 'bytes_match_golden': bool(rec['sha256_ok']), 'verified': bool(rec['verified']), 'threads': 1,
            'workload': 'one proof, ring = 6 keys padded to 8, secLevel 80 (tests/golden/golden.json: ring6_sec80)',
            'note': 'approximation of `npm run bench` (V8 BigInt, this build\'s JS restatement); not the cpu_baseline value'}


def json_batch_rates(Z, ps):
    """zk_proofs_to_json_batch / zk_proofs_from_json_batch over the proofs `ps`: one thread, then every core this process may use;
    round-trip equality on the whole sample."""
    n = len(ps)
    off = (C.c_uint64 * (n + 1))()
    for i, p in enumerate(ps):
        off[i + 1] = off[i] + len(p)
    blob = (C.c_uint8 * max(1, off[n])).from_buffer_copy(b''.join(ps))
    cap = int(3.8 * off[n]) + 4096 * n
    out, toff, st = (C.c_uint8 * cap)(), (C.c_uint64 * (n + 1))(), (C.c_int32 * n)()
    back, poff = (C.c_uint8 * max(1, off[n]))(), (C.c_uint64 * (n + 1))()
    L = Z.lib()
    rec = {'proofs': n}
    for name, th in (('one_thread', 1), ('all_threads', host_cores())):
        best_w = best_r = None
        for _ in range(2):   # the first pass touches the output pages
            t0 = time.time()
            rc = L.zk_proofs_to_json_batch(n, blob, off, out, cap, toff, st, th)
            t1 = time.time()
            rc2 = L.zk_proofs_from_json_batch(n, out, toff, back, off[n], poff, st, th)
            t2 = time.time()
            assert rc == 0 and rc2 == 0, (rc, rc2)
            best_w = t1 - t0 if best_w is None else min(best_w, t1 - t0)
            best_r = t2 - t1 if best_r is None else min(best_r, t2 - t1)
        assert bytes(back) == bytes(blob) and list(poff) == list(off)
        rec[name] = {'threads': th, 'to_json_per_s': round(n / best_w, 1), 'from_json_per_s': round(n / best_r, 1)}
    rec['json_bytes_per_proof'] = int(toff[n]) // max(1, n)
    rec['to_json_per_s'], rec['from_json_per_s'], rec['threads'] = rec['all_threads']['to_json_per_s'], rec['all_threads']['from_json_per_s'], rec['all_threads']['threads']
    return rec


def pcie_bandwidth(dev, nbytes=2 << 30):
    """Plain page-locked copies of `nbytes` in each direction on this box (GB/s): the roofline of the host-pointer calls."""
    import torch
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    h = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    out = {}
    for name, dst, src in (('d2h_gbps', h, d), ('h2d_gbps', d, h)):
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.time()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        out[name] = round(nbytes / (time.time() - t0) / 1e9, 2)
    del d, h
    return out


def host_io_rates(Z, eng, args, B, sec, msg, sig, pk, which, seeds, dev, device_rate, device_vrate):
    """The SURVEY.md section 8(d) form of the metric: one zk_prove_batch / zk_verify_batch call on HOST buffers -- H2D of the
    inputs, proving, D2H of the binary proofs, and back in for the verifier -- with page-locked buffers from zk_host_alloc
    (per-chunk DMA under the kernels, tapered chunk plan) and, optionally, pageable ones.  ~169 KB per proof cross PCIe, so
    the roofline of these calls is the link: `pcie_frac` = achieved GB/s / this box's measured page-locked copy rate."""
    nb = min(args.host_io, B)
    hm, hs, hp, hw, hseed = msg[:32 * nb], sig[:64 * nb], pk[:64 * nb], which[:nb], seeds[:32 * nb]
    chunk, vchunk = min(args.host_io_chunk, nb), min(args.host_io_verify_chunk, nb)
    eng.set_chunk(chunk)
    eng.set_lanes(args.host_io_lanes)
    eng.prove_batch_host_raw(hm[:32 * 256], hs[:64 * 256], hp[:64 * 256], hw[:256], hseed[:32 * 256])   # warm-up
    host_io = {'proofs': nb, 'chunk': chunk, 'verify_chunk': vchunk, 'lanes': args.host_io_lanes, 'plan': 'staggered lanes, sliced PointAdd phase' if not args.host_io_uniform else 'uniform chunks',
               'note': 'PCIe-inclusive: one zk_prove_batch / zk_verify_batch call on host buffers (SURVEY.md 8(d)); `value` is the device-resident rate'}
    eng.set_host_taper(0 if args.host_io_uniform else 1)
    host_io['pcie'] = pcie_bandwidth(dev)
    t_pin = time.time()
    pin = Z.PinnedBuffer(int(nb * (304 + 336 * sec + 3392 * (sec // 2 + 4) + (4 * 72 + 96) * 20 + 32) + (64 << 20)))
    host_io['zk_host_alloc'] = {'bytes': pin.nbytes, 'seconds': round(time.time() - t_pin, 4)}
    bufs = [('pinned', pin)]
    if args.host_io_pageable:
        bufs.append(('pageable', (C.c_uint8 * pin.nbytes)()))
    for name, buf in bufs:
        best_p, best_v = None, None
        for _ in range(args.host_io_reps + 1):  # the first call allocates the engine's staging buffers (kept afterwards)
            eng.set_chunk(chunk)
            hdt, hout, hoff, hst = eng.prove_batch_host_raw(hm, hs, hp, hw, hseed, out=buf)
            eng.set_chunk(vchunk)
            vdt, vok, vst = eng.verify_batch_host_raw(hm, hout, hoff, nb)
            if _ > 0:
                best_p = hdt if best_p is None else min(best_p, hdt)
                best_v = vdt if best_v is None else min(best_v, vdt)
        nbytes = int(hoff[nb])
        rec = {'prove_s': round(best_p, 4), 'proofs_per_s': round(nb / best_p, 1), 'verify_s': round(best_v, 4),
               'verifies_per_s': round(nb / best_v, 1), 'out_bytes': nbytes,
               'd2h_gbps': round(nbytes / best_p / 1e9, 2), 'h2d_gbps': round(nbytes / best_v / 1e9, 2),
               'failed_proofs': sum(1 for x in hst if x != 0), 'accepted': sum(1 for x in vok if x == 1)}
        if name == 'pinned':
            rec['prove_frac_of_device_resident'] = round(nb / best_p / device_rate, 3) if device_rate else None
            rec['verify_frac_of_device_resident'] = round(nb / best_v / device_vrate, 3) if device_vrate else None
            rec['prove_pcie_frac'] = round(rec['d2h_gbps'] / host_io['pcie']['d2h_gbps'], 3)
            rec['verify_pcie_frac'] = round(rec['h2d_gbps'] / host_io['pcie']['h2d_gbps'], 3)
        host_io[name] = rec
    del hout
    if args.host_io_stream > 1:
        try:
            host_io['stream'] = host_io_stream(Z, eng, args, nb, chunk, vchunk, hm, hs, hp, hw, hseed, pin, device_rate, device_vrate, host_io['pcie'])
        except Exception as e:  # an auxiliary measurement must never cost the bench line
            host_io['stream'] = {'error': repr(e)[:300]}
    pin.free()
    eng.set_chunk(min(args.chunk, B))
    eng.set_lanes(args.lanes)
    return host_io


def host_io_stream(Z, eng, args, nb, chunk, vchunk, hm, hs, hp, hw, hseed, pin, device_rate, device_vrate, pcie):
    """Steady state of the host-pointer path: `--host-io-stream` batches of nb proofs back to back through zk_prove_submit /
    zk_prove_wait (two jobs in flight, two page-locked buffers in turn), then the same batches through zk_verify_submit /
    zk_verify_wait.  Rates over ALL batches, first submit to last wait (ramp-up and drain included), and over the batches after the
    first one (wait-to-wait).  Between calls nothing has to be hidden inside ONE call any more, so the plan may differ from the
    single-call one: --host-io-stream-configs / --host-io-stream-vconfigs list chunk:lanes:slice / chunk:lanes settings, the best is
    reported next to all of them."""
    nj, F = args.host_io_stream, max(2, args.host_io_stream_inflight)
    bufs = [pin] + [Z.PinnedBuffer(pin.nbytes) for _ in range(F - 1)]
    rec = {'batches': nj, 'proofs_per_batch': nb, 'in_flight': F, 'prove_configs': [], 'verify_configs': []}
    seeds = [hseed] + [rank_seeds(hseed, 7000 + k) for k in range(1, nj)]
    offs = [None] * nj
    def guarded(cfg, run):   # one configuration running out of memory (workspaces of more lanes) must not cost the others
        try:
            return run()
        except Exception as e:
            return {'config': cfg, 'error': repr(e)[:200]}

    def prove_cfg(cfg):
        c_, l_, s_ = (int(x) for x in cfg.split(':'))
        eng.set_chunk(min(c_, nb))
        eng.set_lanes(l_)
        eng.set_slice(s_)
        for _ in range(2):   # the first round allocates the jobs' staging buffers (kept by the context afterwards)
            t0 = time.time()
            tk, waits = [], []
            for k in range(min(F, nj)):
                tk.append(eng.prove_submit(hm, hs, hp, hw, seeds[k], bufs[k % F]))
            for k in range(nj):
                offs[k], st = eng.prove_wait(tk[k])
                waits.append(time.time())
                assert not any(st)
                if k + F < nj:   # job k's buffer is free again
                    tk.append(eng.prove_submit(hm, hs, hp, hw, seeds[k + F], bufs[(k + F) % F]))
        total_bytes = sum(int(o[nb]) for o in offs)
        return {'chunk': min(c_, nb), 'lanes': l_, 'slice': s_, 'proofs_per_s': round(nj * nb / (waits[-1] - t0), 1),
                'steady_proofs_per_s': round((nj - 1) * nb / (waits[-1] - waits[0]), 1), 'd2h_gbps': round(total_bytes / (waits[-1] - t0) / 1e9, 2),
                'seconds': round(waits[-1] - t0, 4)}

    for cfg in args.host_io_stream_configs.split(','):
        rec['prove_configs'].append(guarded(cfg, lambda: prove_cfg(cfg)))
    eng.set_slice(0)
    rec['prove'] = dict(max(rec['prove_configs'], key=lambda r: r.get('proofs_per_s', 0)))

    def verify_cfg(cfg):
        c_, l_ = (int(x) for x in cfg.split(':'))
        eng.set_chunk(min(c_, nb))
        eng.set_lanes(l_)
        Fv = max(2, min(F, args.host_io_stream_vinflight))
        src = lambda k: nj - 1 - (k % F) if nj >= F else k % nj   # the last F batches sit in the F buffers: verified in turn
        for _ in range(2):
            t0 = time.time()
            tk, waits, acc = [], [], 0
            for k in range(min(Fv, nj)):
                tk.append(eng.verify_submit(hm, bufs[src(k) % F], offs[src(k)], nb))
            for k in range(nj):
                ok, vst = eng.verify_wait(tk[k])
                waits.append(time.time())
                acc += sum(ok)
                if k + Fv < nj:
                    tk.append(eng.verify_submit(hm, bufs[src(k + Fv) % F], offs[src(k + Fv)], nb))
            assert acc == nj * nb, (acc, nj * nb)
        vbytes = sum(int(offs[src(k)][nb]) for k in range(nj))
        return {'chunk': min(c_, nb), 'lanes': l_, 'in_flight': Fv, 'verifies_per_s': round(nj * nb / (waits[-1] - t0), 1),
                'steady_verifies_per_s': round((nj - 1) * nb / (waits[-1] - waits[0]), 1),
                'h2d_gbps': round(vbytes / (waits[-1] - t0) / 1e9, 2), 'seconds': round(waits[-1] - t0, 4)}

    if offs[nj - 1] is not None:
        for cfg in args.host_io_stream_vconfigs.split(','):
            rec['verify_configs'].append(guarded(cfg, lambda: verify_cfg(cfg)))
    good_v = [r for r in rec['verify_configs'] if 'verifies_per_s' in r]
    rec['verify'] = dict(max(good_v, key=lambda r: r['verifies_per_s'])) if good_v else {}
    if 'proofs_per_s' not in rec['prove'] or not rec['verify']:
        for b_ in bufs[1:]:
            b_.free()
        return rec
    if device_rate:
        rec['prove']['frac_of_device_resident'] = round(rec['prove']['proofs_per_s'] / device_rate, 3)
        rec['prove']['steady_frac_of_device_resident'] = round(rec['prove']['steady_proofs_per_s'] / device_rate, 3)
    if device_vrate:
        rec['verify']['frac_of_device_resident'] = round(rec['verify']['verifies_per_s'] / device_vrate, 3)
    rec['prove']['pcie_frac'] = round(rec['prove']['d2h_gbps'] / pcie['d2h_gbps'], 3)
    rec['verify']['pcie_frac'] = round(rec['verify']['h2d_gbps'] / pcie['h2d_gbps'], 3)
    for b_ in bufs[1:]:
        b_.free()
    return rec


def run_pool_mode(args, Z):
    """`--pool`: ONE process, the --gpus devices of the node through the library's own zk_pool (csrc/api_pool.hip): zk_pool_set_ring
    uploads the ring once and broadcasts it device to device (RCCL over xGMI; peer copies when RCCL is unusable), every shard of a
    zk_pool_prove_batch / zk_pool_verify_batch call runs on its own host thread next to its device, the proofs land in one
    page-locked buffer whose per-shard regions sit on the shards' NUMA nodes.  Weak scaling: --batch proofs per device.  This
    is the SURVEY.md section 8(d) form of the metric (host buffers in, host buffers out): the rate is PCIe-inclusive by construction."""
    devs = [int(x) for x in args.pool_devices.split(',')] if args.pool_devices else list(range(args.gpus))
    G, Bg, nkeys, sec = len(devs), args.batch, args.ring, args.sec
    B = Bg * G
    pool = Z.Pool(devs)
    for i in range(G):
        e = pool.engine(i)
        e.set_comb_bits(args.comb_bits)
        e.set_chunk(min(args.host_io_chunk, Bg))
        e.set_lanes(args.host_io_lanes)
    e0 = pool.engine(0)
    nh, tg, th = e0.synth_params(args.seed)
    t0 = time.time()
    pool.set_params(nh, tg, th, sec)
    t_tab, tab_ms = time.time() - t0, pool.shard_ms()
    ring, msg, sig, pk, which, seeds = e0.synth_workload(args.seed, nkeys, Bg)
    t0 = time.time()
    transport = pool.set_ring(ring, nkeys)
    t_ring, ring_ms = time.time() - t0, pool.shard_ms()
    # every shard proves the same Bg statements under its own randomness (distinct proofs, identical work)
    msg_a, sig_a, pk_a, which_a = msg * G, sig * G, pk * G, list(which) * G
    seeds_a = b''.join(rank_seeds(seeds, i) for i in range(G))
    n_log2 = max(1, (nkeys - 1).bit_length())
    per_shard = int(Bg * (304 + 336 * sec + 3392 * (sec // 2 + 4) + (4 * 72 + 96) * n_log2 + 32) + (64 << 20))
    t0 = time.time()
    pin = Z.PinnedBuffer(per_shard * G, pool=pool)
    t_pin = time.time() - t0
    for _ in range(args.warmup):
        pool.prove_batch_raw(msg_a, sig_a, pk_a, which_a, seeds_a, pin, pin.nbytes)
    dts, shard = [], []
    for _ in range(args.steps):
        dt, off, ln, st = pool.prove_batch_raw(msg_a, sig_a, pk_a, which_a, seeds_a, pin, pin.nbytes)
        dts.append(dt)
        shard.append(pool.shard_ms())
    assert not any(st), [b for b in range(B) if st[b]][:8]
    nbytes = sum(ln)
    for i in range(G):
        pool.engine(i).set_chunk(min(args.host_io_verify_chunk, Bg))
    pool.verify_batch_raw(msg_a, pin, off, ln, B)   # warm-up (allocates the verifier workspaces)
    vdt, ok, vst = pool.verify_batch_raw(msg_a, pin, off, ln, B)
    vshard = pool.shard_ms()
    accepted = sum(ok)
    assert accepted == B and not any(vst), (accepted, B)
    # planted forgeries: one per shard, exactly those are rejected
    forged = [i * Bg + (i * 7919) % Bg for i in range(G)]
    for b in forged:
        pin.view[off[b] + ln[b] - 9] ^= 1
    _, ok2, _ = pool.verify_batch_raw(msg_a, pin, off, ln, B)
    assert [b for b in range(B) if not ok2[b]] == forged, 'planted forgeries not (exactly) rejected'
    for b in forged:
        pin.view[off[b] + ln[b] - 9] ^= 1
    # steady state: --host-io-stream batches back to back through zk_pool_prove_submit / zk_pool_prove_wait, F in flight per device
    stream = None
    if args.host_io_stream > 1:
        try:
            nj, F = args.host_io_stream, max(2, min(4, args.host_io_stream_inflight))
            c_, l_, s_ = (int(x) for x in args.host_io_stream_configs.split(',')[0].split(':'))
            for i in range(G):
                e = pool.engine(i)
                e.set_chunk(min(c_, Bg)), e.set_lanes(l_), e.set_slice(s_)
            bufs = [pin] + [Z.PinnedBuffer(pin.nbytes, pool=pool) for _ in range(F - 1)]
            for _ in range(2):
                t0 = time.time()
                tk = [pool.prove_submit(msg_a, sig_a, pk_a, which_a, seeds_a, bufs[k % F], pin.nbytes) for k in range(min(F, nj))]
                for k in range(nj):
                    so, sl, sst = pool.prove_wait(tk[k])
                    assert not any(sst)
                    if k + F < nj:
                        tk.append(pool.prove_submit(msg_a, sig_a, pk_a, which_a, seeds_a, bufs[(k + F) % F], pin.nbytes))
                dts_ = time.time() - t0
            stream = {'batches': nj, 'in_flight': F, 'chunk': min(c_, Bg), 'lanes': l_, 'slice': s_, 'proofs_per_s': round(nj * B / dts_, 1), 'seconds': round(dts_, 4)}
            for b_ in bufs[1:]:
                b_.free()
            for i in range(G):
                e = pool.engine(i)
                e.set_chunk(min(args.host_io_verify_chunk, Bg)), e.set_lanes(args.host_io_lanes), e.set_slice(0)
        except Exception as e:
            stream = {'error': repr(e)[:300]}
    cpu = None
    if not args.no_cpu_baseline:
        sample = args.cpu_sample or 4 * host_cores()
        cpu, oproofs = cpu_baseline(nh, tg, th, ring, nkeys, msg, sig, pk, which, seeds, sec, min(sample, Bg))
        for b in range(min(args.check, len(oproofs))):   # shard 0 proves under the synthetic seeds themselves
            assert bytes(pin.view[off[b]:off[b] + ln[b]]) == oproofs[b], 'GPU proof %d differs from the oracle' % b
        cpu['checked_bit_exact'] = min(args.check, len(oproofs))
    total = sum(dts)
    rate = B * args.steps / total
    line = {
        'metric': 'proveSignatureList proofs/sec (zk_pool: host buffers in, host buffers out)', 'value': round(rate, 2), 'unit': 'proofs/s',
        'value_is': 'PCIe-inclusive (SURVEY.md 8(d) form); this mode has no device-resident form -- bench.py without --pool reports that one',
        'value_pcie_inclusive': round(rate, 2), 'verify_pcie_inclusive': round(B / vdt, 2),
        'n_gpus': G, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(total * 1e3 / args.steps, 2), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u32 (9x30-bit limbs, 256/258-bit modular integers)', 'data': 'synthetic',
        'config': {'workload': 'batch=%d proofs per GPU per step (%d in total), ring=%d keys (n=%d), secLevel=%d, chunk=%d x %d lanes, comb=%d bits'
                               % (Bg, B, nkeys, n_log2, sec, min(args.host_io_chunk, Bg), args.host_io_lanes, args.comb_bits),
                   'parallelism': 'zk_pool: one process, devices %s, contiguous shards on host threads, ring %s' % (devs, transport)},
        'ring_transport': transport, 'set_ring_s': round(t_ring, 3), 'set_ring_shard_ms': ring_ms, 'set_params_s': round(t_tab, 3), 'set_params_shard_ms': tab_ms,
        'numa_nodes': [pool.numa_node(i) for i in range(G)], 'zk_pool_host_alloc': {'bytes': pin.nbytes, 'seconds': round(t_pin, 3)},
        'prove_shard_ms_per_step': shard, 'verify_shard_ms': vshard, 'proof_bytes_per_step': int(nbytes), 'failed_proofs': 0,
        'accepted': int(accepted), 'of': B, 'planted_forgeries_rejected': len(forged),
        'd2h_gbps_total': round(nbytes * args.steps / total / 1e9, 2), 'h2d_gbps_total': round(nbytes / vdt / 1e9, 2), 'cpu_baseline': cpu,
        'stream': stream, 'value_pcie_inclusive_steady': stream.get('proofs_per_s') if stream else None,
    }
    print(json.dumps(line))
    pin.free()
    pool.close()


def run_verify_mode(args, torch, Z, world, rank, local_rank, dev):
    """BASELINE.json configs[4]: verifySignatureList over `--batch` proofs IN TOTAL and a ring of `--ring` keys, sharded over the
    ranks (batch / world proofs each, no data-path collective).  2^20 proofs are ~177 GB and do not fit next to the tables, so
    a rank streams its shard in slabs: prove `--slab` proofs into HBM (untimed: the workload generator of this mode), verify
    them (timed, every call bracketed by synchronize), next slab.  All slabs prove the same `--slab` statements under fresh
    per-slab randomness -- distinct proofs, identical verifier work.  A step = one pass over the rank's shard."""
    import torch.distributed as dist
    total, nkeys, sec = args.batch, args.ring, args.sec
    assert total % world == 0, '--batch must be a multiple of the number of ranks'
    shard = total // world
    slab = min(args.slab, shard)
    nslabs = (shard + slab - 1) // slab
    eng = Z.Engine(local_rank)
    nh, tg, th = eng.synth_params(args.seed)
    eng.set_comb_bits(args.comb_bits)
    eng.set_params(nh, tg, th, sec)
    vchunk = min(args.verify_chunk or args.chunk, slab)
    vlanes = 1 if vchunk >= slab else (args.verify_lanes or args.lanes)   # one chunk per slab: a second lane would only hold memory
    eng.set_chunk(vchunk)
    eng.set_lanes(vlanes)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(args.seed, nkeys, slab)
    d_ring = torch.frombuffer(bytearray(ring), dtype=torch.uint8).to(dev)
    if world > 1:
        if rank != 0:
            d_ring.zero_()
        dist.broadcast(d_ring, src=0)
    torch.cuda.synchronize()
    t_ring = time.time()
    eng.set_ring_device(d_ring.data_ptr(), nkeys)
    t_ring = time.time() - t_ring
    tb = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_msg, d_sig, d_pk = tb(msg), tb(sig), tb(pk)
    d_which = torch.tensor(which, dtype=torch.int32, device=dev)
    n_log2 = max(1, (nkeys - 1).bit_length())
    cap = int(slab * (304 + 336 * sec + 3392 * (sec // 2 + 4) + (4 * 72 + 96) * n_log2 + 32) + (64 << 20))
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.empty(slab + 1, dtype=torch.int64, device=dev)
    d_st = torch.empty(slab, dtype=torch.int32, device=dev)
    d_ok = torch.empty(slab, dtype=torch.uint8, device=dev)
    d_vst = torch.empty(slab, dtype=torch.int32, device=dev)

    def one_pass(pass_no, plant):   # pass_no >= 0 keys the per-slab seeds; plant: forgeries in the first slab
        t_v, acc, t_p, nbytes = 0.0, 0, 0.0, 0
        for sl in range(nslabs):
            cnt = min(slab, shard - sl * slab)
            key = (rank * 1000003 + pass_no) * 4099 + sl + 1
            d_seeds = tb(rank_seeds(seeds[:32 * cnt], key))
            d_vseeds = tb(rank_seeds(seeds[:32 * cnt], key + (1 << 40)))
            t0 = time.time()
            eng.prove_batch_device(cnt, d_msg.data_ptr(), d_sig.data_ptr(), d_pk.data_ptr(), d_which.data_ptr(), d_seeds.data_ptr(),
                                   d_out.data_ptr(), cap, d_off.data_ptr(), d_st.data_ptr())
            torch.cuda.synchronize()
            t_p += time.time() - t0
            forged = []
            if plant and sl == 0 and cnt >= 64:   # planted forgeries in the first timed slab
                off = d_off[:cnt + 1].cpu().tolist()
                forged = [3, cnt // 2, cnt - 1]
                for b in forged:
                    d_out[off[b + 1] - 9] ^= 1   # a byte of the GK response zd: always caught
            torch.cuda.synchronize()
            t0 = time.time()
            eng.verify_batch_device(cnt, d_msg.data_ptr(), d_out.data_ptr(), d_off.data_ptr(), d_vseeds.data_ptr(), d_ok.data_ptr(), d_vst.data_ptr())
            torch.cuda.synchronize()
            t_v += time.time() - t0
            okc = d_ok[:cnt].cpu()
            if forged:
                assert [b for b in range(cnt) if not okc[b]] == forged, 'planted forgeries not (exactly) rejected'
                okc[forged] = 1
            assert int((d_st[:cnt] != 0).sum().item()) == 0
            acc += int(okc.sum().item())
            nbytes += int(d_off[cnt].item())
        return t_v, t_p, acc, nbytes

    for w in range(args.warmup):
        one_pass(w, False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    tv = tp = 0.0
    accepted = nbytes = 0
    for k in range(args.steps):
        a, b, c, d = one_pass(args.warmup + k, k == 0)
        tv, tp, accepted, nbytes = tv + a, tp + b, accepted + c, nbytes + d
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    eng.set_lanes(1)
    eng.verify_batch_device(slab, d_msg.data_ptr(), d_out.data_ptr(), d_off.data_ptr(), None, d_ok.data_ptr(), d_vst.data_ptr())
    _, vfam = eng.last_timing()
    if world > 1:
        t = torch.tensor([tv], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tv = float(t.item())
        t = torch.tensor([accepted], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        accepted = int(t.item())
    free_b, total_b = torch.cuda.mem_get_info()
    if rank == 0:
        ring_modmuls = nkeys * (n_log2 + 1)
        line = {
            'metric': 'verifySignatureList verifies/sec', 'value': round(total * args.steps / tv, 2), 'unit': 'verifies/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(tv * 1e3 / args.steps, 2),
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'u32 (9x30-bit limbs, 256/258-bit modular integers)',
            'data': 'synthetic',
            'config': {'workload': 'verifySignatureList batch=%d proofs in total, ring=%d keys (n=%d), secLevel=%d, %d proofs per rank streamed in %d slabs of %d, chunk=%d'
                                   % (total, nkeys, n_log2, sec, shard, nslabs, slab, vchunk),
                       'parallelism': 'proofs sharded per GPU; ring broadcast over RCCL at set-up' if world > 1 else 'single GPU'},
            'accepted': accepted, 'of': total * args.steps, 'planted_forgeries_rejected': 3 if slab >= 64 else 0,
            'timed_region': 'the zk_verify_batch_device calls only (proofs resident in HBM); generating the slabs took %.2f s per pass on rank 0' % (tp / max(1, args.steps)),
            'proof_bytes_per_pass': nbytes // max(1, args.steps), 'set_ring_s': round(t_ring, 3), 'hbm_used_gb': round((total_b - free_b) / 2**30, 1),
            'gpu_ms_by_family_per_slab': {k: round(v, 2) for k, v in sorted(vfam.items(), key=lambda kv: -kv[1])},
            'ring_fold': {'reference_modmuls_per_proof': ring_modmuls, 'note': 'gk.ts:239-250 does N*(n+1) modular multiplications per proof; the engine folds the ring in ratio form over table E (DESIGN.md section 4)'},
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=65536, help='proofs per GPU per step')
    ap.add_argument('--ring', type=int, default=65536, help='number of keys in the ring')
    ap.add_argument('--chunk', type=int, default=22016, help='proofs per pipeline pass of the prover (3 chunks of a 65536-proof step, one per lane)')
    ap.add_argument('--seed', type=int, default=2024)
    ap.add_argument('--sec', type=int, default=80)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=0, help='proofs in the CPU baseline sample (default 4 x cores, all of them diffed against the GPU output)')
    ap.add_argument('--comb-bits', type=int, default=DEFAULT_COMB_BITS, help='width of the Tom-256 fixed-base comb tables (8..24; 25, 26 = signed digits); 24 = 47 GB of tables')
    ap.add_argument('--lanes', type=int, default=3, help='chunks in flight on separate streams during the timed steps (1 = serial)')
    ap.add_argument('--verify-chunk', type=int, default=32768, help='proofs per pipeline pass of the verify half (0 = --chunk): the cross-proof sums like large chunks')
    ap.add_argument('--verify-lanes', type=int, default=2, help='chunks in flight of the verify half (0 = --lanes)')
    ap.add_argument('--roofline-steps', type=int, default=1, help='extra single-lane passes used only for per-kernel timings')
    ap.add_argument('--verify-steps', type=int, default=1, help='timed verifySignatureList passes over the produced proofs (0 = skip)')
    ap.add_argument('--host-io', type=int, default=1 << 30, help='proofs of the zk_prove_batch / zk_verify_batch calls on HOST buffers (PCIe-inclusive rates; default: the whole batch; 0 = skip)')
    ap.add_argument('--host-io-chunk', type=int, default=16384, help='proofs per chunk of the zk_prove_batch --host-io calls (the PointAdd phase of a chunk runs in slices of 4096 proofs, each followed by its D2H)')
    ap.add_argument('--host-io-verify-chunk', type=int, default=8192, help='proofs per chunk of the zk_verify_batch --host-io calls (H2D-bound: smaller chunks start earlier and leave less work behind the last transfer)')
    ap.add_argument('--host-io-lanes', type=int, default=2, help='chunks in flight during the host-buffer calls')
    ap.add_argument('--host-io-reps', type=int, default=2, help='timed repetitions of the host-buffer calls (best is reported)')
    ap.add_argument('--host-io-uniform', action='store_true', help='uniform chunks instead of the tapered plan')
    ap.add_argument('--host-io-stream', type=int, default=8, help='batches sent back to back through zk_prove_submit / zk_prove_wait (two in flight) for the steady-state PCIe-inclusive rate (0/1 = skip)')
    ap.add_argument('--host-io-stream-inflight', type=int, default=3, help='jobs kept in flight by the streamed measurement (2..4); 3 lets the stage-1 look-ahead always find a queued job')
    ap.add_argument('--host-io-stream-vinflight', type=int, default=2, help='jobs in flight of the streamed verify batches (link-bound: two keep the H2D stream full)')
    ap.add_argument('--host-io-stream-configs', default='22016:3:8192', help='chunk:lanes:slice settings of the streamed prove batches (comma-separated; the best is reported)')
    ap.add_argument('--host-io-stream-vconfigs', default='32768:2', help='chunk:lanes settings of the streamed verify batches')
    ap.add_argument('--device-stream', type=int, default=0, help='also time the steps as jobs kept in flight on the context (zk_prove_submit_device / zk_prove_wait, this many at a time, 2..4; every job its own output buffer): the pipeline does not drain between steps')
    ap.add_argument('--pool', action='store_true', help="ONE process, --gpus devices through the library's own zk_pool (RCCL ring broadcast, shards on host threads, page-locked host buffers): python bench.py --pool --gpus N")
    ap.add_argument('--pool-devices', default='', help='--pool: comma-separated device ids (default 0..gpus-1; a device may repeat: several contexts on one GPU)')
    ap.add_argument('--host-io-pageable', action='store_true', help='also measure ordinary (pageable) host buffers')
    ap.add_argument('--mode', choices=['prove', 'verify'], default='prove',
                    help="verify: BASELINE configs[4] -- --batch proofs IN TOTAL over --ring keys, sharded over the ranks, generated and verified in streamed slabs")
    ap.add_argument('--slab', type=int, default=32768, help='--mode verify: proofs generated and verified per slab (one --verify-chunk by default: 176 k verifies/s at ring 2^20 against 141 k with slabs and chunks of 8192)')
    ap.add_argument('--json-sample', type=int, default=2048, help='proofs converted to the JSON wire format and back by the batch converters, on one host thread and on all of them (toJson / fromJson of the reference bench; 0 = skip)')
    ap.add_argument('--check', type=int, default=1 << 30, help='proofs of the last step diffed against the oracle on rank 0 (at most the CPU sample)')
    args = ap.parse_args()

    import zkp_ecdsa_amd as Z
    if args.pool:
        return run_pool_mode(args, Z)
    import torch

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # ZK_BENCH_ONE_DEVICE=1: every rank uses cuda:0 and the process group runs on gloo (RCCL refuses two ranks on one GPU) -- a
    # smoke test of the N > 1 code path on a one-GPU box (tools/smoke_multirank.sh); never set by the driver
    one_device = os.environ.get('ZK_BENCH_ONE_DEVICE') == '1'
    if one_device:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('gloo' if one_device else 'nccl', rank=rank, world_size=world)
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    if args.mode == 'verify':
        return run_verify_mode(args, torch, Z, world, rank, local_rank, dev)
    B, nkeys, sec = args.batch, args.ring, args.sec
    eng = Z.Engine(local_rank)
    nh, tg, th = eng.synth_params(args.seed)
    eng.set_comb_bits(args.comb_bits)
    t_tab = time.time()
    eng.set_params(nh, tg, th, sec)       # builds the fixed-base tables (one-time, not part of a step)
    t_tab = time.time() - t_tab
    eng.set_chunk(min(args.chunk, B))
    eng.set_lanes(args.lanes)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(args.seed, nkeys, B)

    # the key ring travels rank 0 -> all ranks over RCCL (xGMI); everything else is generated locally from the seed
    d_ring = torch.frombuffer(bytearray(ring), dtype=torch.uint8).to(dev)
    if world > 1:
        if rank != 0:
            d_ring.zero_()
        dist.broadcast(d_ring, src=0)
    torch.cuda.synchronize()
    eng.set_ring_device(d_ring.data_ptr(), nkeys)

    my_seeds = rank_seeds(seeds, rank)
    tb = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_msg, d_sig, d_pk, d_seeds = tb(msg), tb(sig), tb(pk), tb(my_seeds)
    d_which = torch.tensor(which, dtype=torch.int32, device=dev)
    cap = eng.proof_max_size() * B
    # expected size is ~(sec/2) long reps per proof; keep the worst-case bound only when it is small
    exp_cap = int(B * (304 + 336 * sec + 3392 * (sec // 2 + 4) + (4 * 72 + 96) * 20 + 32) + (64 << 20))
    cap = min(cap, max(exp_cap, 1 << 20))
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
    d_st = torch.empty(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step():
        eng.prove_batch_device(B, d_msg.data_ptr(), d_sig.data_ptr(), d_pk.data_ptr(), d_which.data_ptr(), d_seeds.data_ptr(),
                               d_out.data_ptr(), cap, d_off.data_ptr(), d_st.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.time() - t0
    # the same steps as jobs in flight on the context: submit(0) .. submit(n-1); wait(0); submit(n); wait(1); ...
    dstream = None
    if args.device_stream >= 2:
        nf = min(args.device_stream, 4)
        outs = [(d_out, d_off, d_st)] + [(torch.empty(cap, dtype=torch.uint8, device=dev), torch.empty(B + 1, dtype=torch.int64, device=dev),
                                          torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(nf - 1)]
        ref_sum = int(d_out[:int(d_off[B].item())].to(torch.int64).sum().item())
        torch.cuda.synchronize()

        def submit(k):
            o, f, st_ = outs[k % nf]
            return eng.prove_submit_device(B, d_msg.data_ptr(), d_sig.data_ptr(), d_pk.data_ptr(), d_which.data_ptr(), d_seeds.data_ptr(), o.data_ptr(), cap, f.data_ptr(), st_.data_ptr())

        def run(K):
            q = [submit(k) for k in range(min(nf, K))]
            for k in range(K):
                eng.prove_wait(q[k])
                if k + nf < K:
                    q.append(submit(k + nf))
        run(max(args.warmup, nf))
        barrier()
        K = max(args.steps, 2 * nf)
        ts0 = time.time()
        run(K)
        barrier()
        sdt = time.time() - ts0
        if world > 1:
            t = torch.tensor([sdt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sdt = float(t.item())
        same = all(int(o[:int(f[B].item())].to(torch.int64).sum().item()) == ref_sum and int((st_ != 0).sum().item()) == 0 for o, f, st_ in outs)
        dstream = {'jobs_in_flight': nf, 'steps': K, 'value': round(world * B * K / sdt, 2), 'ms_per_step': round(1000 * sdt / K, 2), 'outputs_equal_sync_call': bool(same)}
        del outs
    # per-kernel timings for the roofline: ONE extra pass with strictly serial kernels (single lane), HIP events on the
    # engine's stream around every launch.  In the timed steps above the chunks of a step overlap on their lanes' streams, which makes a
    # single kernel's duration ill-defined; this pass is not part of `value`.
    fam = {}
    gpu_ms = 0.0
    eng.set_lanes(1)
    for _ in range(args.roofline_steps):
        step()
        tot, f = eng.last_timing()
        gpu_ms += tot
        for k, v in f.items():
            fam[k] = fam.get(k, 0.0) + v
    eng.set_lanes(args.lanes)
    torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st = d_st.cpu()
    nbad = int((st != 0).sum().item())
    off = d_off.cpu()
    total_bytes = int(off[B].item())

    # ---- verifySignatureList over the proofs just produced (the "+ verify/sec" half of the metric); proofs stay in HBM
    verify = None
    if args.verify_steps > 0:
        d_ok = torch.empty(B, dtype=torch.uint8, device=dev)
        d_vst = torch.empty(B, dtype=torch.int32, device=dev)
        d_vseeds = tb(rank_seeds(seeds, rank + 1000))

        vchunk, vlanes = min(args.verify_chunk or args.chunk, B), args.verify_lanes or args.lanes
        eng.set_chunk(vchunk)
        eng.set_lanes(vlanes)

        def vstep():
            eng.verify_batch_device(B, d_msg.data_ptr(), d_out.data_ptr(), d_off.data_ptr(), d_vseeds.data_ptr(), d_ok.data_ptr(), d_vst.data_ptr())
        vstep()  # warm-up (allocates the verifier workspace)
        barrier()
        tv0 = time.time()
        for _ in range(args.verify_steps):
            vstep()
        barrier()
        vdt = time.time() - tv0
        # per-kernel-family GPU time from one extra serial (single-lane) pass, like the prover's roofline pass
        eng.set_lanes(1)
        vstep()
        _, vfam = eng.last_timing()
        eng.set_lanes(args.lanes)
        eng.set_chunk(min(args.chunk, B))
        if world > 1:
            t = torch.tensor([vdt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            vdt = float(t.item())
        n_ok = int(d_ok.sum().item())
        verify = {'value': round(world * B * args.verify_steps / vdt, 2), 'unit': 'verifies/s', 'steps': args.verify_steps,
                  'ms_per_step': round(vdt * 1e3 / args.verify_steps, 2), 'accepted': n_ok, 'of': B, 'chunk': vchunk, 'lanes': vlanes,
                  'gpu_ms_by_family_per_step': {k: round(v, 2) for k, v in sorted(vfam.items(), key=lambda kv: -kv[1])},
                  'gpu_ms_note': 'serial single-lane pass; the timed passes overlap %d chunks on %d streams' % (vlanes, vlanes)}

    free_b, total_b = torch.cuda.mem_get_info()
    hbm_used = total_b - free_b
    if rank == 0:
        n_log2 = max(1, (nkeys - 1).bit_length())
        # --- roofline of the dominant kernel, from HIP events recorded on the engine's stream around every launch
        off_l = off.tolist()
        zeros_total = sum(((off_l[i + 1] - off_l[i]) - (304 + 336 * sec + (4 * 72 + 96) * n_log2 + 32)) // 3392 for i in range(B) if off_l[i + 1] > off_l[i])
        commits_per_step = B * (2 + 2 * sec) + zeros_total * 34 + B * 4 * n_log2
        tom_ms = fam.get('tom_commit', 0.0) / max(1, args.roofline_steps)
        launches_per_step = 4 * ((B + eng_chunk(args, B) - 1) // eng_chunk(args, B))   # lists A, B (unpaired + paired slots), C
        # executed additions: 2 * nwin per commitment, except, per PointAdd item, the 9 pairs that share v*g (3 * nwin per pair) and
        # C4 of pi8, whose value is i7 * i8 = 1: its g-windows above the first are skipped wave-wide on unsigned combs (k_tom.hip)
        nwin = (256 + args.comb_bits - 1) // args.comb_bits
        skipped = nwin - 1 if args.comb_bits <= 24 else 0
        adds = (commits_per_step - 34 * zeros_total) * 2 * nwin + zeros_total * ((16 * 2 + 9 * 3) * nwin - skipped)
        # 8 modmuls per addition; the first one of a comb is 1 (identity + entry), the last one 7 (no T coordinate)
        modmuls = adds * 8 - (commits_per_step - 34 * zeros_total) * 8 - zeros_total * (16 * 8 + 9 * 9)
        modmuls_per_commit = round(modmuls / commits_per_step, 2)
        pmc_bytes = TOM_COMMIT_PMC_BYTES.get(args.comb_bits)
        macs = modmuls * MACS_PER_MODMUL
        achieved_tmacs = macs / (tom_ms * 1e-3) / 1e12 if tom_ms > 0 else 0.0
        hbm_gbps = commits_per_step * TOM_COMMIT_BYTES / (tom_ms * 1e-3) / 1e9 if tom_ms > 0 else 0.0
        wt, wq, wring = nominal_modmuls(n_log2)
        roofline = {
            'bound': 'valu_int32',
            'kernel': 'k_tom_commit',
            'achieved': round(achieved_tmacs, 3), 'peak': VALU_MAD_PEAK_TOPS, 'unit': 'T multiplier-instr/s (v_mad_u64_u32 + v_mul_lo_u32 lane-ops, peak measured by tools/valu_peak.hip)',
            'frac': round(achieved_tmacs / VALU_MAD_PEAK_TOPS, 4),
            'frac_of_rate_at_kernel_ilp': round(achieved_tmacs / VALU_MAD_8CHAIN_TOPS, 4),
            'rate_at_kernel_ilp': {'value': VALU_MAD_8CHAIN_TOPS, 'note': 'v_mad_u64_u32 microbenchmark at 8 independent chains per SIMD (4 per wave x 2 waves: what 212 VGPRs allow); '
                                   'the kernel issues 1.39 VALU instructions per multiplier instruction on top (profiles/r03_valu_peak_microbench.txt, DESIGN.md section 8)'},
            'traffic': int(commits_per_step / max(1, launches_per_step) * pmc_bytes) if pmc_bytes else None,
            'traffic_note': ('bytes per launch = units per launch x %d B (FETCH_SIZE + WRITE_SIZE per commitment, separate rocprofv3 --pmc passes, '
                             'profiles/r03_pmc_summary.txt); table gathers, not the 180 algorithmic bytes, dominate' % pmc_bytes) if pmc_bytes
                            else 'no PMC pass recorded for this comb width',
            'valu_active_per_wave': TOM_COMMIT_VALU_ACTIVE_PER_WAVE.get(args.comb_bits),
            'pmc_source': PMC_SOURCE,
            'comb_bits': args.comb_bits,
            'avg_launch_ms': round(tom_ms / max(1, launches_per_step), 3),
            'launches_per_step': launches_per_step,
            'units_per_step': commits_per_step,
            'executed_modmuls_per_unit': modmuls_per_commit, 'nominal_modmuls_per_unit': TOM_COMMIT_NOMINAL,
            'hbm': {'achieved': round(hbm_gbps, 2), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': round(hbm_gbps / HBM_PEAK_GBPS, 5),
                    'algorithmic_bytes_per_unit': TOM_COMMIT_BYTES},
            'share_of_gpu_time': round(fam.get('tom_commit', 0.0) / gpu_ms, 3) if gpu_ms else None,
            'nominal_modmuls_per_proof': {'F_t': wt, 'F_q_ec': wq, 'F_q_ring': wring},
        }
        cpu = None
        if not args.no_cpu_baseline:   # rank 0, at every N (the other ranks wait at the final barrier)
            sample = args.cpu_sample or 4 * host_cores()
            cpu, oproofs = cpu_baseline(nh, tg, th, ring, nkeys, msg, sig, pk, which, seeds, sec, min(sample, B))
            # spot-check: the first proofs of the last step against the oracle, byte for byte
            ncheck = min(args.check, len(oproofs))
            raw = d_out[:int(off[ncheck].item())].cpu().numpy().tobytes()
            for b in range(ncheck):
                assert raw[int(off[b]):int(off[b + 1])] == oproofs[b], 'GPU proof %d differs from the oracle' % b
            cpu['checked_bit_exact'] = ncheck
            cpu['v8_bigint'] = v8_bigint_indicator()
        json_rates = None
        if args.json_sample > 0:   # bench/zkpAttestList.bench.ts:63-68 (toJson / fromJson): the batch converters of the C ABI on every host core
            nj = min(args.json_sample, B)
            raw = d_out[:int(off[nj].item())].cpu().numpy().tobytes()
            ps = [raw[int(off[b]):int(off[b + 1])] for b in range(nj)]
            json_rates = json_batch_rates(Z, ps)
        host_io = None
        if args.host_io > 0 and world == 1:
            try:
                host_io = host_io_rates(Z, eng, args, B, sec, msg, sig, pk, which, seeds, dev, B * args.steps / dt,
                                        verify['value'] / world if verify else None)
            except Exception as e:  # an auxiliary measurement must never cost the bench line
                host_io = {'error': repr(e)[:300]}
            eng.set_chunk(min(args.chunk, B))
        ms_per_step = dt * 1e3 / args.steps
        line = {
            'metric': 'proveSignatureList proofs/sec', 'value': round(world * B * args.steps / dt, 2), 'unit': 'proofs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u32 (9x30-bit limbs, 256/258-bit modular integers)',
            'data': 'synthetic',
            'config': {'workload': 'batch=%d proofs per GPU per step, ring=%d keys (n=%d), secLevel=%d, P-256 + Tom-256, chunk=%d x %d lanes, comb=%d bits%s'
                                   % (B, nkeys, n_log2, sec, eng_chunk(args, B), args.lanes, args.comb_bits,
                                      ', per-key tables of the ring' if n_log2 <= 16 and os.environ.get('ZKATTEST_KEYTAB', '1') != '0' else ''),
                       'parallelism': 'proofs sharded per GPU; ring broadcast over RCCL at set-up' if world > 1 else 'single GPU'},
            'set_params_s': round(t_tab, 3),
            'hbm_used_gb': round(hbm_used / 2**30, 1),   # tables + both lanes' prover and verifier workspaces + this step's proofs
            'proof_bytes_per_step': total_bytes, 'failed_proofs': nbad,
            'gpu_ms_by_family_per_step': {k: round(v / max(1, args.roofline_steps), 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])},
            'gpu_ms_note': 'serial single-lane pass (sum = %.1f ms); the timed steps overlap %d chunks on %d streams' % (gpu_ms / max(1, args.roofline_steps), args.lanes, args.lanes),
            'roofline': roofline, 'cpu_baseline': cpu, 'verify': verify, 'host_io': host_io, 'json': json_rates,
        }
        if dstream:
            line['device_stream'] = dstream
        if host_io and 'pinned' in host_io:   # the SURVEY.md 8(d) form of the metric, next to the device-resident `value`
            line['value_pcie_inclusive'] = host_io['pinned']['proofs_per_s']
            line['verify_pcie_inclusive'] = host_io['pinned']['verifies_per_s']
            line['value_note'] = ('value: inputs and proofs resident in HBM (bench contract); value_pcie_inclusive: one zk_prove_batch call on '
                                  'page-locked host buffers incl. H2D of the inputs and D2H of %.2f GB of proofs; value_pcie_inclusive_steady: %d such '
                                  'batches back to back through zk_prove_submit / zk_prove_wait, two in flight (first submit to last wait)'
                                  % (host_io['pinned']['out_bytes'] / 1e9, args.host_io_stream))
            st = host_io.get('stream') or {}
            if 'proofs_per_s' in (st.get('prove') or {}):
                line['value_pcie_inclusive_steady'] = st['prove']['proofs_per_s']
            if 'verifies_per_s' in (st.get('verify') or {}):
                line['verify_pcie_inclusive_steady'] = st['verify']['verifies_per_s']
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def eng_chunk(args, B):
    return min(args.chunk, B)


if __name__ == '__main__':
    main()
