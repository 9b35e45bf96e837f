"""The auxiliary measurements and alternative modes of bench.py (the default path -- the timed steps, the roofline, the JSON line --
stays in bench.py): JSON converter rates, the PCIe-inclusive host-pointer calls (one call, streamed batches), small-batch latency,
`--pool` (one process, the devices through zk_pool) and `--mode verify` (BASELINE configs[4])."""
import ctypes as C
import json
import os
import time

from bench_common import cpu_baseline, host_cores, rank_seeds

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
            import sys
            print('host_io stream failed: %r' % (e,), file=sys.stderr)
            host_io['stream'] = {'error': repr(e)[:300]}
    if args.host_io_stream > 1 and args.host_io_packed and 'proofs_per_s' in (host_io.get('stream', {}).get('prove') or {}):
        # the same streamed batches in the packed wire layout (ZKA1P, zk_ctx_set_wire): 5.3 % fewer bytes across the link in both directions
        try:
            eng.set_wire(True)
            sp = host_io_stream(Z, eng, args, nb, chunk, vchunk, hm, hs, hp, hw, hseed, pin, device_rate, device_vrate, host_io['pcie'])
            host_io['stream_packed'] = {'wire': 'ZKA1P (33-byte Tom coordinates)', 'prove': sp.get('prove'), 'verify': sp.get('verify')}
        except Exception as e:
            host_io['stream_packed'] = {'error': repr(e)[:300]}
        eng.set_wire(False)
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
    pending = []   # tickets submitted and not yet waited for: a configuration that fails must not leave jobs queued on the context (every setter would refuse)

    def guarded(cfg, run):   # one configuration running out of memory (workspaces of more lanes) must not cost the others
        try:
            return run()
        except Exception as e:
            import sys
            print('host_io_stream: configuration %s failed: %r' % (cfg, e), file=sys.stderr)
            for kind, t in pending:
                try:
                    (eng.prove_wait if kind == 'p' else eng.verify_wait)(t)
                except Exception:
                    pass
            del pending[:]
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
                pending.append(('p', tk[-1]))
            for k in range(nj):
                pending.pop(0)
                offs[k], st = eng.prove_wait(tk[k])
                waits.append(time.time())
                assert not any(st)
                if k + F < nj:   # job k's buffer is free again
                    tk.append(eng.prove_submit(hm, hs, hp, hw, seeds[k + F], bufs[(k + F) % F]))
                    pending.append(('p', tk[-1]))
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
                pending.append(('v', tk[-1]))
            for k in range(nj):
                pending.pop(0)
                ok, vst = eng.verify_wait(tk[k])
                waits.append(time.time())
                acc += sum(ok)
                if k + Fv < nj:
                    tk.append(eng.verify_submit(hm, bufs[src(k + Fv) % F], offs[src(k + Fv)], nb))
                    pending.append(('v', tk[-1]))
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


def latency_table(Z, eng, args, nh, tg, th, sec, ring, nkeys, msg, sig, pk, which, seeds, sizes=(1, 8, 64, 512, 4096), small_ring=1024):
    """What ONE call costs at the sizes the reference's own API has (src/zkpAttestList.ts:104-145 proves one signature per call): wall time of
    zk_prove_batch / zk_verify_batch on host buffers (page-locked output; inputs, proofs and verdicts cross PCIe) for B = 1 .. 4096 proofs,
    on the bench's ring and on a ring of `small_ring` keys.  Median of a few calls after one untimed call per size (the first call of a new
    chunk size re-carves the workspace)."""
    out = {'unit': 'ms per call (median)', 'path': 'zk_prove_batch / zk_verify_batch on host pointers, page-locked proof buffer', 'rings': {}}
    pin = Z.PinnedBuffer(int(max(sizes) * (304 + 336 * sec + 3392 * (sec // 2 + 4) + 384 * 20 + 32) + (64 << 20)))
    rings = [(nkeys, ring, msg, sig, pk, which, seeds)]
    if small_ring and small_ring != nkeys:
        r2, m2, s2, p2, w2, sd2 = eng.synth_workload(args.seed + 1, small_ring, min(max(sizes), small_ring))   # proof b's key sits at ring slot b mod n_keys
        rings.append((small_ring, r2, m2, s2, p2, w2, sd2))
    lanes0 = args.lanes
    eng.set_lanes(1)
    eng.set_timing(2)   # the default: no per-family events in a call of <= 8 192 proofs (0.15-0.45 ms of a small call)
    for nk, rg, m, s_, p, w, sd in rings:
        if nk != nkeys:
            eng.set_ring(rg, nk)
        rows = {}
        for B in sizes:
            if B > len(w) or B > nk:   # the synthetic workload plants key b at ring slot b mod n_keys: more proofs than keys would overwrite signers
                continue
            eng.set_chunk(B)
            a = (m[:32 * B], s_[:64 * B], p[:64 * B], w[:B], sd[:32 * B])
            reps = 7 if B <= 64 else 3
            tp, tv = [], []
            for k in range(reps + 1):
                dt, hout, hoff, hst = eng.prove_batch_host_raw(*a, out=pin)
                assert not any(hst)
                vdt, vok, vst = eng.verify_batch_host_raw(a[0], hout, hoff, B)
                assert sum(vok) == B
                if k:
                    tp.append(dt), tv.append(vdt)
            tp.sort(), tv.sort()
            rows[str(B)] = {'prove_ms': round(1e3 * tp[len(tp) // 2], 3), 'verify_ms': round(1e3 * tv[len(tv) // 2], 3),
                            'proofs_per_s': round(B / tp[len(tp) // 2], 1), 'verifies_per_s': round(B / tv[len(tv) // 2], 1)}
        out['rings'][str(nk)] = rows
    if len(rings) > 1:
        eng.set_ring(ring, nkeys)
    eng.set_lanes(lanes0)
    eng.set_timing(1)
    pin.free()
    return out


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
    # the same call with the proofs left in each device's HBM (zk_pool_prove_batch_device): what the GPUs and their host threads do without the
    # node's host memory in the path -- on N devices the ratio to `value` separates compute scaling from host-memory scaling
    dev_out = None
    try:
        # the device-resident shape of bench.py's `value` (round 4 measured this with the verify pass's 8 192-proof chunks on two lanes still set: the "8 %
        # of the pool" of that round's review was the chunk size, not the pool)
        for i in range(G):
            e = pool.engine(i)
            e.set_chunk(min(args.chunk, Bg)), e.set_lanes(args.lanes), e.set_slice(0)
        dbuf = [pool.device_alloc(i, per_shard) for i in range(G)]
        pool.prove_batch_device_out(msg_a, sig_a, pk_a, which_a, seeds_a, dbuf, [per_shard] * G)
        ddts = []
        for _ in range(args.steps):
            ddt, doff, dln, dst = pool.prove_batch_device_out(msg_a, sig_a, pk_a, which_a, seeds_a, dbuf, [per_shard] * G)
            ddts.append(ddt)
        assert not any(dst) and list(dln) == list(ln)
        dev_out = {'proofs_per_s': round(B * args.steps / sum(ddts), 1), 'ms_per_step': round(sum(ddts) * 1e3 / args.steps, 2), 'shard_ms': pool.shard_ms(),
                   'chunk': min(args.chunk, Bg), 'lanes': args.lanes}
        for i in range(G):
            pool.device_free(i, dbuf[i])
    except Exception as e:
        dev_out = {'error': repr(e)[:300]}
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
        'device_resident_output': dev_out, 'ring_transport_note': pool.last_error() if transport != 'rccl' and G > 1 else None,
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
    eng.set_timing(1)
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


