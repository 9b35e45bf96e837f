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

from bench_common import (DEFAULT_COMB_BITS, EXP_ADD_MACS, EXP_KT_ADDS, EXP_RTAB_ADDS, HBM_PEAK_GBPS, MACS_PER_MODMUL, PMC_SOURCE, TOM_COMMIT_BYTES, TOM_COMMIT_NOMINAL, TOM_COMMIT_PMC_BYTES,  # noqa: E402,F401
                          TOM_COMMIT_VALU_ACTIVE_PER_WAVE, VALU_MAD_8CHAIN_TOPS, VALU_MAD_PEAK_TOPS, VERIFY_PMC_SOURCE, VERIFY_WHOLE_STEP_SIMD_BUSY, cpu_baseline, host_cores, nominal_modmuls, rank_seeds,
                          tom_commit_modmuls, v8_bigint_indicator)
from bench_modes import host_io_rates, json_batch_rates, latency_table, run_pool_mode, run_verify_mode  # noqa: E402


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
    ap.add_argument('--verify-steps', type=int, default=-1, help='timed verifySignatureList passes over the produced proofs, each timed on its own (default: max(5, --steps); 0 = skip)')
    ap.add_argument('--verify-warmup', type=int, default=1, help='untimed verify passes before the timed ones (the first allocates the verifier workspace)')
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
    ap.add_argument('--host-io-packed', type=int, default=1, help='also stream the batches in the packed wire layout ZKA1P (0 = skip)')
    ap.add_argument('--host-io-pageable', action='store_true', help='also measure ordinary (pageable) host buffers')
    ap.add_argument('--mode', choices=['prove', 'verify'], default='prove',
                    help="verify: BASELINE configs[4] -- --batch proofs IN TOTAL over --ring keys, sharded over the ranks, generated and verified in streamed slabs")
    ap.add_argument('--slab', type=int, default=32768, help='--mode verify: proofs generated and verified per slab (one --verify-chunk by default: 176 k verifies/s at ring 2^20 against 141 k with slabs and chunks of 8192)')
    ap.add_argument('--json-sample', type=int, default=2048, help='proofs converted to the JSON wire format and back by the batch converters, on one host thread and on all of them (toJson / fromJson of the reference bench; 0 = skip)')
    ap.add_argument('--latency', type=int, default=1, help='small-batch latency table (B = 1 .. 4096 proofs per zk_prove_batch / zk_verify_batch call, two ring sizes) in the bench line; 0 = skip')
    ap.add_argument('--check', type=int, default=1 << 30, help='proofs of the last step diffed against the oracle on rank 0 (at most the CPU sample)')
    args = ap.parse_args()

    if args.verify_steps < 0:
        args.verify_steps = max(5, args.steps)
    # A bare `python3 bench.py --gpus N` (N > 1, no launcher in the environment) starts its own ranks: the documented launch line, one rank per GPU.
    if args.gpus > 1 and not args.pool and 'WORLD_SIZE' not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        print('bench.py: --gpus %d without WORLD_SIZE: re-executing as `%s`' % (args.gpus, ' '.join(cmd[1:])), file=sys.stderr, flush=True)
        os.execv(sys.executable, cmd)
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
    assert world == args.gpus, 'WORLD_SIZE=%d but --gpus %d: launch with torch.distributed.run --nproc-per-node %d (or without a launcher: bench.py starts its own ranks)' % (world, args.gpus, args.gpus)
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    if args.mode == 'verify':
        return run_verify_mode(args, torch, Z, world, rank, local_rank, dev)
    B, nkeys, sec = args.batch, args.ring, args.sec
    eng = Z.Engine(local_rank)
    eng.set_timing(1)   # the per-family tables below come from HIP events around every kernel family; the latency table switches them off (the product's default for small calls)
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
    kt_count = eng.test_counter(1)
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
        for _ in range(max(1, args.verify_warmup)):   # warm-up (the first call allocates the verifier workspace)
            vstep()
        barrier()
        terms0 = eng.test_counter(2)
        tv0 = time.time()
        vtimes = []   # every call returns with its verdicts in place (zk_verify_batch_device is synchronous): a step is one call
        for _ in range(args.verify_steps):
            ts = time.time()
            vstep()
            vtimes.append(time.time() - ts)
        barrier()
        vdt = time.time() - tv0
        msm_terms = (eng.test_counter(2) - terms0) // max(1, args.verify_steps)   # live terms of one step's batched Tom-256 check
        # per-kernel-family GPU time from one extra serial (single-lane) pass, like the prover's roofline pass
        eng.set_lanes(1)
        vstep()
        _, vfam = eng.last_timing()
        vwall = eng.last_wall_ms()   # first start -> last end of that pass (the families' sum counts v_msm_p256 twice: it runs beside v_msm_tom on an auxiliary stream)
        eng.set_lanes(args.lanes)
        eng.set_chunk(min(args.chunk, B))
        if world > 1:
            t = torch.tensor([vdt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            vdt = float(t.item())
        n_ok = int(d_ok.sum().item())
        vs = sorted(vtimes)
        verify = {'value': round(world * B * args.verify_steps / vdt, 2), 'unit': 'verifies/s', 'steps': args.verify_steps, 'warmup': max(1, args.verify_warmup),
                  'ms_per_step': round(vdt * 1e3 / args.verify_steps, 2), 'min_ms': round(vs[0] * 1e3, 2), 'median_ms': round(vs[len(vs) // 2] * 1e3, 2), 'max_ms': round(vs[-1] * 1e3, 2),
                  'accepted': n_ok, 'of': B, 'chunk': vchunk, 'lanes': vlanes, 'msm_live_terms': int(msm_terms),
                  'gpu_ms_by_family_per_step': {k: round(v, 2) for k, v in sorted(vfam.items(), key=lambda kv: -kv[1])},
                  'gpu_ms_single_lane_wall': round(vwall, 2),
                  'gpu_ms_note': 'single-lane pass (chunks one after the other; gpu_ms_single_lane_wall = its first kernel start to last kernel end); the timed passes overlap '
                                 '%d chunks on %d streams' % (vlanes, vlanes)}

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
                             'profiles/r05_pmc_summary.txt); table gathers, not the 180 algorithmic bytes, dominate' % pmc_bytes) if pmc_bytes
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
        # ---- the two next-largest arithmetic kernels, priced the same way (multiplier instructions of the shipped ISA / measured peak)
        others = {}
        exp_ms = fam.get('p256_exp_commit', 0.0) / max(1, args.roofline_steps)
        if exp_ms > 0:
            lanes = B * (sec + 1)
            adds = EXP_KT_ADDS if kt_count else EXP_RTAB_ADDS
            ops = lanes * adds * EXP_ADD_MACS
            others['k_exp_commit_kt' if kt_count else 'k_exp_commit'] = {
                'units': lanes, 'unit': 'T_i = alpha_i R, A_i = T_i + r_i h (one lane each; exp.ts:144-149)', 'mixed_additions_per_unit': adds,
                'multiplier_instr_per_addition': EXP_ADD_MACS, 'ms_per_step': round(exp_ms, 2), 'achieved': round(ops / (exp_ms * 1e-3) / 1e12, 3),
                'peak': VALU_MAD_PEAK_TOPS, 'frac': round(ops / (exp_ms * 1e-3) / 1e12 / VALU_MAD_PEAK_TOPS, 4),
                'note': 'ISA count of the loop body (tools/isa_blocks.py: 1872 v_mad_u64_u32 + 99 v_mul_lo_u32 per mixed complete addition of 11 products); 13 + 33 + 13 gathered '
                        'table entries per lane minus the first (no addition)'}
        if verify and verify.get('msm_live_terms'):
            bms = verify['gpu_ms_by_family_per_step'].get('+v_msm_bucket', 0.0)
            terms = verify['msm_live_terms']
            # a 256-bit term has a digit in all 16 windows, a 128-bit one in 8; per proof 135 of the former and 312 of the latter are expected at n = 16
            # (10 zero-bit slots x (10 + 26), 10 one-bit slots x 2, 8 membership groups x (4 + 4), 3 per proof); a bucket's first term is a copy
            nq = (n_log2 + 1) // 2
            t256, t128 = 10 * 10 + nq * 4 + 3, 10 * 26 + 10 * 2 + nq * 4
            adds = terms * (16 * t256 + 8 * t128) / (t256 + t128) - 2 * 16 * 8 * 65535 * ((B + vchunk - 1) // vchunk) / 2
            ops = adds * 8 * MACS_PER_MODMUL
            if bms > 0:
                others['k_msm_bucket'] = {'units': int(adds), 'unit': 'niels additions into (window, group, digit) buckets, 8 products each (estimate from the live-term count)',
                                          'live_terms': terms, 'ms_per_step': round(bms, 2), 'achieved': round(ops / (bms * 1e-3) / 1e12, 3), 'peak': VALU_MAD_PEAK_TOPS,
                                          'frac': round(ops / (bms * 1e-3) / 1e12 / VALU_MAD_PEAK_TOPS, 4),
                                          'note': 'one lane per bucket, lanes of a wave walk lists of equal length (buckets ordered by size); gathers of 128-byte entries'}
        roofline['others'] = others
        if verify and 'k_msm_bucket' in others:
            # the verify half priced like the prove half: its largest arithmetic kernel against the multiplier peak, and how much of the (serial) verify step is
            # not modular arithmetic at all -- SHA-256 (challenges, sampler), the grouping of the bucket pass's keys, packing, parsing
            vf = verify['gpu_ms_by_family_per_step']
            vsum = sum(v for k, v in vf.items() if not k.startswith('+'))
            non_arith = {k: vf[k] for k in ('v_hash', '+v_msm_group') if k in vf}
            verify['roofline'] = {'bound': 'valu_int32', 'kernel': 'k_msm_bucket', 'achieved': others['k_msm_bucket']['achieved'], 'peak': VALU_MAD_PEAK_TOPS,
                                  'unit': roofline['unit'], 'frac': others['k_msm_bucket']['frac'], 'ms_per_step': others['k_msm_bucket']['ms_per_step'],
                                  'share_of_gpu_time': round(others['k_msm_bucket']['ms_per_step'] / vsum, 3) if vsum else None,
                                  'traffic': 19015000000, 'traffic_note': 'bytes fetched per launch of a 32 768-proof chunk (FETCH_SIZE, its own rocprofv3 --pmc pass, raw: '
                                  'profiles/r05_pmc_verify.txt; a constant of bench.py, NOT measured in this run): the 137 M gathered 128-byte entries and their ids (18.2 GB), nothing re-read; '
                                  'SIMD busy 0.945',
                                  'family_sum_ms_per_step': round(vsum, 2), 'single_lane_wall_ms_per_step': verify.get('gpu_ms_single_lane_wall'),
                                  'overlapped_families': ['v_msm_p256'],   # on an auxiliary stream beside v_msm_tom even with one lane: the family sum exceeds the wall time by that overlap
                                  'whole_step_simd_busy': VERIFY_WHOLE_STEP_SIMD_BUSY, 'whole_step_source': VERIFY_PMC_SOURCE,
                                  'non_arithmetic_ms': non_arith,
                                  'non_arithmetic_share': round(sum(non_arith.values()) / vsum, 3) if vsum else None,
                                  'note': 'non_arithmetic = SHA-256 (both challenges, the sampler\'s fills) and the hand-written grouping of the bucket pass\'s keys '
                                          '(k_msm_hist / _scatter / _binsort); v_parse_validate is arithmetic (one curve check per point)'}
        cpu = None
        if not args.no_cpu_baseline:   # rank 0, at every N (the other ranks wait at the final barrier)
            sample = args.cpu_sample or 4 * host_cores()
            cpu, oproofs = cpu_baseline(nh, tg, th, ring, nkeys, msg, sig, pk, which, seeds, sec, min(sample, B),
                                        vseeds=rank_seeds(seeds, rank + 1000) if verify else None)
            if '_verify_verdicts' in cpu:   # the oracle's verdicts and statuses for the sample against the engine's (same proofs, same verifier seeds)
                ook, ovst = cpu.pop('_verify_verdicts')
                ns = len(ook)
                assert d_ok[:ns].cpu().tolist() == ook and d_vst[:ns].cpu().tolist() == ovst, 'GPU verdicts differ from the oracle'
                cpu['verify']['checked_verdicts_equal'] = ns
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
        latency = None
        if args.latency and world == 1:
            try:
                latency = latency_table(Z, eng, args, nh, tg, th, sec, ring, nkeys, msg, sig, pk, which, seeds)
            except Exception as e:  # an auxiliary measurement must never cost the bench line
                import traceback
                latency = {'error': repr(e)[:300], 'where': traceback.format_exc()[-600:]}
            eng.set_chunk(min(args.chunk, B))
        ms_per_step = dt * 1e3 / args.steps
        line = {
            'metric': 'proveSignatureList proofs/sec', 'value': round(world * B * args.steps / dt, 2), 'unit': 'proofs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u32 (9x30-bit limbs, 256/258-bit modular integers)',
            'data': 'synthetic',
            'config': {'workload': 'B=%d proofs/GPU/step, ring=%d keys (n=%d)%s, secLevel=%d, P-256+Tom-256, %d-bit combs, chunk=%d x %d lanes'
                                   % (B, nkeys, n_log2, ', per-key tables' if kt_count else '', sec, args.comb_bits, eng_chunk(args, B), args.lanes),
                       'parallelism': 'proofs sharded per GPU; ring broadcast over RCCL at set-up' if world > 1 else 'single GPU'},
            'set_params_s': round(t_tab, 3),
            'key_table_proofs_last_chunk': kt_count,   # zk_test_counter(ctx, 1): proofs of the last chunk whose multiples of the signer's key came from the per-key tables
            'hbm_used_gb': round(hbm_used / 2**30, 1),   # tables + both lanes' prover and verifier workspaces + this step's proofs
            'proof_bytes_per_step': total_bytes, 'failed_proofs': nbad,
            'gpu_ms_by_family_per_step': {k: round(v / max(1, args.roofline_steps), 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])},
            'gpu_ms_note': 'serial single-lane pass (sum = %.1f ms); the timed steps overlap %d chunks on %d streams' % (gpu_ms / max(1, args.roofline_steps), args.lanes, args.lanes),
            'roofline': roofline, 'cpu_baseline': cpu, 'verify': verify, 'host_io': host_io, 'json': json_rates, 'latency': latency,
        }
        if latency and 'rings' in latency:   # one proof per call is the reference's only shape (src/zkpAttestList.ts:104-145)
            b1 = latency['rings'].get(str(nkeys), {}).get('1')
            if b1:
                line['latency_ms_b1'] = b1['prove_ms']
                line['verify_latency_ms_b1'] = b1['verify_ms']
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
            sp = host_io.get('stream_packed') or {}
            if 'proofs_per_s' in (sp.get('prove') or {}):   # the same batches in the packed wire layout (zk_ctx_set_wire(ZK_WIRE_ZKA1P))
                line['value_pcie_inclusive_steady_packed'] = sp['prove']['proofs_per_s']
            if 'verifies_per_s' in (sp.get('verify') or {}):
                line['verify_pcie_inclusive_steady_packed'] = sp['verify']['verifies_per_s']
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def eng_chunk(args, B):
    return min(args.chunk, B)


if __name__ == '__main__':
    main()
