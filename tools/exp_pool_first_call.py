#!/usr/bin/env python3
"""Reproducer for the "first synchronous pool call" anomaly (profiles/README.md round 3): ONE fresh process,
zk_pool_create([0]) -> params -> ring -> zk_pool_host_alloc -> three synchronous zk_pool_prove_batch calls of --batch proofs, then the
same batches streamed.  Prints one JSON line: proofs/s and D2H GB/s of every call, where the pages of the output buffer sit
(/proc/self/numa_maps), how much of it is backed by huge pages (/proc/self/smaps), the CPU / NUMA node the calling thread ran on, and
the allocation strategy (ZKATTEST_POOL_ALLOC).  Run it N times from a shell loop: every start is a fresh process.
  python tools/exp_pool_first_call.py [--batch 65536] [--stream 4]"""
import argparse
import ctypes as C
import json
import os
import sys
import time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mapping_info(addr):
    """(numa_maps line fields, AnonHugePages kB, Rss kB) of the mapping that starts at addr"""
    info = {}
    try:
        for line in open('/proc/self/numa_maps'):
            if line.split()[0] == '%x' % addr:
                info['numa_maps'] = ' '.join(f for f in line.split()[1:] if f.startswith(('N', 'anon', 'kernelpagesize', 'prefer', 'bind', 'default')))
    except OSError:
        pass
    try:
        cur = False
        for line in open('/proc/self/smaps'):
            if '-' in line.split()[0] and len(line.split()) >= 5 and line.split()[0].count('-') == 1 and not line.split()[0].endswith(':'):
                cur = line.split('-')[0] == '%x' % addr
            elif cur and line.startswith(('AnonHugePages', 'Rss', 'Size')):
                info[line.split(':')[0] + '_kB'] = int(line.split()[1])
    except OSError:
        pass
    return info


def cpu_node():
    cpu = C.CDLL(None).sched_getcpu()
    node = -1
    try:
        for d in os.listdir('/sys/devices/system/cpu/cpu%d' % cpu):
            if d.startswith('node'):
                node = int(d[4:])
    except OSError:
        pass
    return cpu, node


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=65536)
    ap.add_argument('--ring', type=int, default=65536)
    ap.add_argument('--comb-bits', type=int, default=24)
    ap.add_argument('--sync-calls', type=int, default=3)
    ap.add_argument('--stream', type=int, default=4)
    ap.add_argument('--tag', default='')
    ap.add_argument('--clean-exit', type=int, default=0, help='1: destroy the pool and free the buffers before exiting (default: os._exit)')
    ap.add_argument('--probe', type=int, default=1, help='zk_ctx_copy_probe on the four copy streams at start and at the end')
    ap.add_argument('--hold', type=float, default=0, help='only create the pool, probe, and sleep this many seconds (a process that holds its queues)')
    ap.add_argument('--torch-first', type=int, default=0, help='1: import torch (and touch the device) before the engine library: the process then runs on the HIP runtime bundled with the wheel')
    args = ap.parse_args()
    if args.torch_first:
        import torch
        torch.zeros(1, device='cuda:0')
    import zkp_ecdsa_amd as Z
    B, sec = args.batch, 80
    t_start = time.time()
    pool = Z.Pool([0])
    e0 = pool.engine(0)
    e0.set_comb_bits(args.comb_bits), e0.set_chunk(min(16384, B)), e0.set_lanes(2)
    probe0 = [e0.copy_probe(l) for l in range(4)] if args.probe else None   # before anything else ran on the device
    if args.hold:   # holder process: keeps its queues alive while others start
        print(json.dumps({'tag': args.tag, 'probe_at_start': probe0}), flush=True)
        time.sleep(args.hold)
        os._exit(0)
    pool.set_params(*e0.synth_params(2024), sec)
    ring, msg, sig, pk, which, seeds = e0.synth_workload(2024, args.ring, B)
    pool.set_ring(ring, args.ring)
    n_log2 = max(1, (args.ring - 1).bit_length())
    cap = int(B * (304 + 336 * sec + 3392 * (sec // 2 + 4) + 384 * n_log2 + 32) + (64 << 20))
    t0 = time.time()
    pin = Z.PinnedBuffer(cap, pool=pool)
    rec = {'tag': args.tag, 'alloc': os.environ.get('ZKATTEST_POOL_ALLOC', 'register'), 'device_numa_node': pool.numa_node(0), 'alloc_s': round(time.time() - t0, 3),
           'setup_s': round(t0 - t_start, 2), 'buffer': mapping_info(pin.ptr), 'calls': []}
    for k in range(args.sync_calls):
        cpu, node = cpu_node()
        dt, off, ln, st = pool.prove_batch_raw(msg, sig, pk, which, seeds, pin, cap)
        nbytes = sum(ln)
        rec['calls'].append({'kind': 'sync', 'proofs_per_s': round(B / dt), 'd2h_gbps': round(nbytes / dt / 1e9, 1), 'cpu': cpu, 'cpu_node': node})
    if args.stream > 1:
        for i in range(1):
            e0.set_chunk(min(22016, B)), e0.set_lanes(3), e0.set_slice(8192)
        bufs = [pin] + [Z.PinnedBuffer(cap, pool=pool) for _ in range(2)]
        t0 = time.time()
        tk = [pool.prove_submit(msg, sig, pk, which, seeds, bufs[k % 3], cap) for k in range(min(3, args.stream))]
        for k in range(args.stream):
            pool.prove_wait(tk[k])
            if k + 3 < args.stream:
                tk.append(pool.prove_submit(msg, sig, pk, which, seeds, bufs[(k + 3) % 3], cap))
        dt = time.time() - t0
        rec['calls'].append({'kind': 'stream x%d' % args.stream, 'proofs_per_s': round(args.stream * B / dt), 'd2h_gbps': round(args.stream * nbytes / dt / 1e9, 1)})
        rec['other_buffers'] = [mapping_info(b.ptr) for b in bufs[1:]]
    rec['probe_at_start'] = probe0
    rec['hip_runtime'] = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime64' in l})
    if args.probe:
        e0.set_slice(0)
        rec['probe_at_end'] = [e0.copy_probe(l) for l in range(4)]
    print(json.dumps(rec), flush=True)
    if args.clean_exit:
        pin.free()
        pool.close()
        return
    os._exit(0)   # a fresh process per measurement: no tear-down in the timed path of the loop that starts us


if __name__ == '__main__':
    main()
