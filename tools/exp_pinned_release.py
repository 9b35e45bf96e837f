#!/usr/bin/env python3
"""Does releasing PAGE-LOCKED HOST memory halve the PCIe copy rate for a while?  One process on an idle GPU, a persistent 64 MiB page-locked
probe buffer; for G in sizes: allocate G GB page-locked (hipHostMalloc), touch, release, then time a device-to-host copy into the probe
buffer every ~30 ms.  Prints per size how long the copies stayed below 0.8 of the settled rate.
  python tools/exp_pinned_release.py 1,4,16"""
import ctypes as C
import json
import os
import sys
import time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
os.environ['ZKATTEST_HOST_ALLOC_PROBE'] = '0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    sizes = [float(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '1,4,16').split(',')]
    mode = sys.argv[2] if len(sys.argv) > 2 else 'host'
    import torch
    import zkp_ecdsa_amd as Z
    L = Z.lib()
    dev = torch.device('cuda:0')
    n = 64 << 20
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    d = torch.ones(n, dtype=torch.uint8, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def rate():
        e0.record()
        h.copy_(d, non_blocking=True)
        e1.record()
        torch.cuda.synchronize()
        return round(n / 1e6 / e0.elapsed_time(e1), 1)
    time.sleep(6)
    rate()
    base = max(rate() for _ in range(5))
    out = {'settled_gbps': base, 'mode': mode, 'runs': []}
    for g in sizes:
        nbytes = int(g * (1 << 30))
        if mode == 'host':
            p = L.zk_host_alloc(nbytes)
            C.memset(p, 1, nbytes)
            before = rate()
            L.zk_host_free(p)
        else:   # 'register': mmap'ed memory registered and unregistered (the zk_pool_host_alloc way) -- through a pool of one device
            pool = out.setdefault('_pool', None) or Z.Pool([0])
            out['_pool'] = pool
            p = L.zk_pool_host_alloc(pool.h, nbytes)
            before = rate()
            L.zk_pool_host_free(p)
        t0 = time.time()
        tl = []
        while time.time() - t0 < max(2.0, g / 3):
            tl.append((round(time.time() - t0, 3), rate()))
            time.sleep(0.02)
        slow = [t for t, r in tl if r < 0.8 * base]
        out['runs'].append({'released_gb': g, 'rate_while_allocated': before, 'slow_until_s': max(slow) if slow else 0.0, 'n_slow': len(slow), 'n': len(tl),
                            'min_rate': min(r for _, r in tl), 'timeline_head': tl[:10]})
        time.sleep(3)
    out.pop('_pool', None)
    print(json.dumps(out), flush=True)
    os._exit(0)


if __name__ == '__main__':
    main()
