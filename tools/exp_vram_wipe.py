#!/usr/bin/env python3
"""Does releasing HBM halve the PCIe copy rate for a while?  (the kernel driver wipes released VRAM in the background.)
One process on an idle GPU: for G in sizes: hipMalloc G GB, touch it, hipFree it, then zk_ctx_copy_probe every ~60 ms for a few seconds.
Prints, per size, how long the probe stayed below 0.8 of the settled rate.
  python tools/exp_vram_wipe.py 4,32,128"""
import json
import os
import sys
import time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '4,32,128').split(',')]
    import torch
    import zkp_ecdsa_amd as Z
    eng = Z.Engine(0)
    time.sleep(6)                      # whatever the previous process released is wiped by now
    base = eng.copy_probe(0, 64 << 20)[0]
    out = {'settled_gbps': base, 'runs': []}
    for g in sizes:
        x = torch.empty(g << 30, dtype=torch.uint8, device='cuda:0')
        x.fill_(1)
        torch.cuda.synchronize()
        before = eng.copy_probe(0, 64 << 20)[0]
        del x
        torch.cuda.empty_cache()
        t0 = time.time()
        tl = []
        while time.time() - t0 < max(3.0, g / 20):
            r = eng.copy_probe(0, 64 << 20)[0]
            tl.append((round(time.time() - t0, 3), r))
            time.sleep(0.03)
        slow = [t for t, r in tl if r < 0.8 * base]
        out['runs'].append({'released_gb': g, 'rate_while_allocated': before, 'slow_until_s': max(slow) if slow else 0.0, 'n_slow': len(slow), 'n': len(tl),
                            'min_rate': min(r for _, r in tl), 'timeline_head': tl[:12]})
        time.sleep(2)
    print(json.dumps(out), flush=True)
    os._exit(0)


if __name__ == '__main__':
    main()
