#!/usr/bin/env python3
"""Is the half-rate state a property of the BUFFER or of the moment?  Run right after a process that held GBs of page-locked memory has
exited: allocate page-locked buffer h1 at once, h2 after 3 s, h3 after 6 s; time a 256 MiB device-to-host copy into each of them every 0.5 s.
  python tools/exp_slow_buffer_heals.py hold   (the predecessor: allocates 12 GB page-locked + 60 GB of HBM, touches them, exits abruptly)
  python tools/exp_slow_buffer_heals.py        (the observer)"""
import ctypes as C
import json
import os
import sys
import time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
os.environ['ZKATTEST_HOST_ALLOC_PROBE'] = '0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402
import zkp_ecdsa_amd as Z   # noqa: E402

L = Z.lib()
if len(sys.argv) > 1 and sys.argv[1] == 'hold':
    p = L.zk_host_alloc(12 << 30)
    C.memset(p, 1, 12 << 30)
    x = torch.ones(60 << 30, dtype=torch.uint8, device='cuda:0')
    h = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
    h.copy_(x[:1 << 30])
    torch.cuda.synchronize()
    os._exit(0)
n = 256 << 20
d = torch.ones(n, dtype=torch.uint8, device='cuda:0')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time()
bufs = {}


def alloc(name):
    p = L.zk_host_alloc(n)
    C.memset(p, 1, n)
    bufs[name] = (p, torch.frombuffer((C.c_uint8 * n).from_address(p), dtype=torch.uint8), round(time.time() - t0, 2))


def rate(h):
    e0.record()
    h.copy_(d, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    return round(n / 1e6 / e0.elapsed_time(e1), 1)


alloc('h1')
tl = []
while time.time() - t0 < 9:
    t = time.time() - t0
    if t > 3 and 'h2' not in bufs:
        alloc('h2')
    if t > 6 and 'h3' not in bufs:
        alloc('h3')
    tl.append((round(t, 1), {k: rate(v[1]) for k, v in bufs.items()}))
    time.sleep(0.5)
print(json.dumps({'allocated_at_s': {k: v[2] for k, v in bufs.items()}, 'timeline': tl}), flush=True)
os._exit(0)
