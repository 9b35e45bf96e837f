import os, sys, time, ctypes as C
sys.path.insert(0, '/root/repo')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import zkp_ecdsa_amd as Z
n = 4 << 30
pool = Z.Pool([0])
d = torch.empty(n, dtype=torch.uint8, device='cuda:0')
hip = C.CDLL('libamdhip64.so')
for name, buf in (('zk_host_alloc', Z.PinnedBuffer(n)), ('zk_pool_host_alloc', Z.PinnedBuffer(n, pool=pool))):
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        rc = hip.hipMemcpy(C.c_void_p(buf.ptr), C.c_void_p(d.data_ptr()), C.c_size_t(n), 2)   # D2H
        torch.cuda.synchronize()
        dt = time.time() - t0
    print('%-20s D2H %.1f GB/s (rc %d)' % (name, n / dt / 1e9, rc))
    buf.free()
print(open('/sys/kernel/mm/transparent_hugepage/enabled').read().strip())
print([l for l in open('/proc/meminfo') if 'Huge' in l or 'MemFree' in l])
