"""-m gpu: the streamed host-pointer calls (zk_prove_submit / zk_prove_wait, zk_verify_submit / zk_verify_wait; include/zkattest.h
"two batches in flight"): several jobs queued on one context, the stage-1 look-ahead crossing job boundaries.  The bytes, statuses
and verdicts must be those of the synchronous calls, whatever the job sizes, the chunk size and the number of lanes."""
import ctypes as C
import os

import pytest
import torch   # (tests/conftest.py has loaded it already: one HIP runtime per process)

pytestmark = pytest.mark.gpu


def _setup(S, nkeys, B, chunk, lanes):
    import zkp_ecdsa_amd as Z
    eng = Z.Engine(0)
    params = eng.synth_params(S)
    eng.set_comb_bits(16)
    eng.set_params(*params, 80)
    work = eng.synth_workload(S, nkeys, B)
    eng.set_ring(work[0], nkeys)
    eng.set_chunk(chunk)
    eng.set_lanes(lanes)
    return Z, eng, work


def _cut(work, a, b):
    ring, msg, sig, pk, which, seeds = work
    return msg[32 * a:32 * b], sig[64 * a:64 * b], pk[64 * a:64 * b], which[a:b], seeds[32 * a:32 * b]


@pytest.mark.parametrize('chunk,lanes,sizes', [(256, 2, (700, 300, 1000, 130)), (512, 3, (512, 1536, 100)), (4096, 2, (900, 900))])
def test_streamed_prove_and_verify_equal_the_synchronous_calls(chunk, lanes, sizes):
    B = sum(sizes)
    Z, eng, work = _setup(77, 4096, B, chunk, lanes)   # ring >= batch: every proof's key stays in the ring
    cap = eng.proof_max_size()
    bounds, a = [], 0
    for n in sizes:
        bounds.append((a, a + n))
        a += n
    # synchronous reference: one zk_prove_batch per job on a page-locked buffer
    ref = []
    for (a, b) in bounds:
        m, s, p, w, sd = _cut(work, a, b)
        pin = Z.PinnedBuffer(cap * (b - a) * 7 // 10 + (8 << 20))
        _, _, off, st = eng.prove_batch_host_raw(m, s, p, w, sd, out=pin)
        assert not any(st)
        ref.append((bytes(pin.view[:off[b - a]]), list(off)))
        pin.free()
    # streamed: submit(0); submit(1); wait(0); submit(2); wait(1); ...
    pins = [Z.PinnedBuffer(cap * (b - a) * 7 // 10 + (8 << 20)) for (a, b) in bounds]
    tickets, got = [], []
    for k, (a, b) in enumerate(bounds):
        tickets.append(eng.prove_submit(*_cut(work, a, b), pins[k]))
        if k >= 1:
            got.append(eng.prove_wait(tickets[k - 1]))
    got.append(eng.prove_wait(tickets[-1]))
    for k, (a, b) in enumerate(bounds):
        off, st = got[k]
        assert not any(st)
        assert list(off) == ref[k][1]
        assert bytes(pins[k].view[:off[b - a]]) == ref[k][0], 'job %d differs from the synchronous call' % k
    # ---- verify: a forged proof in job 1, a truncated header claim in job 2; fixed verifier seeds -> the synchronous verdicts
    vseeds = [os.urandom(32 * (b - a)) for (a, b) in bounds]
    offs = [got[k][0] for k in range(len(bounds))]
    if len(bounds) > 1:
        o = offs[1]
        pins[1].view[o[5 + 1] - 9] ^= 1          # a response byte of proof 5 of job 1
    if len(bounds) > 2:
        o = offs[2]
        pins[2].view[o[3] + 11] ^= 1             # secLevel field of proof 3 of job 2
    sync = []
    for k, (a, b) in enumerate(bounds):
        _, ok, vst = eng.verify_batch_host_raw(work[1][32 * a:32 * b], pins[k], offs[k], b - a, vseeds[k])
        sync.append((list(ok), list(vst)))
    vt, vgot = [], []
    for k, (a, b) in enumerate(bounds):
        vt.append(eng.verify_submit(work[1][32 * a:32 * b], pins[k], offs[k], b - a, vseeds[k]))
        if k >= 1:
            vgot.append(eng.verify_wait(vt[k - 1]))
    vgot.append(eng.verify_wait(vt[-1]))
    for k in range(len(bounds)):
        assert (list(vgot[k][0]), list(vgot[k][1])) == sync[k], 'verdicts of job %d differ' % k
    if len(bounds) > 1:
        assert sync[1][0][5] == 0 and sum(sync[1][0]) == bounds[1][1] - bounds[1][0] - 1
    assert sum(sync[0][0]) == bounds[0][1] - bounds[0][0]
    # default verifier seeds (drawn on the device from OS randomness): honest jobs are accepted
    t = eng.verify_submit(work[1][:32 * sizes[0]], pins[0], offs[0], sizes[0])
    ok, vst = eng.verify_wait(t)
    assert sum(ok) == sizes[0] and not any(vst)
    for p in pins:
        p.free()
    eng.close()


def test_streamed_calls_enforce_their_rules():
    B = 600
    Z, eng, work = _setup(78, 1024, B, 256, 2)
    cap = eng.proof_max_size()
    pins = [Z.PinnedBuffer(cap * 200 * 7 // 10 + (8 << 20)) for _ in range(5)]
    pageable = (C.c_uint8 * (cap * 10))()
    args = _cut(work, 0, 200)

    class Fake:   # a pageable buffer dressed up as a PinnedBuffer
        ptr, nbytes = C.addressof(pageable), C.sizeof(pageable)
    with pytest.raises(Z.ZkError) as e:
        eng.prove_submit(*_cut(work, 0, 10), Fake)
    assert e.value.status == 14 and 'page-locked' in str(e.value)
    t0 = eng.prove_submit(*args, pins[0])
    t1 = eng.prove_submit(*_cut(work, 200, 400), pins[1])
    with pytest.raises(Z.ZkError) as e:   # the synchronous calls are refused while jobs are queued
        eng.prove_batch_host_raw(*_cut(work, 0, 4))
    assert e.value.status == 14 and 'streamed' in str(e.value)
    with pytest.raises(Z.ZkError) as e:   # nor may the ring (table E, key tables) or the parameters be replaced under queued jobs
        eng.set_ring(work[0], 4096)
    assert e.value.status == 14 and 'streamed' in str(e.value)
    # the settings the queued jobs were planned with are frozen: lanes, chunk, slices, mode, ring fold, verifier shape, comb width
    for setter, arg in ((eng.set_lanes, 3), (eng.set_chunk, 128), (eng.set_slice, 64), (eng.set_mode, 1), (eng.set_ring_fold, 0), (eng.set_verify_groups, 64),
                        (eng.set_batch_verify, 0), (eng.set_comb_bits, 12), (eng.set_host_taper, 0)):
        with pytest.raises(Z.ZkError) as e:
            setter(arg)
        assert e.value.status == 14 and 'streamed' in str(e.value), setter
    with pytest.raises(Z.ZkError) as e:   # waits in submission order
        eng.prove_wait(t1)
    assert 'submission order' in str(e.value)
    t2 = eng.prove_submit(*_cut(work, 400, 600), pins[2])
    t3 = eng.prove_submit(*args, pins[3])
    with pytest.raises(Z.ZkError) as e:   # at most four jobs
        eng.prove_submit(*args, pins[4])
    assert 'too many' in str(e.value)
    off0, st0 = eng.prove_wait(t0)
    with pytest.raises(Z.ZkError) as e:   # kinds do not mix
        eng.verify_submit(work[1][:32 * 200], pins[0], off0, 200)
    assert 'together' in str(e.value)
    for t in (t1, t2, t3):
        off, st = eng.prove_wait(t)
        assert not any(st)
    assert bytes(pins[3].view[:off[200]]) == bytes(pins[0].view[:off0[200]])   # the same statements and seeds: the same bytes
    # an output buffer that is too small fails that job only
    small = Z.PinnedBuffer(1 << 20)
    ta = eng.prove_submit(*args, small)
    tb = eng.prove_submit(*_cut(work, 200, 400), pins[1])
    with pytest.raises(Z.ZkError) as e:
        eng.prove_wait(ta)
    assert e.value.status == 12
    off, st = eng.prove_wait(tb)
    assert not any(st)
    ok, vst = eng.verify_batch_host_raw(work[1][32 * 200:32 * 400], pins[1], off, 200)[1:]
    assert sum(ok) == 200
    # a context destroyed with jobs still queued releases them
    eng.prove_submit(*args, pins[0])
    eng.prove_submit(*args, pins[2])
    eng.close()
    for p in pins + [small]:
        p.free()


def test_streamed_pool_calls_equal_the_synchronous_pool_calls():
    """zk_pool_prove_submit / _wait and the verify pair: every device's shard streamed on its own context (two contexts on device 0
    here), three jobs, two in flight; bytes, (offset, length) pairs and verdicts are those of zk_pool_prove_batch / zk_pool_verify_batch."""
    import zkp_ecdsa_amd as Z
    B, nkeys = 1500, 2048
    pool = Z.Pool([0, 0])
    e0 = pool.engine(0)
    params = e0.synth_params(91)
    for i in range(2):
        pool.engine(i).set_comb_bits(16)
        pool.engine(i).set_chunk(256)
    pool.set_params(*params, 80)
    ring, msg, sig, pk, which, seeds = e0.synth_workload(91, nkeys, B)
    pool.set_ring(ring, nkeys)
    jobs = [(0, 500), (500, 1100), (1100, 1500)]
    cap = 2 * ((e0.proof_max_size() * 300 * 7 // 10 + (4 << 20)) & ~255)
    cut = lambda a, b: (msg[32 * a:32 * b], sig[64 * a:64 * b], pk[64 * a:64 * b], which[a:b], seeds[32 * a:32 * b])
    ref = []
    for (a, b) in jobs:
        pin = Z.PinnedBuffer(cap, pool=pool)
        _, off, ln, st = pool.prove_batch_raw(*cut(a, b), pin, cap)
        assert not any(st)
        ref.append([bytes(pin.view[off[k]:off[k] + ln[k]]) for k in range(b - a)])
        pin.free()
    pins = [Z.PinnedBuffer(cap, pool=pool) for _ in jobs]
    tk, got = [], []
    for k, (a, b) in enumerate(jobs):
        tk.append(pool.prove_submit(*cut(a, b), pins[k], cap))
        if k:
            got.append(pool.prove_wait(tk[k - 1]))
    got.append(pool.prove_wait(tk[-1]))
    for k, (a, b) in enumerate(jobs):
        off, ln, st = got[k]
        assert not any(st)
        assert [bytes(pins[k].view[off[i]:off[i] + ln[i]]) for i in range(b - a)] == ref[k], 'job %d differs' % k
    # verify, with a forged proof in the second shard of job 1
    off1, ln1, _ = got[1]
    pins[1].view[off1[450] + ln1[450] - 9] ^= 1
    vs = [os.urandom(32 * (b - a)) for (a, b) in jobs]
    sync = [pool.verify_batch_raw(msg[32 * a:32 * b], pins[k], got[k][0], got[k][1], b - a, vs[k])[1:] for k, (a, b) in enumerate(jobs)]
    vt, vg = [], []
    for k, (a, b) in enumerate(jobs):
        vt.append(pool.verify_submit(msg[32 * a:32 * b], pins[k], got[k][0], got[k][1], b - a, vs[k]))
        if k:
            vg.append(pool.verify_wait(vt[k - 1]))
    vg.append(pool.verify_wait(vt[-1]))
    for k, (a, b) in enumerate(jobs):
        assert (list(vg[k][0]), list(vg[k][1])) == (list(sync[k][0]), list(sync[k][1]))
    assert [i for i in range(600) if not vg[1][0][i]] == [450] and sum(vg[0][0]) == 500 and sum(vg[2][0]) == 400
    for p_ in pins:
        p_.free()
    pool.close()


def test_a_pool_submit_that_fails_half_way_leaves_the_older_jobs_waitable():
    """zk_pool_prove_submit / zk_pool_verify_submit failing at device slot 1 after slot 0 was submitted, with an older pool job still in
    flight on both contexts: the half-submitted job is taken out again (not waited for out of order), the older job completes with its
    bytes, and the pool goes on working -- also when the failing submit met an idle pipeline (the shard then runs out and is dropped)."""
    import subprocess
    import sys
    import zkp_ecdsa_amd as Z
    th = os.path.join(os.path.dirname(Z.LIB_PATH), 'libzkattest_hip_testhooks.so')
    if not os.path.exists(th):
        pytest.skip('the test-hooks build is not there (make -C zkp-ecdsa_amd/csrc testhooks)')
    # the product library must not be able to inject a failure
    assert not hasattr(Z.lib(), 'zk_test_pool_fail_next_submit')
    env = dict(os.environ)
    env['ZKATTEST_LIB'] = th
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, 'tests', 'pool_fail_check.py')], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and 'pool_fail_check ok' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]



def test_streamed_device_pointer_jobs_equal_the_synchronous_device_call():
    """zk_prove_submit_device: three jobs of different sizes in flight on buffers that never leave HBM; bytes, offsets and statuses
    are those of zk_prove_batch_device."""
    sizes = (600, 1000, 200)
    B = sum(sizes)
    Z, eng, work = _setup(78, 4096, B, 256, 3)
    dev = torch.device('cuda:0')
    tb = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    cap1 = eng.proof_max_size()
    jobs, a = [], 0
    for n in sizes:
        m, s, p, w, sd = _cut(work, a, a + n)
        a += n
        jobs.append({'n': n, 'in': (tb(m), tb(s), tb(p), torch.tensor(list(w), dtype=torch.int32, device=dev), tb(sd)),
                     'out': [(torch.zeros(cap1 * n, dtype=torch.uint8, device=dev), torch.zeros(n + 1, dtype=torch.int64, device=dev),
                              torch.ones(n, dtype=torch.int32, device=dev)) for _ in range(2)]})
    torch.cuda.synchronize()
    args = lambda j, o: (j['n'],) + tuple(t.data_ptr() for t in j['in']) + (o[0].data_ptr(), cap1 * j['n'], o[1].data_ptr(), o[2].data_ptr())
    for j in jobs:
        eng.prove_batch_device(*args(j, j['out'][0]))
    tickets = [eng.prove_submit_device(*args(j, j['out'][1])) for j in jobs]
    with pytest.raises(Z.ZkError):   # the synchronous calls are refused while jobs are queued
        eng.prove_batch_device(*args(jobs[0], jobs[0]['out'][0]))
    for t in tickets:
        eng.prove_wait(t)
    torch.cuda.synchronize()
    for j in jobs:
        (o0, f0, s0), (o1, f1, s1) = j['out']
        end = int(f0[j['n']].item())
        assert end > 0 and torch.equal(f0, f1) and torch.equal(s0, s1) and int(s0.abs().sum().item()) == 0
        assert torch.equal(o0[:end], o1[:end])
    eng.close()
