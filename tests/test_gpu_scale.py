"""-m gpu: the BASELINE.json configurations at their stated sizes, and the multi-device pool of the C ABI.

  configs[2]  batch = 65 536 proofs, ring = 2^16, one GPU: prove all through the host-pointer entry point (page-locked output,
              tapered chunk plan), verify ALL on the GPU, diff a seeded 1 % sample against the oracle byte for byte
              ("diff a random 1 % sample + verify all"), planted forgeries rejected.
  configs[4]  verifySignatureList over a ring of 2^20 keys with thousands of proofs per GPU (the per-GPU shard of the
              2^20 x 2^20 job runs as a stream of such calls, bench.py --mode verify): honest proofs accepted, planted
              forgeries rejected, verdict parity with the oracle on a few of them.
  configs[3]  the sharded multi-GPU call: zk_pool over min(2, visible) devices -- the same device twice on a one-GPU box --
              against the single-device bytes and the oracle.
"""
import ctypes as C
import hashlib
import os
import random

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(S, nkeys, B, sec=80, comb_bits=None):
    import zkp_ecdsa_amd as Z
    eng = Z.Engine(0)
    params = eng.synth_params(S)
    if comb_bits:
        eng.set_comb_bits(comb_bits)
    eng.set_params(*params, sec)
    work = eng.synth_workload(S, nkeys, B)
    eng.set_ring(work[0], nkeys)
    return eng, params, work


def _oracle(params, ring, nkeys, sec=80):
    import coracle as CO
    octx = CO.OracleCtx(*params, sec)
    octx.set_ring(ring, nkeys)
    return octx


def test_baseline_config3_batch65536_ring65536_verify_all_diff_one_percent():
    import zkp_ecdsa_amd as Z
    B, nkeys = 65536, 65536
    eng, params, (ring, msg, sig, pk, which, seeds) = _engine(2024, nkeys, B, comb_bits=20)
    eng.set_chunk(8192)
    est = int(B * (304 + 336 * 80 + 3392 * 44 + 16 * 384 + 32) + (64 << 20))
    pin = Z.PinnedBuffer(est)
    _, _, off, st = eng.prove_batch_host_raw(msg, sig, pk, which, seeds, out=pin)
    assert not any(st), [b for b in range(B) if st[b]][:8]
    assert off[0] == 0 and off[B] <= est
    # --- verify all
    _, ok, vst = eng.verify_batch_host_raw(msg, pin, off, B)
    assert sum(ok) == B and not any(vst)
    # --- 1 % sample against the oracle (reference-faithful 2*N*n ring loop, window-4 multiplications, per-point inversions)
    rnd = random.Random(65536)
    sample = sorted(rnd.sample(range(B), B // 100))
    sub = lambda buf, w: b''.join(buf[w * b:w * b + w] for b in sample)
    octx = _oracle(params, ring, nkeys)
    exp, est_ = octx.prove_batch(sub(msg, 32), sub(sig, 64), sub(pk, 64), [which[b] for b in sample], seeds=sub(seeds, 32), nthreads=64)
    assert est_ == [0] * len(sample)
    bad = [b for j, b in enumerate(sample) if bytes(pin.view[off[b]:off[b + 1]]) != exp[j]]
    assert not bad, bad[:8]
    # --- planted forgeries: one flipped byte in 5 proofs (a response scalar near the end of each) -> exactly those are rejected
    forged = sorted(rnd.sample(range(B), 5))
    for b in forged:
        pin.view[off[b + 1] - 9] ^= 0x10
    before = eng.test_counter(0)
    _, ok, vst = eng.verify_batch_host_raw(msg, pin, off, B)
    assert [b for b in range(B) if not ok[b]] == forged
    # five forged proofs in 65 536: only the groups holding one (an eighth of a 8192-proof chunk each) went through the per-proof
    # sums -- asserted on the work counter, not on wall time (the boxes of the pool differ by 2-3x)
    rechecked = eng.test_counter(0) - before
    assert 0 < rechecked <= 5 * (8192 // 8), rechecked
    for b in forged:
        pin.view[off[b + 1] - 9] ^= 0x10
    before = eng.test_counter(0)
    _, ok, vst = eng.verify_batch_host_raw(msg, pin, off, B)
    assert sum(ok) == B and eng.test_counter(0) == before
    pin.free()
    eng.close()


def test_the_bench_configuration_itself_matches_the_oracle():
    """bench.py's headline configuration as a test: 24-bit combs, per-key tables, 65 536 proofs over a ring of 2^16 keys, chunks of 22 016 on three lanes,
    inputs and proofs resident in HBM (zk_prove_batch_device), then zk_verify_batch_device in chunks of 32 768 on two lanes.  64 seeded proofs are diffed
    byte for byte against the oracle and their verdicts / statuses compared for the bench's verifier seeds; every proof must verify.  (The bench asserts the
    same after its timed region; here a kernel regression at the headline shape turns the tier red.)"""
    import torch
    import zkp_ecdsa_amd as Z
    from bench_common import DEFAULT_COMB_BITS, rank_seeds
    assert DEFAULT_COMB_BITS == 24
    B, nkeys, sec = 65536, 65536, 80
    dev = torch.device('cuda', 0)
    eng = Z.Engine(0)
    params = eng.synth_params(2024)
    eng.set_comb_bits(DEFAULT_COMB_BITS)
    eng.set_params(*params, sec)
    eng.set_chunk(22016)
    eng.set_lanes(3)
    ring, msg, sig, pk, which, seeds = eng.synth_workload(2024, nkeys, B)
    tb = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_ring = tb(ring)
    eng.set_ring_device(d_ring.data_ptr(), nkeys)
    my_seeds, vseeds = rank_seeds(seeds, 0), rank_seeds(seeds, 1000)
    d_msg, d_sig, d_pk, d_seeds, d_vseeds = tb(msg), tb(sig), tb(pk), tb(my_seeds), tb(vseeds)
    d_which = torch.tensor(which, dtype=torch.int32, device=dev)
    cap = int(B * (304 + 336 * sec + 3392 * (sec // 2 + 4) + (4 * 72 + 96) * 20 + 32) + (64 << 20))
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
    d_st = torch.empty(B, dtype=torch.int32, device=dev)
    eng.prove_batch_device(B, d_msg.data_ptr(), d_sig.data_ptr(), d_pk.data_ptr(), d_which.data_ptr(), d_seeds.data_ptr(), d_out.data_ptr(), cap, d_off.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert int((d_st != 0).sum().item()) == 0
    assert eng.test_counter(1) > 0            # the last chunk's multiples of the signers' keys came from the per-key tables
    off = d_off.cpu().tolist()
    d_ok = torch.empty(B, dtype=torch.uint8, device=dev)
    d_vst = torch.empty(B, dtype=torch.int32, device=dev)
    eng.set_chunk(32768)
    eng.set_lanes(2)
    eng.verify_batch_device(B, d_msg.data_ptr(), d_out.data_ptr(), d_off.data_ptr(), d_vseeds.data_ptr(), d_ok.data_ptr(), d_vst.data_ptr())
    torch.cuda.synchronize()
    assert int(d_ok.sum().item()) == B and int((d_vst != 0).sum().item()) == 0
    # 64 seeded proofs (the first of every chunk and lane boundary among them) against the oracle
    rnd = random.Random(24)
    sample = sorted(set([0, 22015, 22016, 44031, 44032, 65535, 32767, 32768]) | set(rnd.sample(range(B), 56)))
    sub = lambda buf, w: b''.join(buf[w * b:w * b + w] for b in sample)
    octx = _oracle(params, ring, nkeys, sec)
    exp, est = octx.prove_batch(sub(msg, 32), sub(sig, 64), sub(pk, 64), [which[b] for b in sample], seeds=sub(my_seeds, 32), nthreads=64)
    assert est == [0] * len(sample)
    raw = {b: d_out[off[b]:off[b + 1]].cpu().numpy().tobytes() for b in sample}
    bad = [b for j, b in enumerate(sample) if raw[b] != exp[j]]
    assert not bad, bad[:8]
    ook, ovst = octx.verify_batch(sub(msg, 32), [raw[b] for b in sample], nthreads=64, vseeds=sub(vseeds, 32))[:2]
    ok, vst = d_ok.cpu().tolist(), d_vst.cpu().tolist()
    assert [ok[b] for b in sample] == ook and [vst[b] for b in sample] == ovst
    # a forged proof at the headline shape: the batched check fails for its group, the per-proof sums find it, the verdicts of the others stand
    victim = sample[7]
    d_out[off[victim + 1] - 9] ^= 0x10
    eng.verify_batch_device(B, d_msg.data_ptr(), d_out.data_ptr(), d_off.data_ptr(), d_vseeds.data_ptr(), d_ok.data_ptr(), d_vst.data_ptr())
    torch.cuda.synchronize()
    assert (d_ok == 0).nonzero().flatten().tolist() == [victim]
    eng.close()


def test_baseline_config5_shape_ring_2_20_verify_4096():
    import zkp_ecdsa_amd as Z
    B, nkeys = 4096, 1 << 20
    eng, params, (ring, msg, sig, pk, which, seeds) = _engine(77, nkeys, B, comb_bits=20)
    eng.set_chunk(2048)
    pin = Z.PinnedBuffer(int(B * (304 + 336 * 80 + 3392 * 46 + 20 * 384 + 32)))
    _, _, off, st = eng.prove_batch_host_raw(msg, sig, pk, which, seeds, out=pin)
    assert not any(st)
    vs = b''.join(hashlib.sha256(b'c5' + i.to_bytes(4, 'big')).digest() for i in range(B))
    _, ok, vst = eng.verify_batch_host_raw(msg, pin, off, B, vseeds=vs)
    assert sum(ok) == B and not any(vst)
    rnd = random.Random(20)
    forged = sorted(rnd.sample(range(B), 7))
    for j, b in enumerate(forged):   # alternately a GK response (the last scalars of a proof) and a PointAdd/Exp byte
        pos = off[b + 1] - 9 if j % 2 == 0 else off[b] + 304 + 336 + 100
        pin.view[pos] ^= 0x04
    _, ok, vst = eng.verify_batch_host_raw(msg, pin, off, B, vseeds=vs)
    rejected = [b for b in range(B) if not ok[b]]
    # a GK forgery is always caught; a forged repetition only if it is among the 20 sampled ones (the reference's behaviour):
    # the oracle decides which, for the same verifier seeds
    octx = _oracle(params, ring, nkeys)
    chk = forged[:3] + [b for b in (0, B - 1) if b not in forged]
    proofs = [bytes(pin.view[off[b]:off[b + 1]]) for b in chk]
    ook, ovst = octx.verify_batch(b''.join(msg[32 * b:32 * b + 32] for b in chk), proofs, nthreads=len(chk),
                                  vseeds=b''.join(vs[32 * b:32 * b + 32] for b in chk))[:2]
    assert [int(ok[b]) for b in chk] == ook and [int(vst[b]) for b in chk] == ovst
    assert set(forged[0::2]) <= set(rejected) <= set(forged)
    pin.free()
    eng.close()


@pytest.fixture(scope='module')
def rccl_stub(tmp_path_factory):
    """tests/rccl_stub: a librccl stand-in whose grouped broadcast is hipMemcpyPeerAsync (it accepts two ranks on one device, the real one does not)"""
    import shutil
    import subprocess
    if shutil.which('g++') is None or not os.path.exists('/opt/rocm/include/hip/hip_runtime.h'):
        pytest.skip('no g++ / HIP headers to build the stub')
    out = tmp_path_factory.mktemp('rccl_stub') / 'librccl_stub.so'
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-shared', '-fPIC', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include',
                           os.path.join(ROOT, 'tests', 'rccl_stub', 'rccl_stub.cpp'), '-o', str(out), '-L/opt/rocm/lib', '-lamdhip64'])
    return str(out)


def test_pool_rccl_branch_runs_through_a_stub_library(rccl_stub, monkeypatch):
    """zk_pool_set_ring's RCCL branch (csrc/api_pool.hip) on the one-GPU tier: ZKATTEST_RCCL_LIB selects tests/rccl_stub, ZKATTEST_RCCL_SAME_DEVICE=1 lets a
    pool of two contexts on device 0 take the branch.  Success path (grouped in-place ncclBroadcast on the contexts' streams, same bytes as one context),
    the three failure paths (ncclCommInitAll, ncclBroadcast, ncclGroupEnd) each ending in peer copies with the reason in zk_pool_last_error, communicators
    torn down after a failed broadcast and never trusted again, every communicator destroyed with the pool."""
    import ctypes
    import zkp_ecdsa_amd as Z
    monkeypatch.setenv('ZKATTEST_RCCL_LIB', rccl_stub)
    monkeypatch.setenv('ZKATTEST_RCCL_SAME_DEVICE', '1')
    monkeypatch.delenv('RCCL_STUB_FAIL', raising=False)
    stub = ctypes.CDLL(rccl_stub)   # the same mapping the pool's dlopen returns: its counters are the pool's
    stub.rccl_stub_reset()
    B, nkeys = 21, 300
    eng, params, (ring, msg, sig, pk, which, seeds) = _engine(5151, nkeys, B)
    ref, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    ring2 = bytes(ring[:32 * 7]) + bytes(ring[32 * 8:32 * 9]) + bytes(ring[32 * 8:])   # another ring (key 7 replaced): the second broadcast must deliver IT
    # ---- success
    pool = Z.Pool([0, 0])
    pool.set_params(*params, 80)
    assert pool.set_ring(ring, nkeys) == 'rccl', pool.last_error()
    assert pool.rccl_library().endswith('librccl_stub.so'), pool.rccl_library()
    assert [stub.rccl_stub_counter(i) for i in range(5)] == [1, 2, 0, 2, 1]   # one init, two communicators, two broadcasts in one group
    got, pst = pool.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert pst == [0] * B and got == ref
    assert pool.set_ring(ring2, nkeys) == 'rccl'        # the communicators are reused
    assert [stub.rccl_stub_counter(i) for i in range(5)] == [1, 2, 0, 4, 2]
    eng.set_ring(ring2, nkeys)
    which2 = [w if w != 7 else 8 for w in which]
    ref2, st2 = eng.prove_batch(msg, sig, pk, which2, seeds=seeds)
    got2, pst2 = pool.prove_batch(msg, sig, pk, which2, seeds=seeds)
    assert pst2 == st2 and got2 == ref2
    pool.close()
    assert stub.rccl_stub_counter(2) == 2               # both communicators destroyed with the pool
    # ---- ncclCommInitAll fails: peer copies, the reason kept, no communicator exists
    for mode, text in (('init', 'ncclCommInitAll failed'), ('broadcast', 'ncclBroadcast of the ring failed'), ('groupend', 'ncclBroadcast of the ring failed')):
        stub.rccl_stub_reset()
        monkeypatch.setenv('RCCL_STUB_FAIL', mode)
        pool = Z.Pool([0, 0])
        pool.set_params(*params, 80)
        assert pool.set_ring(ring, nkeys) == 'peer-copy'
        assert pool.last_error().startswith('rccl:') and text in pool.last_error() and 'stub:' in pool.last_error(), pool.last_error()
        made, gone = stub.rccl_stub_counter(1), stub.rccl_stub_counter(2)
        assert (made, gone) == ((0, 0) if mode == 'init' else (2, 2))     # a failed broadcast tears its communicators down at once
        monkeypatch.delenv('RCCL_STUB_FAIL')
        assert pool.set_ring(ring, nkeys) == 'peer-copy'                     # ... and RCCL is not tried again on this pool
        assert stub.rccl_stub_counter(0) == 1
        got, pst = pool.prove_batch(msg, sig, pk, which, seeds=seeds)
        eng.set_ring(ring, nkeys)
        assert pst == [0] * B and got == ref
        pool.close()
        assert stub.rccl_stub_counter(2) == gone
    # ---- an explicit library that cannot be loaded is not silently replaced by another one
    monkeypatch.setenv('ZKATTEST_RCCL_LIB', rccl_stub + '.missing')
    pool = Z.Pool([0, 0])
    pool.set_params(*params, 80)
    assert pool.set_ring(ring, nkeys) == 'peer-copy' and 'not found' in pool.last_error() and pool.rccl_library() == ''
    pool.close()
    eng.close()


def test_pool_shards_one_call_over_devices_and_matches_single_device():
    """zk_pool_*: ring uploaded once and broadcast, shards proved and verified by one host thread per device.  On a one-GPU box
    the pool holds two contexts on device 0 (same code path: threads, shard arithmetic, device-to-device ring copy)."""
    import torch
    import zkp_ecdsa_amd as Z
    ndev = torch.cuda.device_count()
    ids = [0, 1] if ndev >= 2 else [0, 0]
    B, nkeys = 37, 600
    eng, params, (ring, msg, sig, pk, which, seeds) = _engine(4242, nkeys, B)
    ref, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert st == [0] * B
    pool = Z.Pool(ids)
    assert pool.shard(B, 0) == (0, 18) and pool.shard(B, 1) == (18, 19)
    pool.set_params(*params, 80)
    transport = pool.set_ring(ring, nkeys)
    assert transport in ('rccl', 'peer-copy'), transport
    if ids[0] == ids[1]:
        assert transport == 'peer-copy'   # RCCL refuses one device twice
    else:
        # two distinct devices: ncclBroadcast is what the ring should have travelled by -- but a fallback to peer copies is a CORRECT result; it must
        # name its reason (library not found / init failed / broadcast failed), and that reason is printed instead of turning the tier red
        print('ring transport: %s; librccl: %r; %s' % (transport, pool.rccl_library(), pool.last_error()))
        assert transport == 'rccl' or pool.last_error().startswith('rccl:'), (transport, pool.last_error())
    for i in range(2):
        pool.engine(i).set_chunk(7)
    got, pst = pool.prove_batch(msg, sig, pk, which, seeds=seeds)
    assert pst == [0] * B and got == ref
    octx = _oracle(params, ring, nkeys)
    exp, _ = octx.prove_batch(msg[:32 * 4], sig[:64 * 4], pk[:64 * 4], which[:4], seeds=seeds[:32 * 4], nthreads=4)
    assert got[:4] == exp
    vs = b''.join(hashlib.sha256(b'pool%d' % i).digest() for i in range(B))
    assert pool.verify_batch(msg, got, vseeds=vs) == eng.verify_batch(msg, got, vseeds=vs) == ([1] * B, [0] * B)
    forged = list(got)
    forged[20] = forged[20][:-9] + bytes([forged[20][-9] ^ 1]) + forged[20][-8:]
    ok, vst = pool.verify_batch(msg, forged, vseeds=vs)
    assert ok == [1] * 20 + [0] + [1] * 16
    # page-locked output: every shard's DMA lands in its own region; (off, len) describe the proofs, with a gap between shards
    cap = 2 * ((eng.proof_max_size() * 19 + 255) & ~255)
    pin = Z.PinnedBuffer(cap)
    _, off, ln, pst = pool.prove_batch_raw(msg, sig, pk, which, seeds, pin, cap)
    assert [bytes(pin.view[off[b]:off[b] + ln[b]]) for b in range(B)] == ref
    assert off[18] == cap // 2 and off[17] + ln[17] < off[18]
    _, ok, vst = pool.verify_batch_raw(msg, pin, off, ln, B, vseeds=vs)
    assert list(ok) == [1] * B
    # the pool's own page-locked buffer (per-shard regions first touched next to their devices) behaves like zk_host_alloc's
    npin = Z.PinnedBuffer(cap, pool=pool)
    _, off3, ln3, pst = pool.prove_batch_raw(msg, sig, pk, which, seeds, npin, cap)
    assert [bytes(npin.view[off3[b]:off3[b] + ln3[b]]) for b in range(B)] == ref and list(off3) == list(off)
    assert len(pool.shard_ms()) == 2 and all(ms > 0 for ms in pool.shard_ms()) and pool.numa_node(0) >= -1
    npin.free()
    # a layout with a hole inside a shard is refused, not misread
    off2 = (C.c_uint64 * B)(*off)
    off2[5] += 4
    with pytest.raises(Z.ZkError) as e:
        pool.verify_batch_raw(msg, pin, off2, ln, B, vseeds=vs)
    assert e.value.status == 14
    # one device listed once behaves like the plain context
    solo = Z.Pool([0])
    solo.set_params(*params, 80)
    assert solo.set_ring(ring, nkeys) == 'single'
    assert solo.prove_batch(msg, sig, pk, which, seeds=seeds)[0] == ref
    solo.close(), pool.close(), pin.free(), eng.close()


def _bench(args, env=None, timeout=900):
    import json
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


def test_baseline_config5_full_per_gpu_shard_131072_proofs_over_ring_2_20():
    """BASELINE configs[4] (verifySignatureList, batch 2^20, ring 2^20, 8 GPUs) at the scale ONE of its eight GPUs sees: 131 072
    proofs over a ring of 2^20 keys, streamed through HBM in slabs (bench.py --mode verify, the very command the driver's scaling
    run executes per rank).  bench.py itself asserts that every honest proof is accepted, that exactly the three planted
    forgeries of the first slab are rejected and that no proof failed to be made; verdict parity with the oracle at this ring
    size is test_baseline_config5_shape_ring_2_20_verify_4096."""
    line = _bench(['--mode', 'verify', '--batch', '131072', '--ring', '1048576', '--steps', '1', '--warmup', '0'])
    assert line['metric'].startswith('verifySignatureList') and line['n_gpus'] == 1
    assert line['accepted'] == line['of'] == 131072 and line['planted_forgeries_rejected'] == 3
    assert '131072 proofs per rank' in line['config']['workload'] and 'ring=1048576' in line['config']['workload']
    assert line['value'] > 0


def test_the_references_own_bench_ring_of_100001_keys_runs_on_the_key_tables():
    """bench/zkpAttestList.bench.ts:38-68 proves over 1 + 100 000 keys (padded to 2^17).  bench.py --ring 100001: the per-key tables cover
    that ring too (35 GB at 2^17 keys; they stopped at 2^16 until round 4), every proof of the CPU sample is diffed byte for byte against
    the oracle inside bench.py, every proof verifies, and the line carries the small-batch latency table."""
    line = _bench(['--ring', '100001', '--batch', '8192', '--chunk', '4096', '--lanes', '2', '--verify-chunk', '4096', '--steps', '1', '--warmup', '1',
                   '--cpu-sample', '8', '--host-io', '0', '--json-sample', '0'], timeout=1500)
    assert 'ring=100001 keys (n=17)' in line['config']['workload'] and 'per-key tables' in line['config']['workload']
    assert line['key_table_proofs_last_chunk'] == 4096          # zk_test_counter(ctx, 1): every proof of the last chunk went through them
    assert line['failed_proofs'] == 0 and line['cpu_baseline']['checked_bit_exact'] == 8
    assert line['verify']['accepted'] == line['verify']['of'] == 8192
    lat = line['latency']['rings']
    assert set(lat) == {'100001', '1024'} and set(lat['1024']) == {'1', '8', '64', '512'} and set(lat['100001']) == {'1', '8', '64', '512', '4096'}
    assert line['latency_ms_b1'] == lat['100001']['1']['prove_ms'] > 0


def test_pool_prove_with_device_resident_output_equals_the_host_buffer_call():
    """zk_pool_prove_batch_device (and zk_prove_batch with `out` in HBM): the proofs stay on each shard's device; copied back by hand they are
    the bytes zk_pool_prove_batch delivers."""
    import torch
    import zkp_ecdsa_amd as Z
    B, nkeys = 700, 1024
    pool = Z.Pool([0, 0])
    e0 = pool.engine(0)
    for i in range(2):
        pool.engine(i).set_comb_bits(16)
        pool.engine(i).set_chunk(256)
    pool.set_params(*e0.synth_params(93), 80)
    ring, msg, sig, pk, which, seeds = e0.synth_workload(93, nkeys, B)
    pool.set_ring(ring, nkeys)
    ref, st = pool.prove_batch(msg, sig, pk, which, seeds)
    assert not any(st)
    cap = e0.proof_max_size() * 350 * 7 // 10 + (4 << 20)
    t = [torch.zeros(cap, dtype=torch.uint8, device='cuda:0') for _ in range(2)]
    torch.cuda.synchronize()
    _, off, ln, st = pool.prove_batch_device_out(msg, sig, pk, which, seeds, [x.data_ptr() for x in t], [cap, cap])
    assert not any(st)
    for i in range(2):
        first, cnt = pool.shard(B, i)
        raw = t[i].cpu().numpy().tobytes()
        assert off[first] == 0                                  # offsets are relative to the shard's own buffer
        for k in range(first, first + cnt):
            assert raw[off[k]:off[k] + ln[k]] == ref[k], (i, k)
    # the pool's own allocator for such buffers
    d = [pool.device_alloc(i, cap) for i in range(2)]
    _, off2, ln2, st2 = pool.prove_batch_device_out(msg, sig, pk, which, seeds, d, [cap, cap])
    assert list(off2) == list(off) and list(ln2) == list(ln) and not any(st2)
    for i in range(2):
        pool.device_free(i, d[i])
    # a buffer that is too small fails that call, not the process
    with pytest.raises(Z.ZkError) as e:
        pool.prove_batch_device_out(msg, sig, pk, which, seeds, [x.data_ptr() for x in t], [cap, 4096])
    assert e.value.status == 12
    pool.close()


def test_bench_pool_mode_runs_the_librarys_own_multi_gpu_path():
    """bench.py --pool: one process, zk_pool over the listed devices (two contexts on device 0 here; the visible devices on a
    multi-GPU box), page-locked NUMA-placed buffers, ring transport reported, every proof verified, forgeries rejected, shard 0
    byte for byte against the oracle."""
    import torch
    ndev = torch.cuda.device_count()
    devs = ','.join(str(i) for i in range(min(ndev, 8))) if ndev >= 2 else '0,0'
    line = _bench(['--pool', '--pool-devices', devs, '--batch', '2048', '--ring', '4096', '--comb-bits', '16', '--host-io-chunk', '1024',
                   '--host-io-verify-chunk', '1024', '--steps', '1', '--warmup', '1', '--cpu-sample', '4'])
    G = len(devs.split(','))
    assert line['n_gpus'] == G and line['accepted'] == line['of'] == 2048 * G and line['planted_forgeries_rejected'] == G
    assert line['ring_transport'] == ('rccl' if ndev >= 2 else 'peer-copy')
    assert len(line['prove_shard_ms_per_step'][0]) == G and line['cpu_baseline']['checked_bit_exact'] == 4
    assert line['value_pcie_inclusive'] == line['value'] > 0


def test_bench_eight_ranks_on_one_gpu_dry_run():
    """The driver's `--gpus 8` launch line, with all eight ranks on this one GPU (gloo group, small tables): rendezvous, ring
    broadcast, per-rank seeds, max-over-ranks timing and the rank-0 JSON line at world size 8."""
    import subprocess
    import json
    res = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'smoke_multirank.sh'), 'prove8'], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 8 and line['failed_proofs'] == 0 and line['verify']['accepted'] == line['verify']['of']
    assert line['cpu_baseline']['checked_bit_exact'] >= 1


def test_tapered_chunk_plan_of_the_host_pointer_calls_keeps_the_bytes():
    """Page-locked buffers switch the host-pointer calls to the tapered plan (half-sized first chunk, shrinking tail); uniform
    chunks, pageable buffers and the device-pointer call must give the same bytes and verdicts."""
    import zkp_ecdsa_amd as Z
    B, nkeys, sec = 9000, 16384, 20   # ring >= batch: every proof's key is in the ring
    eng, params, (ring, msg, sig, pk, which, seeds) = _engine(99, nkeys, B, sec=sec)
    cap = eng.proof_max_size() * B
    pin = Z.PinnedBuffer(cap)
    digests = []
    for taper, chunk in ((1, 4096), (0, 4096), (1, 3000)):
        eng.set_host_taper(taper), eng.set_chunk(chunk)
        C.memset(pin.ptr, 0x5A, 1 << 20)
        _, _, off, st = eng.prove_batch_host_raw(msg, sig, pk, which, seeds, out=pin)
        assert not any(st) and off[0] == 0
        digests.append(hashlib.sha256(bytes(pin.view[:off[B]])).hexdigest() + ':%d' % off[B])
        _, ok, vst = eng.verify_batch_host_raw(msg, pin, off, B)
        assert sum(ok) == B and not any(vst)
    page = (C.c_uint8 * off[B])()
    _, _, off2, st = eng.prove_batch_host_raw(msg, sig, pk, which, seeds, out=page)
    digests.append(hashlib.sha256(bytes(page)).hexdigest() + ':%d' % off2[B])
    assert len(set(digests)) == 1, digests
    # a call that is ONE chunk of at most 8192 proofs on a page-locked buffer slices its PointAdd phase by 512 (<= 2048 proofs) or 1024 (api.hip: stage2): same bytes
    eng.set_host_taper(1), eng.set_chunk(4096)
    for nb in (3000, 1500, 700):
        C.memset(pin.ptr, 0x5A, 1 << 20)
        _, _, offn, st = eng.prove_batch_host_raw(msg[:32 * nb], sig[:64 * nb], pk[:64 * nb], which[:nb], seeds[:32 * nb], out=pin)
        assert not any(st) and list(offn[:nb + 1]) == list(off2[:nb + 1])
        assert bytes(pin.view[:offn[nb]]) == bytes(page[:off2[nb]]), nb
    octx = _oracle(params, ring, nkeys, sec)
    pick = [0, 2047, 2048, 4095, 8999]
    exp, _ = octx.prove_batch(b''.join(msg[32 * b:32 * b + 32] for b in pick), b''.join(sig[64 * b:64 * b + 64] for b in pick),
                              b''.join(pk[64 * b:64 * b + 64] for b in pick), [which[b] for b in pick],
                              seeds=b''.join(seeds[32 * b:32 * b + 32] for b in pick), nthreads=5)
    assert [bytes(page[off2[b]:off2[b + 1]]) for b in pick] == exp
    pin.free()
    eng.close()


def test_bench_starts_its_own_ranks_from_a_bare_gpus_line():
    """`python3 bench.py --gpus 2` with no launcher in the environment (what a driver types): bench.py re-executes itself under torch.distributed.run, one
    rank per GPU, and rank 0 prints ONE line with n_gpus = 2.  Both ranks share cuda:0 here (ZK_BENCH_ONE_DEVICE, gloo group)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['ZK_BENCH_ONE_DEVICE'] = '1'
    import json
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--batch', '2048', '--ring', '4096', '--chunk', '1024', '--lanes', '2',
                          '--verify-chunk', '1024', '--comb-bits', '16', '--host-io', '0', '--cpu-sample', '2', '--json-sample', '0', '--latency', '0'],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert res.returncode == 0 and len(lines) == 1, res.stdout[-2000:] + res.stderr[-3000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['failed_proofs'] == 0 and d['verify']['accepted'] == d['verify']['of'] and d['verify']['steps'] >= 5
    assert 're-executing' in res.stderr


@pytest.mark.parametrize('mode', ['prove', 'verify'])
def test_bench_two_ranks_on_one_gpu(mode):
    """The N > 1 path of bench.py with the ENGINE on the device: two ranks under torch.distributed.run, both on cuda:0 over a gloo
    group (tools/smoke_multirank.sh; RCCL refuses two ranks on one GPU).  Ring broadcast, per-rank seeds, max-over-ranks timing and
    the rank-0 JSON line, which must carry cpu_baseline at N > 1."""
    import json
    import subprocess
    r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'smoke_multirank.sh'), mode], capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['value'] > 0
    if mode == 'prove':
        assert d['scaling'] == 'weak' and d['failed_proofs'] == 0
        assert d['cpu_baseline'] and d['cpu_baseline']['value'] > 0 and d['cpu_baseline']['checked_bit_exact'] >= 1
        assert d['verify']['accepted'] == d['verify']['of']
    else:
        assert d['scaling'] == 'strong' and d['accepted'] == d['of'] and d['planted_forgeries_rejected'] == 3
