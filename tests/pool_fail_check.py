"""Run by tests/test_gpu_stream.py::test_a_pool_submit_that_fails_half_way_leaves_the_older_jobs_waitable in its own process with
ZKATTEST_LIB = lib/libzkattest_hip_testhooks.so (the only build with the fault-injection entry point; one process holds one build)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import zkp_ecdsa_amd as Z
    B, nkeys = 600, 1024
    pool = Z.Pool([0, 0])
    e0 = pool.engine(0)
    params = e0.synth_params(92)
    for i in range(2):
        pool.engine(i).set_comb_bits(16)
        pool.engine(i).set_chunk(128)
    pool.set_params(*params, 80)
    ring, msg, sig, pk, which, seeds = e0.synth_workload(92, nkeys, B)
    pool.set_ring(ring, nkeys)
    cap = 2 * ((e0.proof_max_size() * 300 * 7 // 10 + (4 << 20)) & ~255)
    cut = lambda a, b: (msg[32 * a:32 * b], sig[64 * a:64 * b], pk[64 * a:64 * b], which[a:b], seeds[32 * a:32 * b])
    pin_ref = Z.PinnedBuffer(cap, pool=pool)
    _, off, ln, st = pool.prove_batch_raw(*cut(0, 600), pin_ref, cap)
    ref = [bytes(pin_ref.view[off[k]:off[k] + ln[k]]) for k in range(600)]
    pins = [Z.PinnedBuffer(cap, pool=pool) for _ in range(3)]
    # (1) idle pipeline: slot 0's shard is already running when slot 1 refuses
    pool.test_fail_submit(1)
    with pytest.raises(Z.ZkError) as e:
        pool.prove_submit(*cut(0, 600), pins[0], cap)
    assert e.value.status == 15 and 'injected' in str(e.value)
    # (2) an older job in flight on both contexts when the next submit fails at slot 1
    t0 = pool.prove_submit(*cut(0, 600), pins[0], cap)
    pool.test_fail_submit(1)
    with pytest.raises(Z.ZkError):
        pool.prove_submit(*cut(0, 600), pins[1], cap)
    t2 = pool.prove_submit(*cut(0, 600), pins[2], cap)       # the queue is intact: another job goes in behind the survivor
    for t, pin in ((t0, pins[0]), (t2, pins[2])):
        off, ln, st = pool.prove_wait(t)
        assert not any(st)
        assert [bytes(pin.view[off[k]:off[k] + ln[k]]) for k in range(600)] == ref
    # (3) the same for the verifier
    off0, ln0, _ = off, ln, st
    vs = os.urandom(32 * 600)
    v0 = pool.verify_submit(msg, pins[0], off0, ln0, 600, vs)
    pool.test_fail_submit(1)
    with pytest.raises(Z.ZkError):
        pool.verify_submit(msg, pins[2], off0, ln0, 600, vs)
    v2 = pool.verify_submit(msg, pins[2], off0, ln0, 600, vs)
    for v in (v0, v2):
        ok, vst = pool.verify_wait(v)
        assert sum(ok) == 600 and not any(vst)
    # the synchronous calls work again (nothing is left queued)
    _, off, ln, st = pool.prove_batch_raw(*cut(0, 600), pins[1], cap)
    assert [bytes(pins[1].view[off[k]:off[k] + ln[k]]) for k in range(600)] == ref
    for p_ in pins + [pin_ref]:
        p_.free()
    pool.close()


if __name__ == '__main__':
    main()
    print('pool_fail_check ok')
