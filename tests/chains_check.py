"""Run by tests/test_gpu_prove.py::test_one_lane_and_cooperative_chains_make_the_same_bytes, once with ZKATTEST_ONE_LANE_CHAINS and once without: one proof per
call (the cooperating-wave kernels of the small-call paths), a 3 000-proof call (batched checks: k_msm_red_last, k_pm_final), tampered proofs in both; prints a
digest of the proof bytes and every verdict / status."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkp_ecdsa_amd as Z

os.environ['ZKATTEST_P256_BATCH'] = '1024'   # the cross-proof P-256 pass (k_pmsm.hip) at this batch size too
eng = Z.Engine(0)
eng.set_comb_bits(16)
eng.set_params(*eng.synth_params(606), 80)
B = 3000
ring, msg, sig, pk, which, seeds = eng.synth_workload(606, 4096, B)
eng.set_ring(ring, 4096)
eng.set_chunk(4096)
eng.set_batch_verify(1024)       # the cross-proof Tom-256 check (k_msm.hip) from 1 024 proofs on
proofs, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
assert not any(st)
h = hashlib.sha256()
for p in proofs:
    h.update(p)
verdicts = []
vs = b''.join(hashlib.sha256(b'cc' + i.to_bytes(4, 'big')).digest() for i in range(B))
ok, vst = eng.verify_batch(msg, proofs, vseeds=vs)
verdicts.append([sum(ok), sum(vst)])
bad = list(proofs)
for k in (5, 1500, 2999):
    bad[k] = bad[k][:-9] + bytes([bad[k][-9] ^ 2]) + bad[k][-8:]
bad[77] = bad[77][:2000] + bytes([bad[77][2000] ^ 1]) + bad[77][2001:]
ok, vst = eng.verify_batch(msg, bad, vseeds=vs)
verdicts.append([[i for i in range(B) if not ok[i]], [vst[i] for i in range(B) if vst[i]]])
# one proof per call, honest and tampered, several of them
for k in (0, 5, 77, 1500):
    one, st1 = eng.prove_batch(msg[32 * k:32 * k + 32], sig[64 * k:64 * k + 64], pk[64 * k:64 * k + 64], which[k:k + 1], seeds=seeds[32 * k:32 * k + 32])
    assert st1 == [0] and one[0] == proofs[k]
    for cand in (proofs[k], bad[k]):
        ok1, vst1 = eng.verify_batch(msg[32 * k:32 * k + 32], [cand], vseeds=vs[32 * k:32 * k + 32])
        verdicts.append([ok1, vst1])
eng.close()
print(json.dumps({'one_lane': 'ZKATTEST_ONE_LANE_CHAINS' in os.environ, 'sha256': h.hexdigest(), 'verdicts': verdicts}))
