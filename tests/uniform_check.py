"""Helper of tests/test_gpu_prove.py::test_uniform_control_flow_build_makes_the_same_bytes: run in a subprocess with ZKATTEST_LIB pointing at the library
under test; proves a fixed small workload twice -- 40 proofs in one call (the one-lane kernels) and 3 of them in a second call (the wide kernels of small
chunks) -- and prints the SHA-256 of all proof bytes and the per-family GPU time of the large call."""
import hashlib
import json
import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkp_ecdsa_amd as Z

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
eng = Z.Engine(0)
eng.set_timing(1)
eng.set_comb_bits(16)
eng.set_params(*eng.synth_params(99), 80)
ring, msg, sig, pk, which, seeds = eng.synth_workload(99, 2048, B)
eng.set_ring(ring, 2048)
eng.set_chunk(B), eng.set_lanes(1)
big, st = eng.prove_batch(msg, sig, pk, which, seeds=seeds)
assert not any(st)
tot, fam = eng.last_timing()
small, st = eng.prove_batch(msg[:96], sig[:192], pk[:192], which[:3], seeds=seeds[:96])
assert not any(st) and small == big[:3]
print(json.dumps({'lib': Z.LIB_PATH, 'sha256': hashlib.sha256(b''.join(big)).hexdigest(), 'gpu_ms': round(tot, 2),
                  'families_ms': {k: round(v, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:6]}}))
