"""Shared helpers of bench.py and bench_modes.py: measured peaks, the workload's operation counts, per-rank seeds, the CPU baseline
(the oracle, timed -- the only place the bench touches oracle/ besides the post-timing spot check)."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

# measured on MI355X with tools/valu_peak.hip (profiles/r01_valu_peak_microbench.txt, r02_...): v_mad_u64_u32 chip-wide issue
# rate at 16 independent accumulators x 8 waves/SIMD (4.2 cycles per wave-instruction; 39.3 T/s would be 16 lanes/clk at 2.4 GHz);
# 36.66 and 37.11 T/s were measured on two boxes of the pool in round 1 (37.28-37.36 T/s sustained over 2 s in round 2); the
# denominator stays 37.11 so that the fractions of the two rounds compare
VALU_MAD_PEAK_TOPS = 37.11
# the same instruction at the kernel's OWN parallelism: 4 lock-step chains per wave x 2 waves per SIMD (212 VGPRs) = 8 independent
# accumulator chains per SIMD issue at 5.7 cycles per wave-instruction (tools/valu_peak.hip "NACC= 4 waves/SIMD=2",
# profiles/r03_valu_peak_microbench.txt: 27.83 T/s; 16 chains 31.9, 64 chains 36.3, 128 chains 37.2)
VALU_MAD_8CHAIN_TOPS = 27.83
HBM_PEAK_GBPS = 8000.0
# multiplier-pipe instructions (v_mad_u64_u32 + v_mul_lo_u32) per Tom-field Montgomery product: 1224 + 72 per table addition
# of 8 products in k_tom_commit (tools/isa_blocks.py; nominal 171 = 81 + 81 + 9, the modulus limb that is zero costs nothing);
# PMC (profiles/r03_pmc_summary.txt; r02: 36 674): 36 769 VALU wave-instructions per unpaired commitment of 163 products on average = 225
# instructions per product, 230 in the paired kernel (round 1: 239)
MACS_PER_MODMUL = 162


# k_exp_commit_kt / k_exp_commit (csrc/k_p256.hip): multiplier instructions of ONE mixed complete P-256 addition in the shipped ISA (tools/isa_blocks.py:
# 1872 v_mad_u64_u32 + 99 v_mul_lo_u32 for its 11 products) and the additions per lane: key-table path 12 (comb of G, first entry loaded) + 33 (key
# table) + 13 (comb of h); per-proof table of R: 43 complete additions (12 products each, priced like 12/11 mixed ones) + 13
EXP_ADD_MACS = 1872 + 99
EXP_KT_ADDS = 12 + 33 + 13
EXP_RTAB_ADDS = round(43 * 12 / 11) + 13


def tom_commit_modmuls(comb_bits):
    """executed per commitment: 2 x ceil(256/W) table additions of a W-bit comb, 8 modmuls each"""
    return 2 * ((256 + comb_bits - 1) // comb_bits) * 8


TOM_COMMIT_NOMINAL = 4064      # reference: 256 dbl + 160 add (src/curves/group.ts:97-132)
TOM_COMMIT_BYTES = 2 * 36 + 3 * 36  # algorithmic HBM bytes per commitment: read (v, r), write (X, Y, Z)
# PMC passes (profiles/r05_pmc_summary.txt (r04_... before it): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate runs at the bench's own shape, batch 65536 in
# chunks of 22016, tools/repro_profiles.sh), bytes per commitment through the L2's memory-side port, keyed by comb width.  24 bits (128-byte table
# entries, 47 GB of tables): k_tom_commit + k_tom_commit_pairs fetch (7.673e7 + 5.016e7) KB for 103.88 M commitments = 1251 B raw, i.e. 2 x 1251 B
# (gfx950 tallies 16-byte-per-lane loads at half their bytes, MI355X_MICROARCH.md section HBM; 20.3 gathers x 128 B = 2600 B expected; round 3's pass at
# batch 16384 read 2 x 1310) + 111 B written.  16 bits (112-byte entries, 235 MB; round-1 pass): 3238 B (raw) + 111 B.
TOM_COMMIT_PMC_BYTES = {24: 2502 + 111, 16: 3238 + 111}
# whole verify step (65 536 proofs, 2 chunks of 32 768, one lane): 32 * sum SQ_ACTIVE_INST_VALU / (1024 SIMDs * sum GRBM_GUI_ACTIVE) over every kernel of the step
# (tools/pmc_families.py --verify, tools/pmc_verify.sh)
VERIFY_WHOLE_STEP_SIMD_BUSY = 0.606
VERIFY_PMC_SOURCE = 'profiles/r06_pmc_families_verify.txt (separate rocprofv3 --pmc passes over one verify step; a constant of bench.py, NOT measured in this run)'
PMC_SOURCE = 'profiles/r05_pmc_summary.txt (separate rocprofv3 --pmc passes at batch 65536, chunk 22016; constants of bench.py, NOT measured in this run)'
# same passes, SQ counters at 24 bits: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES per wave at 2 waves per SIMD over both kernels, (3.251e10 + 2.157e10) /
# (6.678e10 + 4.226e10) (VALU pipe busy 99 % of the time), SQ_WAIT_INST_ANY 0.377, SQ_WAIT_ANY (memory) 0.117
TOM_COMMIT_VALU_ACTIVE_PER_WAVE = {24: 0.496}
DEFAULT_COMB_BITS = 24


def rank_seeds(base_seeds: bytes, rank: int) -> bytes:
    """Per-rank RNG seeds: rank 0 keeps the synthetic seeds, rank r > 0 re-keys them (distinct proofs, same statements)."""
    if rank == 0:
        return base_seeds
    out = bytearray()
    tag = b'rank' + (rank.to_bytes(4, 'big') if rank < (1 << 32) else rank.to_bytes(12, 'big'))
    for i in range(0, len(base_seeds), 32):
        out += hashlib.sha256(tag + base_seeds[i:i + 32]).digest()
    return bytes(out)


def nominal_modmuls(n_log2, z=40):
    """Reference-algorithm modular multiplications per proof (SURVEY.md section 8(d) / BASELINE.md section 2)."""
    wt = (162 + 26 * z + 4 * n_log2) * 4064 + 8 * z * 3184
    wq = (163 + z) * 4448 + 5568
    ring = 2 * (1 << n_log2) * n_log2
    return wt, wq, ring


def host_cores():
    """CPUs this process may actually use: min(affinity, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return max(1, min(n, 256))


def cpu_baseline(nh, tg, th, ring, nkeys, msg, sig, pk, which, seeds, sec, budget_proofs, vseeds=None):
    """Oracle (C restatement, reference-faithful algorithms) on this box's host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import coracle as CO
    nthreads = host_cores()
    n = min(len(which), budget_proofs)
    octx = CO.OracleCtx(nh, tg, th, sec)
    octx.set_ring(ring, nkeys)
    t0 = time.time()
    proofs, st = octx.prove_batch(msg[:32 * n], sig[:64 * n], pk[:64 * n], which[:n], seeds=seeds[:32 * n], nthreads=nthreads)
    dt = time.time() - t0
    assert all(s == 0 for s in st)
    rec = {'value': n / dt, 'unit': 'proofs/s', 'cores': nthreads, 'kind': 'port',
           'sample': '%d proofs of the same workload (ring=%d keys, secLevel %d), %d threads, %.1f s wall' % (n, nkeys, sec, nthreads, dt)}
    if vseeds is not None:   # the verify half (bench/zkpAttestList.bench.ts:56-62 times verifySignatureList beside prove): the same sample, the GPU's verifier seeds
        t0 = time.time()
        ok, vst = octx.verify_batch(msg[:32 * n], proofs, nthreads=nthreads, vseeds=vseeds[:32 * n])
        vdt = time.time() - t0
        rec['verify'] = {'value': n / vdt, 'unit': 'verifies/s', 'cores': nthreads, 'kind': 'port', 'accepted': int(sum(ok)), 'of': n,
                         'sample': 'verifySignatureList over the same %d proofs with the verifier seeds of the GPU run, %d threads, %.2f s wall' % (n, nthreads, vdt)}
        rec['_verify_verdicts'] = (ok, vst)
    return rec, proofs


def v8_bigint_indicator():
    """Optional: the plain-JS BigInt restatement oracle/js/zkattest_ref.js proves and verifies
    one golden proof at ring = 6 keys padded to 8, secLevel 80 (the shape of BASELINE configs[0] and of the reference's own
    test) on whatever `node` the box has -- an approximation of `npm run bench`, which needs Node >= 24 and cannot run here."""
    import shutil
    import subprocess
    if shutil.which('node') is None:
        return None
    try:
        p = subprocess.run(['node', os.path.join(ROOT, 'oracle', 'js', 'zkattest_ref.js'), 'bench', os.path.join(ROOT, 'tests', 'golden', 'golden.json'),
                            'ring6_sec80'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        rec = json.loads(p.stdout.decode().strip().splitlines()[-1])
    except Exception as e:  # the indicator must never break the bench line
        return {'error': repr(e)[:200]}
    return {'prove_s': rec['prove_ms'] / 1e3, 'verify_s': rec['verify_ms'] / 1e3, 'proofs_per_s': round(1e3 / max(rec['prove_ms'], 1), 3),
            'node': rec['node'], 'bytes_match_golden': bool(rec['sha256_ok']), 'verified': bool(rec['verified']), 'threads': 1,
            'workload': 'one proof, ring = 6 keys padded to 8, secLevel 80 (tests/golden/golden.json: ring6_sec80)',
            'note': 'approximation of `npm run bench` (V8 BigInt, this build\'s JS restatement); not the cpu_baseline value'}


