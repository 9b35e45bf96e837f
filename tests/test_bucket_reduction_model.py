"""Index arithmetic of the verifier's bucket reductions, restated over the integers (a "point" is an int, addition is +, doubling is * 2) and held against
sum_d d * B_d computed directly.  No GPU, no library: this pins the SCHEMES of zkp-ecdsa_amd/csrc/k_msm.hip (k_msm_red1 / k_msm_redk / k_msm_red_last: four levels
of sixteen with the plain sums of the lower levels' F carried along) and k_pmsm.hip (k_pm_reduce: per-thread running sums, then a tree of (F, G) segments whose
joins double G_R log2(m) times) for both shapes of each pass; the kernels themselves are held to the oracle by tests/test_gpu_verify.py."""
import random

import pytest

R = 16


def red1(b):
    """thread r: F = sum_j j * B[16 r + j], G = sum_j B[16 r + j] by running sums from the top (k_msm_red1)"""
    F, G = [], []
    for r in range(len(b) // R):
        run = acc = 0
        for j in range(R - 1, 0, -1):
            run += b[R * r + j]
            acc += run
        run += b[R * r]
        F.append(acc), G.append(run)
    return F, G


def redk(Gin, Pin, r):
    """one level (k_msm_redk): role 0 -> weighted and plain sum of r entries of G; role k -> plain sum of the k-th carried array"""
    nout = len(Gin) // r
    Fout, Gout, Pout = [], [], [[] for _ in Pin]
    for t in range(nout):
        run = acc = 0
        for j in range(r - 1, 0, -1):
            run += Gin[t * r + j]
            acc += run
        run += Gin[t * r]
        Fout.append(acc), Gout.append(run)
        for k, p in enumerate(Pin):
            Pout[k].append(sum(p[t * r:(t + 1) * r]))
    return Fout, Gout, Pout


@pytest.mark.parametrize('C', [16, 13])
def test_four_levels_of_sixteen_give_the_weighted_bucket_sum(C):
    rng = random.Random(C)
    nb = 1 << C
    b = [rng.randrange(1 << 40) if rng.random() < 0.7 else 0 for _ in range(nb)]
    b[0] = rng.randrange(1 << 40)   # bucket 0 is whatever the bucket kernel left there times 0
    want = sum(d * v for d, v in enumerate(b))
    F1, G1 = red1(b)
    F2, G2, (P2a,) = redk(G1, [F1], R)
    F3, G3, (P3a, P3b) = redk(G2, [F2, P2a], R)
    n3 = len(G3)
    assert n3 == (16 if C == 16 else 2)
    F4, G4, (P4a, P4b, P4c) = redk(G3, [F3, P3a, P3b], n3)
    assert len(F4) == 1 and G4[0] == sum(b)
    t = F4[0]
    for carried in (P4a[0], P4b[0], P4c[0]):   # k_msm_red_last: sum of all F3, of all F2, of all F1
        t = 16 * t + carried
    assert t == want


@pytest.mark.parametrize('C', [13, 10])
def test_running_sums_then_a_tree_of_segments(C):
    """k_pm_reduce: 256 threads, PER = 2^C / 256 buckets each; level o joins the segments of threads t and t + o (t a multiple of 2 o): F += F_R + m * G_R with m = PER * o
    buckets per segment (log2 m doublings), G += G_R"""
    rng = random.Random(C)
    nb, T = 1 << C, 256
    per = nb // T
    b = [rng.randrange(1 << 40) for _ in range(nb)]
    F, G = [], []
    for t in range(T):
        run = acc = 0
        for j in range(per - 1, 0, -1):
            run += b[t * per + j]
            acc += run
        run += b[t * per]
        F.append(acc), G.append(run)
    logm = per.bit_length() - 1
    assert 1 << logm == per
    o = 1
    while o < T:
        for t in range(0, T, 2 * o):
            gr = G[t + o]
            G[t] += gr
            for _ in range(logm):
                gr *= 2
            F[t] = F[t] + F[t + o] + gr
        o, logm = 2 * o, logm + 1
    assert F[0] == sum(d * v for d, v in enumerate(b)) and G[0] == sum(b)


def test_digits_of_the_p256_pass_cover_the_scalars_it_is_given():
    """k_pm_pack: C-bit digits over ceil(134 / C) windows; the rho_j are below 2^128 and SL, a sum of at most 20 of them, below 2^133: the digits reassemble the scalar, and
    the bound the kernel checks (bits from nw * C on are zero) holds for every such scalar"""
    rng = random.Random(7)
    for C in (13, 10):
        nw = (134 + C - 1) // C
        assert 128 < nw * C < 160
        for _ in range(200):
            s = sum(rng.randrange(1 << 128) for _ in range(rng.randrange(0, 21)))
            assert s >> (nw * C) == 0
            digs = [(s >> (C * w)) & ((1 << C) - 1) for w in range(nw)]
            assert sum(d << (C * w) for w, d in enumerate(digs)) == s
