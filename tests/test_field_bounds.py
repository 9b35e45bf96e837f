"""CPU checks of the bounds the device field arithmetic (zkp-ecdsa_amd/csrc/field.h) relies on: interval arithmetic over
the generated constants, no GPU.  The device code tracks magnitudes in the TYPE (Fe<M, K>: normalised limbs, value < K*M);
these tests show that the worst case allowed by those types cannot overflow a 32-bit limb or a 64-bit column."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, NL = 30, 9
MASK = (1 << W) - 1
KCAP = 512


def _consts():
    txt = open(os.path.join(ROOT, 'zkp-ecdsa_amd', 'csrc', 'consts_gen.h')).read()
    out = {}
    for name, body in re.findall(r'struct Mod(\w) \{(.*?)\n\};', txt, re.S):
        d = {}
        for key, vals in re.findall(r'uint32_t (\w+)\[9\] = \{([^}]*)\}', body):
            d[key] = [int(v.strip().rstrip('u'), 16) for v in vals.split(',')]
        d['n0'] = int(re.search(r'n0 = (0x[0-9a-f]+)u', body).group(1), 16)
        d['kmax'] = int(re.search(r'kmax = (\d+)', body).group(1))
        out[name] = d
    return out


def _val(limbs):
    return sum(x << (W * i) for i, x in enumerate(limbs))


def _top(K, M):
    """largest top limb of a normalised value < K*M"""
    return (K * M - 1) >> (W * (NL - 1))


def test_generated_constants_are_consistent():
    for name, d in _consts().items():
        M = _val(d['mod'])
        assert all(x <= MASK for x in d['mod'])
        assert (M * d['n0'] + 1) % (1 << W) == 0                      # n0 = -1/M mod 2^30
        assert _val(d['one']) == (1 << (W * NL)) % M
        assert _val(d['r2']) == pow(1 << (W * NL), 2, M)
        assert d['kmax'] == (1 << (W * NL)) // M
        assert _top(KCAP, M) < 1 << 28                                 # field.h: KCAP keeps the top limb < 2^28


def test_subtraction_constants_never_underflow_or_overflow_a_limb():
    """operator-: r_i = a_i + S_i - b_i;  fe_sub2: r_i = a_i + S_i + lend_i - b_i - c_i  (uint32 arithmetic)."""
    for name, d in _consts().items():
        M = _val(d['mod'])
        for C in (4, 8, 16, 32, 64, 128, 256):
            S = d['sub%d' % C]
            assert _val(S) == C * M
            # one subtrahend of magnitude Kb < C: limbs 0..7 of b are <= 2^30 - 1, its top limb <= top(Kb*M)
            assert all(S[i] >= MASK for i in range(NL - 1))
            assert S[NL - 1] >= _top(C - 1, M)
            assert all(MASK + S[i] < 1 << 32 for i in range(NL))       # a_i + S_i fits 32 bits
            # two subtrahends with Kb + Kc < C (fe_sub2): lend once more along the chain, same value
            lend = [1 << W] + [MASK] * (NL - 2) + [-1]
            S2 = [S[i] + lend[i] for i in range(NL)]
            assert _val(S2) == C * M
            assert all(S2[i] >= 2 * MASK for i in range(NL - 1))
            assert S2[NL - 1] >= _top(C - 1, M) + 1                    # b_8 + c_8 <= top((Kb + Kc) * M) + 1
            assert all(MASK + S2[i] < 1 << 32 for i in range(NL - 1)) and _top(KCAP, M) + S2[NL - 1] < 1 << 32
            # the carry pass (limbs_normalize) adds at most (2^32 - 1) >> 30 = 3 to the next limb
            assert all(MASK + S2[i] + 3 < 1 << 32 for i in range(NL - 1))


def test_montgomery_product_columns_fit_64_bits():
    """limbs_mont_mul: column k accumulates sum a_i*b_(k-i) + sum m_i*N_(k-i) + carry in ONE 64-bit register.  Worst case over
    every magnitude pair the static_assert admits (Ka*Kb <= kmax, Ka, Kb <= KCAP)."""
    for name, d in _consts().items():
        M = _val(d['mod'])
        N = d['mod']
        ks = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
        worst = 0
        for Ka in ks:
            for Kb in ks:
                if Ka * Kb > d['kmax'] or Ka > KCAP or Kb > KCAP:
                    continue
                A = [MASK] * (NL - 1) + [_top(Ka, M)]
                B = [MASK] * (NL - 1) + [_top(Kb, M)]
                carry = 0
                for k in range(2 * NL - 1):
                    col = carry
                    for i in range(NL):
                        j = k - i
                        if 0 <= j < NL:
                            col += A[i] * B[j] + MASK * N[j]           # a_i*b_j and m_i*N_j with m_i <= 2^30 - 1
                    assert col < 1 << 64, (name, Ka, Kb, k, col.bit_length())
                    worst = max(worst, col)
                    carry = col >> W
                assert carry < 1 << 32                                 # the last limb is stored without a mask
                # value bound: (a*b + m*N) / R < 2N needs a*b < R*N, i.e. Ka*Kb*N < R
                assert Ka * Kb * M <= 1 << (W * NL)
        assert worst >= 1 << 63                                        # the bound is tight: no spare bit to spend
