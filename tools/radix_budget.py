#!/usr/bin/env python3
"""Why k_tom_commit's products stay at radix 2^30 with normalised operands ("instruction diet").

Two budgets decide whether a cheaper representation exists for the 258-bit Tom-256 field prime t:

  (1) the COLUMN budget of a product-scanning Montgomery product with 64-bit column sums and no carry flags: column k holds up
      to 9 partial products a_i b_(k-i) and up to 9 products m_i t_(k-i); their sum must stay below 2^64.  It decides whether an
      operand may enter a product UNNORMALISED (limbs of a sum of two elements, one bit wider), which would remove the carry sweep
      of every addition / subtraction in the curve formulas (~25 of the 220 VALU instructions per product).
  (2) the MAGNITUDE budget kappa = R / t of lazy reduction: a product of inputs < Ka t and < Kb t comes out < (Ka Kb / kappa + 1) t,
      and the unified addition law multiplies sums and differences of earlier outputs (E ~ P4 - A - B, F ~ Z - C ...).

Radix 2^30 (R = 2^270) has kappa = 4096 but no column headroom; radix 2^29 (R = 2^261) has one bit of column headroom per operand
but kappa = 8.0000000005, because t = 2^258 - 2^226 + ... sits just below 2^258.  The tables below are computed from the constants,
nothing is assumed."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'ref_facts.json')))
t = int(F['groups']['tomEdwards256']['constants'][0], 16)


def limbs(x, bits, n):
    return [(x >> (bits * i)) & ((1 << bits) - 1) for i in range(n)]


def column_table(bits, a_max, b_max, note):
    """worst column sum of a product with operand limbs <= a_max / b_max (top limbs bounded by the value bound 512 t)"""
    n = 9
    M = limbs(t, bits, n)
    top = (512 * t) >> (bits * (n - 1))                  # top limb of any value < 512 t (KCAP of field.h)
    amax = [a_max] * (n - 1) + [min(a_max, top)]
    bmax = [b_max] * (n - 1) + [min(b_max, top)]
    mmax = (1 << bits) - 1
    worst, where = 0, 0
    carry = 0
    for k in range(2 * n - 1):
        s = carry
        for i in range(n):
            j = k - i
            if 0 <= j < n:
                s += amax[i] * bmax[j]
                if i < min(k + 1, n):                    # m_i exists for i <= k (and i < n)
                    s += mmax * M[j]
        if s > worst:
            worst, where = s, k
        carry = s >> bits
    ok = worst < (1 << 64)
    print('  radix 2^%d, operand limbs < 2^%.2f x 2^%.2f %-28s worst column %2d: 2^%.3f  -> %s'
          % (bits, (a_max + 1).bit_length() - 1 + 0.0 if a_max + 1 == 1 << ((a_max + 1).bit_length() - 1) else __import__('math').log2(a_max + 1),
             __import__('math').log2(b_max + 1), '(' + note + ')', where, __import__('math').log2(worst), 'fits 64 bits' if ok else 'OVERFLOWS'))
    return ok


def magnitude_fixed_point(kappa):
    """bounds (in units of t) of the coordinates of the running point under repeated unified additions with canonical table entries,
    when sums / differences enter the next products unreduced: iterate X3 = E F, Y3 = G H, T3 = E H, Z3 = F G"""
    x = y = tt = z = 2.0
    for it in range(60):
        A, B, C, P4 = x * 1 / kappa + 1, y * 1 / kappa + 1, tt * 1 / kappa + 1, (x + y) * 2 / kappa + 1
        sub = lambda v: 2 ** max(1, (int(v) + 1).bit_length())     # the power-of-two multiple of t added before subtracting v t
        E, Fv, G, H = P4 + sub(A + B), z + sub(C), z + C, B + sub(A)
        x, y, tt, z = E * Fv / kappa + 1, G * H / kappa + 1, E * H / kappa + 1, Fv * G / kappa + 1
        if max(x, y, tt, z) > 1e6:
            return None
    return max(x, y, tt, z)


print('t = 2^258 - 2^%.1f   (bit length %d)' % (__import__('math').log2((1 << 258) - t), t.bit_length()))
print('(1) column budget (64-bit column sums, no carry flags)')
n30, n29 = (1 << 30) - 1, (1 << 29) - 1
column_table(30, n30, n30, 'both normalised: what ships')
column_table(30, 2 * n30, n30, 'one operand a raw sum')
column_table(30, 2 * n30, 2 * n30, 'both operands raw sums')
column_table(29, n29, n29, 'both normalised')
column_table(29, 2 * n29, 2 * n29, 'both operands raw sums')
column_table(29, 3 * n29, 2 * n29, 'raw a + C t - b  x  raw sum')
print('(2) magnitude budget kappa = R / t')
for bits in (30, 29):
    R = 1 << (bits * 9)
    kappa = R / t
    fp = magnitude_fixed_point(kappa)
    print('  radix 2^%d: R = 2^%d, kappa = %.10f: coordinates of the running point %s'
          % (bits, bits * 9, kappa, 'stay below %.2f t (closed)' % fp if fp else 'GROW WITHOUT BOUND (no fixed point): every sum would need an extra reduction'))
print('=> radix 2^30 has the magnitudes but not the columns for unreduced operands; radix 2^29 has the columns but not the magnitudes;')
print('   10 limbs of 29 bits would have both at 200 instead of 162 multiplier instructions per product.')
