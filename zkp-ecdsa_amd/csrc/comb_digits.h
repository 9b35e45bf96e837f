// Comb digit extraction for the Tom-256 fixed-base tables (k_tom.hip): pure integer code, also compiled for the host by
// tests/host_arith (ZK_HOST_BUILD) where the recoding is checked to reproduce the scalar for every supported width.
#pragma once
#include "zkdev.h"

// widths above 24 bits use signed digits (half the entries per window); engine.h's tom_signed() is the same predicate
ZK_DEV bool comb_signed(uint32_t bits) { return bits > 24; }
// 256-bit right shift by a run-time amount < 32 (v_alignbit_b32 per word)
ZK_DEV void shr256_rt(uint32_t* w, uint32_t sh) {
#pragma unroll
    for (int i = 0; i < 7; i++) w[i] = zk_funnelshift_r(w[i], w[i + 1], sh);
    w[7] >>= sh;
}
// successive comb digits of a 256-bit scalar, lowest window first: table index and sign (always + for unsigned widths)
struct CombDigits {
    uint32_t w[8];
    uint32_t bits, mask, half, carry;
    bool sgn;
    ZK_DEV void init(uint32_t b) { bits = b, mask = (1u << b) - 1, half = 1u << (b - 1), carry = 0, sgn = comb_signed(b); }
    ZK_DEV void next(uint32_t& idx, bool& neg) {
        uint32_t d = (w[0] & mask) + carry;
        shr256_rt(w, bits);
        neg = sgn && d > half;
        carry = neg ? 1u : 0u;
        idx = neg ? (mask + 1) - d : d;
#ifdef ZK_DEBUG_IDX_MASK  // timing experiments only (wrong results): confine the gathers to the first entries of each window
        idx &= ZK_DEBUG_IDX_MASK;
#endif
    }
};

// Signed 8-bit digits of a 256-bit scalar for the per-key tables (k_ktab.hip, ktab.h): 33 windows, digit in [-127, 128], the table
// holds the multiples 1..128 of every window base (slot d - 1); the 33rd window only sees the carry out of the 32nd.
struct KeyDigits {
    uint32_t w[8];
    uint32_t carry;
    ZK_DEV void init() { carry = 0; }
    ZK_DEV void next(uint32_t& d, bool& neg) {   // d = 0: nothing to add
        d = (w[0] & 255u) + carry;
        shr256_rt(w, 8);
        neg = d > 128u;
        carry = neg ? 1u : 0u;
        if (neg) d = 256u - d;
    }
};
