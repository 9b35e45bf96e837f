// Function qualifiers of the per-lane device code.  The product compiles these headers with hipcc for gfx950 only.
// tests/host_arith defines ZK_HOST_BUILD and compiles the SAME headers (field.h, curve.h, sha256.h, rng.h) with g++ for the host
// CPU, so that the CPU test tier exercises this source against the oracle; the opaque-operand asm statements of field.h and the
// AMDGPU builtins are switched off / replaced by their portable definitions there.  One product unit is built that way too: h2c_host.cpp, the
// one-time host-side derivation of the hardened mode's generators (no kernel involved).
#pragma once
#include <stdint.h>
#ifdef ZK_HOST_BUILD
#define ZK_DEV inline
#define ZK_DEV_NOINLINE
#define ZK_CONSTANT static const
#define ZK_LAUNDER_MOD 0
#define ZK_PIN_LIMBS32 0
ZK_DEV uint32_t zk_rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
ZK_DEV uint32_t zk_funnelshift_r(uint32_t lo, uint32_t hi, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
ZK_DEV uint32_t zk_xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
ZK_DEV uint32_t zk_bfi(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }   // bitwise m ? a : b
ZK_DEV uint32_t zk_maj(uint32_t a, uint32_t b, uint32_t c) { return (a & b) | (c & (a | b)); }
#else
#include <hip/hip_runtime.h>
#define ZK_DEV __device__ __forceinline__
#define ZK_DEV_NOINLINE __device__ __noinline__
#define ZK_CONSTANT __constant__
ZK_DEV uint32_t zk_rotr32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
ZK_DEV uint32_t zk_funnelshift_r(uint32_t lo, uint32_t hi, uint32_t sh) { return __funnelshift_r(lo, hi, sh); }
// Three-input boolean functions in ONE instruction: gfx950 has no v_xor3_b32 (the assembler refuses it) but it has v_bitop3_b32 -- any function of three
// operands by its truth table, bit (a << 2 | b << 1 | c) of the immediate --, which the compiler does not form by itself: 0x96 = a ^ b ^ c (the four sigma
// functions of SHA-256: 2 xors -> 1), 0xe8 = majority (xor + bfi -> 1).  17 -> 14 boolean / rotate instructions per round.  v_bfi_b32 for Ch, likewise by hand.
ZK_DEV uint32_t zk_xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
ZK_DEV uint32_t zk_maj(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xe8); }
ZK_DEV uint32_t zk_bfi(uint32_t m, uint32_t a, uint32_t b) {   // bitwise m ? a : b
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
#endif
