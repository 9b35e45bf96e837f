// The byte layouts of a proof, shared by the device code (engine.h) and the host-only converters (api_json.hip).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define ZK_HD __host__ __device__
#else
#define ZK_HD
#endif
#define ZK_HDR 32
#define ZK_MAXSEC 128
// The two wire layouts of a proof (include/zkattest.h): ZKA1 pads every Tom-256 coordinate from the reference's 33 bytes to 36, ZKA1P does
// not.  Every run of Tom points in the layout has an even number of points (keyXcom keyYcom | Tx Ty | C8 C10 C11 C13 | the 6 points of a
// MultProof | A_1 A_2 | 4 n membership commitments), so point PAIRS take 132 bytes and every other field stays 4-byte aligned: 5.3 % fewer
// bytes across PCIe.  The prover's writers (k_scalar.hip) emit either; the verifier's kernels read ZKA1 only and are fed through k_v_unpack.
struct Wire {
    uint32_t tc;        // bytes per Tom coordinate: 36 | 33
    uint32_t fixed;     // header, R, comS1, keyXcom, keyYcom: 304 | 292
    uint32_t rep_head;  // A, Tx, Ty, 4 scalars: 336 | 324
    uint32_t mult;      // 6 points + 7 scalars: 656 | 620
    uint32_t eq;        // 2 points + 3 scalars: 240 | 228
    uint32_t padd;      // 4 points + 4 MultProofs + 2 EqualityProofs: 3392 | 3200
    uint32_t gk_n;      // per index bit: 4 points + 3 scalars: 384 | 360
    uint32_t magic;     // first four bytes as a little-endian word: "ZKA1" | "ZK1P"
};
#define ZK_MAGIC_ZKA1 0x31414b5au
#define ZK_MAGIC_ZKA1P 0x50314b5au
ZK_HD static inline Wire wire_make(bool packed) {
    Wire w;
    w.tc = packed ? 33 : 36;
    w.fixed = ZK_HDR + 2 * 64 + 4 * w.tc;
    w.rep_head = 64 + 4 * w.tc + 4 * 32;
    w.mult = 12 * w.tc + 7 * 32;
    w.eq = 4 * w.tc + 3 * 32;
    w.padd = 8 * w.tc + 4 * w.mult + 2 * w.eq;
    w.gk_n = 8 * w.tc + 3 * 32;
    w.magic = packed ? ZK_MAGIC_ZKA1P : ZK_MAGIC_ZKA1;
    return w;
}
ZK_HD static inline uint64_t wire_proof_size(const Wire& w, uint32_t sec, uint32_t n, uint32_t z) {
    return (uint64_t)w.fixed + (uint64_t)w.rep_head * sec + (uint64_t)w.padd * z + (uint64_t)w.gk_n * n + 32;
}
