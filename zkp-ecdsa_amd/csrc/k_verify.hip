// verifySignatureList on the GPU (src/zkpAttestList.ts:147-184): verifyMembership (src/proofGK/gk.ts:197-262) and
// verifyExp with secparam 20 (src/exp/exp.ts:233-349) including aggregatePointAdd / aggregateMult /
// aggregateEquality (src/exp/pointAdd.ts:199-259, src/commit/mult.ts:148-175, src/commit/equality.ts:94-116).
//
// The reference batches every verification equation into "sum of scalar*point = identity" checks with fresh random
// scalars (Relation.drain, src/curves/multimult.ts:168-173) and evaluates them with Bos-Coster (multimult.ts:61-89).
// Only the boolean is observable, so the engine uses its own 128-bit randomisers and evaluates the same three sums
// (membership, Exp over Tom-256, Exp over P-256) as interleaved (Straus) double-and-add, one group of <= 36 terms per
// lane with shared doublings; fixed-base parts (g, h, G, h_NIST, R) go through the comb tables.  Which 20 of the sec
// reps are checked follows the reference's generateIndices (exp.ts:95-109) under the verifier-RNG contract
// (fill k = SHA-256(vseed || be64(k)); randomScalar takes 32 bytes, rnd(small) takes 1 byte), so a proof with
// SOME bad reps gets the same verdict as the reference/oracle for the same verifier seed.
#include "rtab.h"

typedef Fe<ModQ, 1> Sq;
typedef Fe<ModN, 1> Sn;
typedef Fe<ModT, 1> St;

// ------------------------------------------------------------------ byte parsing
template <int NW>
ZK_DEV void ld_words_be(const uint8_t* p, uint32_t* w) {
    const uint32_t* q = (const uint32_t*)p;
#pragma unroll
    for (int i = 0; i < NW; i++) w[i] = bswap32(q[NW - 1 - i]);
}
ZK_DEV Sq ld_scalar_q(const uint8_t* p) {
    uint32_t w[8];
    load_be32(p, w);
    return fe_from_words256_reduce<ModQ>(w);  // Scalar ctor reduces (group.ts:164-167)
}
ZK_DEV Sn ld_scalar_n(const uint8_t* p) {
    uint32_t w[8];
    load_be32(p, w);
    return fe_from_words256_reduce<ModN>(w);
}
// 72-byte Tom point -> plain limbs; false if a coordinate is >= t (edwards.ts:74-77)
ZK_DEV bool ld_tom_bytes(const uint8_t* p, St& x, St& y) {
    uint32_t xw[9], yw[9];
    ld_words_be<9>(p, xw);
    ld_words_be<9>(p + 36, yw);
    bool ok = !words_geq<9>(xw, ModT::mod32) && !words_geq<9>(yw, ModT::mod32);
    limbs_from_words<9>(x.l, xw);
    limbs_from_words<9>(y.l, yw);
    return ok;
}
ZK_DEV bool tom_bytes_valid(const uint8_t* p) {  // edwards.ts:204-209 afterJson: range + curve equation
    uint32_t xw[9], yw[9];
    ld_words_be<9>(p, xw);
    ld_words_be<9>(p + 36, yw);
    return tom_words_on_curve(xw, yw);
}
ZK_DEV bool p256_bytes_valid(const uint8_t* p) {  // weier.ts:256-260
    uint32_t xw[8], yw[8];
    load_be32(p, xw);
    load_be32(p + 32, yw);
    P256Aff a;
    a.x = fe_to_mont(fe_from_words256_reduce<ModQ>(xw));
    a.y = fe_to_mont(fe_from_words256_reduce<ModQ>(yw));
    return p256_on_curve(a);
}
ZK_DEV uint64_t v_proof_size(uint32_t sec, uint32_t n, uint32_t z) {
    return (uint64_t)ZK_FIXED + (uint64_t)ZK_REP_HEAD * sec + (uint64_t)ZK_PADD_SZ * z + (uint64_t)n * (4 * 72 + 3 * 32) + 32;
}
ZK_DEV const uint8_t* v_gk_base(const VWork& V, const uint8_t* pr, uint32_t p) {
    return pr + ZK_FIXED + (uint64_t)ZK_REP_HEAD * V.sec + (uint64_t)ZK_PADD_SZ * V.zcnt[p];
}

// ------------------------------------------------------------------ header + structural validation
__global__ void __launch_bounds__(256) k_v_header(VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    uint32_t p = gtid();
    if (p >= count) return;
    uint64_t o0 = off[first + p], o1 = off[first + p + 1];
    const uint8_t* pr = proofs + o0;
    int32_t st = ZK_OK;
    uint32_t bits[4] = {0, 0, 0, 0};
    uint32_t z = 0, flags = 0;
    if (o1 < o0 + ZK_HDR || (o0 & 3)) st = ZK_E_BAD_ENCODING;
    else {
        const uint32_t* h = (const uint32_t*)pr;
        uint32_t total = bswap32(h[1]), sec = bswap32(h[2]), n = bswap32(h[3]);
        bits[0] = bswap32(h[7]), bits[1] = bswap32(h[6]), bits[2] = bswap32(h[5]), bits[3] = bswap32(h[4]);
        // ZKA1 is this engine's format: a batch is homogeneous in secLevel (the reference would verify a proof of another length with
        // that proof's own repetition count, exp.ts:243-260; include/zkattest.h documents the restriction)
        if (h[0] != 0x31414b5au || total != o1 - o0 || sec != V.sec || n > 63) st = ZK_E_BAD_ENCODING;
        else {
            for (uint32_t b = V.sec; b < 128; b++)
                if ((bits[b >> 5] >> (b & 31)) & 1) st = ZK_E_BAD_ENCODING;   // unused challenge bits are zero
            z = zeros_below(bits, V.sec);
            if (total != v_proof_size(V.sec, n, z)) st = ZK_E_BAD_ENCODING;   // the structure the header announces, with ITS n
            // a GKProof of the wrong length is "return false" in the reference (gk.ts:208-218), not an exception -- but the proof was
            // deserialised first, so every one of its points, the 4 n' membership commitments included, must still be a point
            else if (n != V.n) flags = 8 | (n << 16);
        }
    }
    V.okflags[p] = flags;
    V.st[p] = st;
    V.zcnt[p] = z;
#pragma unroll
    for (int i = 0; i < 4; i++) V.hbits[4 * p + i] = bits[i];
}
// every point of the proof must deserialise (on curve, coordinates in range).  One thread per POINT that exists: the P-256 points first (R, comS1,
// A_i), then the Tom-256 ones -- keyXcom, keyYcom, the 4n membership commitments, Tx_i and Ty_i of every repetition, and the 32 PointAdd points of
// every ZERO-bit repetition, in the order of their rank among the zero bits -- so consecutive lanes check consecutive strings of one kind, every lane of
// a wave has work (round 4 gave every repetition 34 slots and let the 32 PointAdd lanes of a one-bit repetition exit: 45 % of the Tom lanes idle),
// and the branchy part only selects an ADDRESS (one copy of each curve check per wave).  The Tom check runs on plain coordinates
// (curve.h: tom_words_on_curve, 5 products instead of 7).  A proof whose GKProof has another length n' than the ring's (okflags bit 3; never produced
// by an honest prover) keeps its threads for everything else and has its 4 n' membership commitments walked by the thread of its keyXcom.
ZK_DEV uint32_t kth_set_bit(uint32_t x, uint32_t k) {   // position of the k-th (0-based) set bit of x; x has more than k bits set
    uint32_t pos = 0;
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) {
        const uint32_t lo = x & ((1u << sh) - 1), c = __popc(lo);
        if (k >= c) k -= c, x >>= sh, pos += sh;
        else x = lo;
    }
    return pos;
}
// the repetition whose challenge bit is the zr-th (0-based) ZERO among bits [0, sec) (there is one: zr < number of zero bits)
ZK_DEV uint32_t zero_bit_rep(const uint32_t* hb, uint32_t sec, uint32_t zr) {
    uint32_t j = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t lo = 32 * w;
        if (sec <= lo) break;
        const uint32_t nb = sec - lo >= 32 ? 32 : sec - lo;
        const uint32_t z = ~hb[w] & (nb == 32 ? 0xffffffffu : ((1u << nb) - 1));
        const uint32_t c = __popc(z);
        if (zr < c) {
            j = lo + kth_set_bit(z, zr);
            break;
        }
        zr -= c;
    }
    return j;
}
ZK_DEV uint32_t v_validate_threads(uint32_t sec, uint32_t n) { return (2 + sec) + 2 + 4 * n + 2 * sec + 32 * sec; }   // per proof, all bits zero
__global__ void __launch_bounds__(256) k_v_validate(VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    const uint32_t np = 2 + V.sec, ngk = 2 + 4 * V.n, per_max = v_validate_threads(V.sec, V.n);
    uint32_t blocks_per_proof = (per_max + 255) / 256;           // a workgroup never straddles two proofs
    uint32_t p = blockIdx.x / blocks_per_proof, m = (blockIdx.x % blocks_per_proof) * 256 + threadIdx.x;
    if (p >= count) return;
    if (m >= np + ngk + 2 * V.sec + 32 * V.zcnt[p]) return;     // (whole trailing waves: the PointAdd points of this proof's zero bits end here)
    if (V.st[p] != ZK_OK) return;
    const uint32_t other_n = V.okflags[p] & 8 ? V.okflags[p] >> 16 : 0xffffffffu;
    const uint8_t* pr = proofs + off[first + p];
    const uint32_t* hb = V.hbits + 4 * p;
    const uint8_t* ptr;
    const bool is_p = m < np;
    bool ok = true;
    if (is_p) {
        ptr = m < 2 ? pr + 32 + 64 * m : pr + rep_offset(hb, m - 2);
    } else {
        uint32_t u = m - np;
        if (u < 2) {
            ptr = pr + 160 + 72 * u;
            if (u == 0 && other_n != 0xffffffffu) {
                const uint8_t* gk = v_gk_base(V, pr, p);
                for (uint32_t k = 0; k < 4 * other_n; k++) ok = ok && tom_bytes_valid(gk + 72 * k);
            }
        } else if (u < ngk) {
            if (other_n != 0xffffffffu) return;
            ptr = v_gk_base(V, pr, p) + 72 * (u - 2);
        } else if (u < ngk + 2 * V.sec) {   // Tx_j, Ty_j
            const uint32_t r = u - ngk;
            ptr = pr + rep_offset(hb, r >> 1) + 64 + 72 * (r & 1);
        } else {   // point q of the PointAdd proof of the zr-th zero-bit repetition: zeros_below(j) = zr, so its offset needs no bit counting
            const uint32_t r = u - ngk - 2 * V.sec, zr = r >> 5, q = r & 31;
            const uint32_t j = zero_bit_rep(hb, V.sec, zr);
            const uint8_t* pa = pr + ZK_FIXED + (uint64_t)ZK_REP_HEAD * (j + 1) + (uint64_t)ZK_PADD_SZ * zr;
            // q: 0..3 C8..C13, 4..27 the six points of the four MultProofs, 28..31 A_1, A_2 of pi_x, pi_y
            if (q < 4) ptr = pa + 72 * q;
            else if (q < 28) ptr = pa + 288 + 656 * ((q - 4) / 6) + 72 * ((q - 4) % 6);
            else ptr = pa + (q < 30 ? 2912 : 3152) + 72 * (q & 1);
        }
    }
    ok = ok && (is_p ? p256_bytes_valid(ptr) : tom_bytes_valid(ptr));
    if (!ok) atomicCAS(&V.st[p], ZK_OK, ZK_E_BAD_ENCODING);
}

// ------------------------------------------------------------------ ZKA1P -> ZKA1 (include/zkattest.h: the packed wire layout)
// The verifier's kernels read 36-byte Tom coordinates at 4-byte aligned offsets; packed proofs (33-byte coordinates) are expanded once per
// chunk into a staging buffer and everything else runs unchanged on the expanded bytes.  k_v_unpack_scan: one workgroup per chunk, the
// expanded length of every proof from its header -- a proof whose header is not a well-formed ZKA1P header of this context's secLevel
// expands to its 32 header bytes alone, which k_v_header then refuses (ZK_E_BAD_ENCODING) -- and their prefix sums from `base`.
// k_v_unpack: one workgroup per (proof, part): the repetitions, the fixed part, the membership proof.
ZK_DEV bool v_packed_header_ok(const uint8_t* pr, uint64_t plen, uint32_t want_sec, uint32_t& sec, uint32_t& n, uint32_t& z, uint32_t bits[4]) {
    if (plen < ZK_HDR) return false;
    const uint32_t* h = (const uint32_t*)pr;
    const uint32_t total = bswap32(h[1]);
    sec = bswap32(h[2]), n = bswap32(h[3]);
    bits[0] = bswap32(h[7]), bits[1] = bswap32(h[6]), bits[2] = bswap32(h[5]), bits[3] = bswap32(h[4]);
    if (h[0] != ZK_MAGIC_ZKA1P || total != plen || sec != want_sec || n > 63) return false;
    for (uint32_t b = sec; b < 128; b++)
        if ((bits[b >> 5] >> (b & 31)) & 1) return false;
    z = zeros_below(bits, sec);
    return plen == wire_proof_size(wire_make(true), sec, n, z);
}
__global__ void __launch_bounds__(256) k_v_unpack_scan(uint32_t want_sec, uint32_t count, const uint8_t* packed, const uint64_t* poff, uint64_t first, uint64_t base,
                                                       uint64_t* uoff /* [count + 1], entry p = proof first + p */) {
    __shared__ uint64_t sb[256];
    const uint32_t t = threadIdx.x, per = (count + 255) / 256;
    const uint32_t lo = t * per, hi = lo + per < count ? lo + per : count;
    uint64_t sum = 0;
    for (uint32_t p = lo; p < hi; p++) {
        const uint64_t o0 = poff[first + p], o1 = poff[first + p + 1];
        uint32_t sec, n, z, bits[4];
        const bool ok = o1 >= o0 && !(o0 & 3) && v_packed_header_ok(packed + o0, o1 - o0, want_sec, sec, n, z, bits);
        sum += ok ? wire_proof_size(wire_make(false), sec, n, z) : ZK_HDR;
    }
    sb[t] = sum;
    __syncthreads();
    if (t == 0) {
        uint64_t run = base;
        for (int i = 0; i < 256; i++) {
            const uint64_t v = sb[i];
            sb[i] = run, run += v;
        }
    }
    __syncthreads();
    uint64_t run = sb[t];
    for (uint32_t p = lo; p < hi; p++) {
        const uint64_t o0 = poff[first + p], o1 = poff[first + p + 1];
        uint32_t sec, n, z, bits[4];
        const bool ok = o1 >= o0 && !(o0 & 3) && v_packed_header_ok(packed + o0, o1 - o0, want_sec, sec, n, z, bits);
        uoff[p] = run;
        run += ok ? wire_proof_size(wire_make(false), sec, n, z) : ZK_HDR;
        if (p + 1 == count) uoff[count] = run;
    }
}
// Output dword u of a part -> where it comes from in the packed part: a plain dword (copied) or word j of a Tom coordinate (33 source bytes ->
// 9 dwords, the first one holding the top byte behind three zero bytes).  Pure arithmetic on the fixed shapes of the parts: no tables, no scratch.
struct UnpackSrc {
    uint32_t off;   // byte offset inside the packed part: of the dword, or of the coordinate's first byte
    uint32_t j;     // 0xffffffff: plain dword; else word 0..8 of the coordinate
};
ZK_DEV UnpackSrc unpack_tom(uint32_t base, uint32_t w) { return UnpackSrc{base + 33 * (w / 9), w % 9}; }
ZK_DEV UnpackSrc unpack_plain(uint32_t base, uint32_t w) { return UnpackSrc{base + 4 * w, 0xffffffffu}; }
ZK_DEV UnpackSrc unpack_map_rep(uint32_t u) {   // A | Tx Ty | 4 scalars | C8 C10 C11 C13 | 4 x (6 points, 7 scalars) | 2 x (2 points, 3 scalars)
    if (u < 16) return unpack_plain(0, u);
    if (u < 52) return unpack_tom(64, u - 16);
    if (u < 84) return unpack_plain(196, u - 52);
    uint32_t v = u - 84;
    const uint32_t pa = 324;
    if (v < 72) return unpack_tom(pa, v);
    v -= 72;
    if (v < 4 * 164) {
        const uint32_t m = v / 164, w = v % 164, b = pa + 264 + 620 * m;
        return w < 108 ? unpack_tom(b, w) : unpack_plain(b + 396, w - 108);
    }
    v -= 4 * 164;
    const uint32_t e = v / 60, w = v % 60, b = pa + 264 + 4 * 620 + 228 * e;
    return w < 36 ? unpack_tom(b, w) : unpack_plain(b + 132, w - 36);
}
ZK_DEV uint32_t unpack_fetch(const uint8_t* src, const UnpackSrc& m) {
    const uint8_t* q = src + m.off;
    if (m.j == 0) return (uint32_t)q[0] << 24;
    if (m.j != 0xffffffffu) q += 1 + 4 * (m.j - 1);
    return (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24;
}
__global__ void __launch_bounds__(256) k_v_unpack(uint32_t want_sec, uint32_t count, const uint8_t* packed, const uint64_t* poff, uint64_t first, uint8_t* out,
                                                  const uint64_t* uoff /* entry p = proof first + p */) {
    const uint32_t parts = want_sec + 2, p = blockIdx.x / parts, part = blockIdx.x % parts;
    if (p >= count) return;
    const uint64_t o0 = poff[first + p], o1 = poff[first + p + 1];
    const uint8_t* src = packed + o0;
    uint32_t* dst = (uint32_t*)(out + uoff[p]);
    uint32_t sec, n, z, bits[4];
    const bool ok = o1 >= o0 && !(o0 & 3) && v_packed_header_ok(src, o1 - o0, want_sec, sec, n, z, bits);
    if (!ok) {   // the header alone, so that k_v_header sees a proof it must refuse
        if (part == 0 && threadIdx.x < 8) {
            uint32_t v = o1 >= o0 + ZK_HDR && !(o0 & 3) ? ((const uint32_t*)src)[threadIdx.x] : 0;
            if (threadIdx.x == 0) v = ZK_MAGIC_ZKA1;
            if (threadIdx.x == 1 && bswap32(v) == ZK_HDR) v = 0;   // never a consistent 32-byte "proof"
            dst[threadIdx.x] = v;
        }
        return;
    }
    const Wire pw = wire_make(true), uw = wire_make(false);
    uint32_t ndw;
    uint64_t so = 0, dof = 0;
    if (part < sec) {   // repetition `part`: A, Tx, Ty, four scalars, and for a zero bit the PointAdd proof
        so = rep_offset_w(pw, bits, part), dof = rep_offset_w(uw, bits, part);
        ndw = ((bits[part >> 5] >> (part & 31)) & 1) ? uw.rep_head / 4 : (uw.rep_head + uw.padd) / 4;
    } else if (part == sec) {   // header, R, comS1, keyXcom, keyYcom
        ndw = uw.fixed / 4;
    } else {   // membership proof: 4 n points, 3 n + 1 scalars
        so = pw.fixed + (uint64_t)pw.rep_head * sec + (uint64_t)pw.padd * z, dof = uw.fixed + (uint64_t)uw.rep_head * sec + (uint64_t)uw.padd * z;
        ndw = (uw.gk_n * n + 32) / 4;
    }
    for (uint32_t u = threadIdx.x; u < ndw; u += 256) {
        UnpackSrc m;
        if (part < sec) m = unpack_map_rep(u);
        else if (part == sec) m = u < 40 ? unpack_plain(0, u) : unpack_tom(160, u - 40);
        else m = u < 72 * n ? unpack_tom(0, u) : unpack_plain(264 * n, u - 72 * n);
        uint32_t v = unpack_fetch(src + so, m);
        if (part == sec && u == 0) v = ZK_MAGIC_ZKA1;
        if (part == sec && u == 1) v = bswap32((uint32_t)wire_proof_size(uw, sec, n, z));
        dst[dof / 4 + u] = v;
    }
}
// offsets handed over as a DEVICE array (zk_verify_batch_device with packed proofs): non-decreasing and 4-byte aligned?  The expansion buffer is sized from
// off[B] and every proof's expanded size from its own off[i + 1] - off[i], so overlapping ranges ([0, L, 0, L, ...]) would add up past that buffer.
__global__ void __launch_bounds__(256) k_offsets_monotonic(const uint64_t* __restrict__ off, uint64_t B, uint32_t* bad) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < B && (off[i] > off[i + 1] || (off[i] & 3))) atomicOr(bad, 1u);
}
void launch_offsets_monotonic(hipStream_t s, const uint64_t* d_off, uint64_t B, uint32_t* d_bad) {
    if (B) hipLaunchKernelGGL(k_offsets_monotonic, dim3((uint32_t)((B + 255) / 256)), dim3(256), 0, s, d_off, B, d_bad);
}
void launch_v_unpack(hipStream_t s, uint32_t sec, uint32_t count, const uint8_t* packed, const uint64_t* poff, uint64_t first, uint64_t base, uint8_t* out, uint64_t* uoff) {
    if (!count) return;
    hipLaunchKernelGGL(k_v_unpack_scan, dim3(1), dim3(256), 0, s, sec, count, packed, poff, first, base, uoff);
    hipLaunchKernelGGL(k_v_unpack, dim3(count * (sec + 2)), dim3(256), 0, s, sec, count, packed, poff, first, out, uoff);
}

// ------------------------------------------------------------------ front: R, Q (zkpAttestList.ts:153-164), kx, ky
// R from the proof (what the window table of R needs), then Q = (z / R.x) G: two kernels, so that a small chunk can build R's table (256 doublings in a
// row) while the second one inverts and walks the comb.
__global__ void __launch_bounds__(64) k_v_front_r(Workspace W, VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    uint32_t p = gtid();
    if (p >= count) return;
    W.st[p] = V.st[p];
    if (V.st[p] != ZK_OK || (V.okflags[p] & 8)) {   // keep later kernels on defined data: R = G
        soa_st(W.Rxm, p, fe_const<ModQ, 2>(P256_GX_M)), soa_st(W.Rym, p, fe_const<ModQ, 2>(P256_GY_M));
        return;
    }
    const uint8_t* pr = proofs + off[first + p];
    uint32_t xw[8], yw[8];
    load_be32(pr + 32, xw);
    load_be32(pr + 64, yw);
    soa_st(W.Rxm, p, fe_to_mont(fe_from_words256_reduce<ModQ>(xw))), soa_st(W.Rym, p, fe_to_mont(fe_from_words256_reduce<ModQ>(yw)));
}
__global__ void __launch_bounds__(64) k_v_front_q(DevParams P, Workspace W, VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, const uint8_t* msg, uint64_t first) {
    uint32_t p = gtid();
    if (p >= count) return;
    if (V.st[p] != ZK_OK || (V.okflags[p] & 8)) {
        P256Pt id = p256_identity();
        soa_st(W.Q.x, p, id.x), soa_st(W.Q.y, p, id.y), soa_st(W.Q.z, p, id.z);
        return;
    }
    const uint8_t* pr = proofs + off[first + p];
    uint32_t xw[8], zw[8];
    load_be32(pr + 32, xw);
    load_be32(msg + 32 * (first + p), zw);
    Sq rx = fe_from_words256_reduce<ModQ>(xw);
    // rinv = invMod(R.x, n); z1 = rinv * z; Q = G * z1
    uint32_t rxw[8];
    words_from_limbs<8>(rxw, rx.l);
    Fn2 rxn = fe_to_mont(fe_from_words256_reduce<ModN>(rxw));
    Fn2 z = fe_to_mont(fe_from_words256_reduce<ModN>(zw));
    Sn z1 = fe_from_mont(fe_inv<ModN>(rxn) * z);
    uint32_t kw[8];
    words_from_limbs<8>(kw, z1.l);
    P256Pt acc = p256_identity();
#pragma unroll 1
    for (int w = 0; w < PFIX_NWIN; w++) {
        uint32_t d = kw[0] & (PFIX_WIN_SIZE - 1);
        shr256<PFIX_WIN_BITS>(kw);
        const uint32_t* e = P.pfix_G + (size_t)PFIX_ENTRY_WORDS * (w * PFIX_WIN_SIZE + d);
        P256Aff a;
        for (int l = 0; l < 9; l++) a.x.l[l] = e[l], a.y.l[l] = e[9 + l];
        P256Pt s = p256_add_mixed(acc, a);
        acc = p256_select(d != 0, s, acc);
    }
    soa_st(W.Q.x, p, acc.x), soa_st(W.Q.y, p, acc.y), soa_st(W.Q.z, p, acc.z);
}

// ------------------------------------------------------------------ hashing from proof bytes
ZK_DEV void absorb_tom_bytes(ShaStream& s, const uint8_t* p72) {  // 36-byte padded coordinates -> 33-byte encodings
    uint32_t w[9];
    s.put_byte(4);
    ld_words_be<9>(p72, w);
    s.put_be<33>(w);
    ld_words_be<9>(p72 + 36, w);
    s.put_be<33>(w);
}
// hashPoints serialises through toBytes -> toAffine (group.ts:221-233, weier.ts:231-255), which reduces the coordinates mod p;
// deserializePoint / isOnGroup accept x or y in [p, 2^256) (weier.ts:74-89 works mod p), so a stored coordinate may be
// non-canonical: the REDUCED value is what the reference hashes.
ZK_DEV void absorb_p256_bytes(ShaStream& s, const uint8_t* p64) {
    uint32_t w[8];
    s.put_byte(4);
    load_be32(p64, w);
    words_from_limbs<8>(w, fe_from_words256_reduce<ModQ>(w).l);
    s.put_be<32>(w);
    load_be32(p64 + 32, w);
    words_from_limbs<8>(w, fe_from_words256_reduce<ModQ>(w).l);
    s.put_be<32>(w);
}
ZK_DEV void absorb_tom_soa(ShaStream& s, const Soa& ax, const Soa& ay, uint32_t e) {
    uint32_t w[9];
    s.put_byte(4);
    words_from_limbs<9>(w, soa_ld<ModT, 1>(ax, e).l);
    s.put_be<33>(w);
    words_from_limbs<9>(w, soa_ld<ModT, 1>(ay, e).l);
    s.put_be<33>(w);
}
ZK_DEV void v_challenge_words(const uint32_t h[8], uint32_t c[4]) {
    c[0] = (h[1] << 16) | (h[2] >> 16), c[1] = (h[0] << 16) | (h[1] >> 16), c[2] = h[0] >> 16, c[3] = 0;
}
// Exp challenge over ALL reps (exp.ts:253-260) and the GK challenge x (gk.ts:220-221)
__global__ void __launch_bounds__(64) k_v_challenges(VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, const uint8_t* __restrict__ msg, uint64_t first, uint32_t parts) {
    __shared__ uint32_t lds[16 * 64];
    uint32_t p = gtid();
    bool live = p < count;
    if (!live) p = count - 1;
    const uint8_t* pr = proofs + off[first + p];
    bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8);
    ShaStream s;
    uint32_t h[8], c[4];
    if (parts & 1) {
        s.init(lds, threadIdx.x, 64);
        if (good) {
            const uint32_t* hb = V.hbits + 4 * p;
            absorb_tom_bytes(s, pr + 160);
            absorb_tom_bytes(s, pr + 232);
            for (uint32_t i = 0; i < V.sec; i++) {
                const uint8_t* rep = pr + rep_offset(hb, i);
                absorb_p256_bytes(s, rep);
                absorb_tom_bytes(s, rep + 64);
                absorb_tom_bytes(s, rep + 136);
            }
        }
        s.finish(h);
        v_challenge_words(h, c);
        if (live)
            for (int i = 0; i < 4; i++) V.chal[4 * p + i] = c[i];
    }
    if (!(parts & 2)) return;
    s.init(lds, threadIdx.x, 64);
    if (good) {
        const uint8_t* gk = v_gk_base(V, pr, p);
        for (uint32_t k = 0; k < 4 * V.n; k++) absorb_tom_bytes(s, gk + 72 * k);
        if (V.hardened) {   // the statement binds the challenge (k_hash.hip: k_gk_hash)
            sha_put_gk_statement_head(s, V.ring_digest, msg + 32 * (first + p));
            absorb_p256_bytes(s, pr + 32);
            absorb_tom_bytes(s, pr + 160);
        }
    }
    s.finish(h);
    v_challenge_words(h, c);
    if (live) V.gkx[3 * p] = c[0], V.gkx[3 * p + 1] = c[1], V.gkx[3 * p + 2] = c[2];
}

// The Exp challenge of a small chunk through the three-kernel path (k_hash.hip: k_exph_sched, k_exph_rounds): one lane per point writes the padded message
// straight from the proof bytes -- a Tom coordinate is bytes 3..35 of its 36-byte image, a P-256 coordinate is hashed REDUCED (see absorb_p256_bytes) --;
// a proof that did not parse gets a message of zeros (its challenge is never used, its bytes may not be there to read).
__global__ void __launch_bounds__(256) k_v_exph_msg(Workspace W, VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    const uint32_t ne = 2 + 3 * V.sec + 1, t = gtid();
    if (t >= count * ne) return;
    const uint32_t p = t / ne, e = t % ne;
    uint8_t* m = W.exph_msg + (size_t)p * exph_blocks(V.sec) * 64;
    if (e == ne - 1) {
        exph_put_padding(m, V.sec);
        return;
    }
    const bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8);
    const uint32_t o = exph_elem_offset(e);
    const bool p256 = e >= 2 && (e - 2) % 3 == 0;
    const uint32_t nb = p256 ? 65 : 67;
    if (!good) {
        for (uint32_t i = 0; i < nb; i++) m[o + i] = 0;
        return;
    }
    const uint8_t* pr = proofs + off[first + p];
    const uint8_t* src;
    if (e < 2) src = pr + 160 + 72 * e;
    else {
        const uint32_t j = (e - 2) / 3, k = (e - 2) % 3;
        src = pr + rep_offset(V.hbits + 4 * p, j) + (k == 0 ? 0 : k == 1 ? 64 : 136);
    }
    m[o] = 4;
    if (p256) {
#pragma unroll 1
        for (int c2 = 0; c2 < 2; c2++) {
            uint32_t w[8];
            load_be32(src + 32 * c2, w);
            words_from_limbs<8>(w, fe_from_words256_reduce<ModQ>(w).l);
            for (int i = 0; i < 32; i++) m[o + 1 + 32 * c2 + i] = (uint8_t)(w[(31 - i) >> 2] >> (8 * ((31 - i) & 3)));
        }
    } else {   // the 72-byte string as 18 dwords (wide loads), its bytes 3..35 and 39..71 out of registers (byte LOADS at a 72-byte lane stride cost 6 ms per 32 768 proofs)
        uint32_t sw[18];
        const uint32_t* q = (const uint32_t*)src;
#pragma unroll
        for (int i = 0; i < 18; i++) sw[i] = q[i];
#pragma unroll
        for (int i = 0; i < 33; i++) {
            m[o + 1 + i] = (uint8_t)(sw[(3 + i) >> 2] >> (8 * ((3 + i) & 3)));
            m[o + 34 + i] = (uint8_t)(sw[(39 + i) >> 2] >> (8 * ((39 + i) & 3)));
        }
    }
}
// ------------------------------------------------------------------ verifier randomness
ZK_DEV void v_fill(const uint8_t* vseeds, uint64_t gp, uint32_t k, uint32_t w[8]) {  // fill k of the contract, as LE words
    const uint32_t* sd = (const uint32_t*)(vseeds + 32 * gp);
    uint32_t m[16], h[8];
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] = bswap32(sd[i]);
    m[8] = 0, m[9] = k, m[10] = 0x80000000u, m[11] = 0, m[12] = 0, m[13] = 0, m[14] = 0, m[15] = 320;
    sha256_iv(h);
    sha256_compress(h, m);
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = h[7 - i];
}
// engine-private 128-bit randomisers: two per SHA-256(vseed || be64(2^32 + idx) || be64(proof index)); the proof index keeps
// the multipliers of different proofs independent even if a caller reuses a seed (the batched check sums over proofs)
ZK_DEV void v_rho_pair(const uint8_t* vseeds, uint64_t gp, uint32_t idx, Sq& a, Sq& b) {
    const uint32_t* sd = (const uint32_t*)(vseeds + 32 * gp);
    uint32_t m[16], h[8];
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] = bswap32(sd[i]);
    m[8] = 1, m[9] = idx, m[10] = (uint32_t)(gp >> 32), m[11] = (uint32_t)gp, m[12] = 0x80000000u, m[13] = 0, m[14] = 0, m[15] = 384;
    sha256_iv(h);
    sha256_compress(h, m);
    uint32_t wa[8] = {h[0], h[1], h[2], h[3], 0, 0, 0, 0}, wb[8] = {h[4], h[5], h[6], h[7], 0, 0, 0, 0};
    limbs_from_words<8>(a.l, wa);
    limbs_from_words<8>(b.l, wb);
}
// randomisers r[FIRST .. FIRST + N) of slot `tag` (r[2k], r[2k+1] come from pair k): generated where they are used so
// that only a handful are live at a time (all 26 at once cost 234 VGPRs and spilled)
template <int FIRST, int N>
ZK_DEV void v_rho_range(const uint8_t* vseeds, uint64_t gp, uint32_t tag, Sq* out) {
#pragma unroll
    for (int k = FIRST / 2; 2 * k < FIRST + N; k++) {
        Sq a, b;
        v_rho_pair(vseeds, gp, tag | (uint32_t)k, a, b);
        if (2 * k >= FIRST) out[2 * k - FIRST] = a;
        if (2 * k + 1 >= FIRST && 2 * k + 1 < FIRST + N) out[2 * k + 1 - FIRST] = b;
    }
}
// generateIndices draws one byte per attempt and keeps it only if it is below the shrinking range (rndRange with rejection,
// big.ts:171-181): about 256 * (H_80 - H_2) = 890 fills per proof, i.e. 890 SHA-256 blocks in sequence if computed on demand.
// They do not depend on each other, so the first VS_KMAX fills of every proof are hashed in parallel here and k_v_sample
// only walks the stored bytes (falling back to hashing on demand past VS_KMAX).
// Round 5: only the FIRST VK entries of the permutation are ever read (exp.ts:261-264: `indices[j]`, j < secparam = 20), and Algorithm P fixes position i at
// step i -- later steps swap positions >= i only -- so the walk stops after VK steps: ~73 one-byte draws over ranges 80..61 instead of ~890 over 80..3 (the
// reference consumes the rest of its stream too, but nothing observable is drawn from it afterwards: the relation multipliers of multimult.ts only
// randomise a boolean).  VS_KMAX = the fills hashed up front per proof: 2n + 1 + ~73 on average (sd ~14); past it the walk hashes on demand.
#define VS_KMAX V_SAMPLE_FILLS
__global__ void __launch_bounds__(256) k_v_sample_fills(VWork V, uint32_t count, const uint8_t* vseeds, uint64_t first) {
    uint32_t t = gtid();
    if (t >= count * VS_KMAX) return;
    uint32_t p = t / VS_KMAX, k = t % VS_KMAX;  // a proof's bytes are contiguous: k_v_sample reads them 16 at a time
    uint32_t w[8];
    v_fill(vseeds, first + p, k, w);
    V.vbytes[(size_t)p * VS_KMAX + k] = (uint8_t)(w[7] >> 24);
}
struct VByteRow {  // sequential reader of one proof's stored first bytes
    const uint4* row;
    uint4 cur;
    uint32_t blk;
    ZK_DEV uint32_t get(uint32_t k) {
        if ((k >> 4) != blk) blk = k >> 4, cur = row[blk];
        uint32_t sel = (k >> 2) & 3;
        uint32_t wd = sel == 0 ? cur.x : sel == 1 ? cur.y : sel == 2 ? cur.z : cur.w;
        return (wd >> (8 * (k & 3))) & 0xffu;
    }
};
// generateIndices (exp.ts:95-109): runs after verifyMembership's 2n+1 randomScalar draws (gk.ts:223-259)
__global__ void __launch_bounds__(64) k_v_sample(VWork V, uint32_t count, const uint8_t* vseeds, uint64_t first) {
    uint32_t p = gtid();
    if (p >= count) return;
    uint64_t gp = first + p;
    uint32_t k = 0, w[8];
    VByteRow vb;
    vb.row = (const uint4*)(V.vbytes + (size_t)p * VS_KMAX), vb.blk = 0xffffffffu;
    for (uint32_t i = 0; i < 2 * V.n + 1; i++) {  // randomScalar(): retry while >= q (possible only if the fill starts with 0xff)
        for (;;) {
            bool maybe = k >= VS_KMAX || vb.get(k) == 0xff;
            if (!maybe) {
                k++;
                break;
            }
            v_fill(vseeds, gp, k++, w);
            if (!words_geq<8>(w, ModQ::mod32)) break;
        }
    }
    // the permutation lives in LDS (entry i of lane l at i * 64 + l): indexed by data, it would otherwise sit in scratch memory
    __shared__ uint8_t perm_lds[ZK_MAXSEC * 64];
    uint8_t* perm = perm_lds + (threadIdx.x & 63);
#define PERM(i) perm[(i) * 64]
    for (uint32_t i = 0; i < V.sec; i++) PERM(i) = (uint8_t)i;
    // one byte per loop iteration for every lane (a nested retry loop would make each step wait for the unluckiest lane);
    // the stored bytes first, hashing on demand only past VS_KMAX (kept out of the hot loop)
    const uint32_t steps = V.sec - 2 < VK ? V.sec - 2 : VK;   // positions 0 .. VK-1 are final after that many steps
    uint32_t i = 0;
    while (i < steps && k < VS_KMAX) {
        uint32_t range = V.sec - i, v = vb.get(k++);
        if (v < range) {
            uint8_t t = PERM(i);
            PERM(i) = PERM(i + v), PERM(i + v) = t;
            i++;
        }
    }
    while (i < steps) {
        v_fill(vseeds, gp, k++, w);
        uint32_t range = V.sec - i, v = w[7] >> 24;  // first byte of the fill
        if (v < range) {
            uint8_t t = PERM(i);
            PERM(i) = PERM(i + v), PERM(i + v) = t;
            i++;
        }
    }
    // The sampled repetitions with the bit the proof's HEADER claims for each: the header bits are what the proof's layout follows, so every later kernel parses
    // consistently, and they need nothing from the 16 KB hash that recomputes the challenge -- in a small call that hash (0.45 ms in one lane) runs beside R's
    // window table and the sampled points instead of in front of them.  k_v_sample_check compares the two once the hash is there.
    const uint32_t* hb = V.hbits + 4 * p;
    for (uint32_t j = 0; j < VK; j++) {
        uint32_t i = PERM(j);
        V.idx[p * VK + j] = i | (((hb[i >> 5] >> (i & 31)) & 1) << 8);
    }
    V.exp_jz[p] = VK;   // k_v_exp_points lowers it to the first sampled slot whose point is the identity
}
#undef PERM
// verifyExp walks the sampled repetitions in order and throws at the FIRST one that fails (exp.ts:265-346): 'params not found' where the response's type (the
// header bit) does not match the recomputed challenge bit (exp.ts:269-271,301-303), 'T is at infinity' / 'T1 is at infinity' (exp.ts:274,312) where the point is
// the identity.  jm = first slot whose type mismatches (VK: none); the slots behind it are never looked at by the reference: k_v_exp_status (a large chunk: this
// kernel has run by then) or k_v_final (a small chunk: the hash runs beside the points) picks the earlier of the mismatch and the first identity.  Needs V.chal
// (the recomputed challenge) and V.idx.
__global__ void __launch_bounds__(64) k_v_sample_check(VWork V, uint32_t count) {
    const uint32_t p = gtid();
    if (p >= count) return;
    const uint32_t* c = V.chal + 4 * p;
    uint32_t jm = VK;
    for (uint32_t j = 0; j < VK; j++) {
        const uint32_t iv = V.idx[p * VK + j], i = iv & 255, hbit = iv >> 8;
        const uint32_t bit = (c[i >> 5] >> (i & 31)) & 1;
        if (bit != hbit && jm == VK) jm = j;
    }
    V.exp_jm[p] = jm;
}
// verifyExp's exception from the first identity (slot jz, status est; VK / ZK_OK: none) and the first type mismatch (slot jm): a slot's type is looked at before its point
ZK_DEV int32_t v_exp_exception(int32_t est, uint32_t jz, uint32_t jm) { return jm < VK && (est == ZK_OK || jm <= jz) ? ZK_E_PARAMS_NOT_FOUND : est; }

// ------------------------------------------------------------------ Exp: T = alpha*R or T1 = z*R + Q per checked rep (exp.ts:267,299,311)
// split = 1: one lane per checked repetition walks all 65 windows of R's table.  split = 4 (small chunks): four neighbouring lanes take 17 windows each and
// add their partial sums through the wave's cross-lane moves (the table holds every 2^(4w) R: no doublings) -- 17 + 2 additions in a row instead of 65.
__global__ void __launch_bounds__(256) k_v_exp_points(Workspace W, VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first, uint32_t split) {
    const uint32_t tt = gtid();
    const bool live = tt < count * VK * split;
    const uint32_t t = live ? tt / split : count * VK - 1, part = live ? tt % split : 0;   // dead lanes of the last wave mirror the last slot: the cross-lane moves need every lane
    uint32_t p = t / VK;
    uint32_t iv = V.idx[t], i = iv & 255, bit = iv >> 8;
    const uint8_t* pr = proofs + off[first + p];
    P256Pt acc = p256_identity();
    // every sampled slot, parsed by its header bit: where the recomputed challenge disagrees (k_v_sample_check, possibly still running) the slots from the first
    // mismatch on are ignored by v_exp_exception, as the reference never reaches them
    bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8);
    if (good) {
        const uint8_t* rep = pr + rep_offset(V.hbits + 4 * p, i);
        Sn s = ld_scalar_n(rep + 208);
        uint32_t kw[8];
        words_from_limbs<8>(kw, s.l);
        const uint32_t* tab = W.rtab + (size_t)p * rtab_words(RTAB_VERIFY_BITS);
        if (split == 1) acc = p256_rtab_mul(tab, kw, RTAB_VERIFY_BITS);
        else {
            const uint32_t per = (rtab_nwin(RTAB_VERIFY_BITS) + split - 1) / split;
            acc = p256_rtab_mul_range(tab, kw, RTAB_VERIFY_BITS, part * per, per);
        }
    }
    if (split == 4) {   // uniform for the kernel: every lane of the wave takes part
        acc = p256_add(acc, p256_shfl_xor(acc, 1));
        acc = p256_add(acc, p256_shfl_xor(acc, 2));
    }
    if (!live || part != 0) return;
    if (good) {
        if (!bit) {
            P256Pt q;
            q.x = soa_ld<ModQ, 8>(W.Q.x, p), q.y = soa_ld<ModQ, 8>(W.Q.y, p), q.z = soa_ld<ModQ, 8>(W.Q.z, p);
            acc = p256_add(acc, q);
        }
        if (fe_is_zero(fe_reduce(acc.z))) atomicMin(V.exp_jz + p, t % VK);   // 'T is at infinity' / 'T1 is at infinity' (exp.ts:274,312): the first such slot counts
    } else {
        acc.x = fe_const<ModQ, 8>(P256_GX_M), acc.y = fe_const<ModQ, 8>(P256_GY_M), acc.z = fe_one_mont<ModQ>().as<8>();
    }
    uint32_t e = t;  // compact: p * VK + j
    soa_st(W.Tproj.x, e, acc.x), soa_st(W.Tproj.y, e, acc.y), soa_st(W.Tproj.z, e, acc.z);
}
// the exception verifyExp throws, if any: the first sampled slot, in order, with T = identity (bit 1, exp.ts:274), T1 = identity
// (bit 0, exp.ts:312) or a response of the wrong type.  Runs between k_v_exp_points and the normaliser (which maps Z = 0 to (0, 0)).
// have_jm: k_v_sample_check has run (large chunks), exp_st is final; otherwise exp_st holds the first identity only and k_v_final folds the mismatch in.
__global__ void __launch_bounds__(64) k_v_exp_status(Workspace W, VWork V, uint32_t count, bool have_jm) {
    uint32_t p = gtid();
    if (p >= count) return;
    if (p == 0) V.t1_cnt[0] = 0;   // k_v_t1_scalars' list, next on the stream
    if (V.st[p] != ZK_OK || (V.okflags[p] & 8)) return;
    const uint32_t jz = V.exp_jz[p];   // k_v_exp_points: the first sampled slot whose point is the identity
    const int32_t st = jz >= VK ? ZK_OK : (V.idx[p * VK + jz] >> 8) ? ZK_E_T_INF : ZK_E_T1_INF;
    V.exp_st[p] = have_jm ? v_exp_exception(st, jz, V.exp_jm[p]) : st;
}
// T1x = sx*g + r1*h, T1y = sy*g + r2*h for zero-bit slots (exp.ts:329-330); (0, 0) otherwise
__global__ void __launch_bounds__(256) k_v_t1_scalars(Workspace W, VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    uint32_t t = gtid();
    if (t >= count * VK) return;
    uint32_t p = t / VK, j = t % VK;
    uint32_t iv = V.idx[t], i = iv & 255, bit = iv >> 8;
    uint32_t e = t, slot = p * (2 + 2 * W.sec) + 2 * j;
    Sq zero = fe_zero<ModQ>();
    bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8) && V.exp_st[p] == ZK_OK;
    const bool act = good && !bit;
    if (act) {
        const uint8_t* rep = proofs + off[first + p] + rep_offset(V.hbits + 4 * p, i);
        soa_st(W.la.v, slot, soa_ld<ModQ, 1>(W.Tx, e)), soa_st(W.la.r, slot, ld_scalar_q(rep + 272));
        soa_st(W.la.v, slot + 1, soa_ld<ModQ, 1>(W.Ty, e)), soa_st(W.la.r, slot + 1, ld_scalar_q(rep + 304));
    } else {   // commit(0; 0) = the identity: written here, so that the commitment kernel may leave these slots out (launch_tom_commit_list)
        const TomPt id = tom_identity();
        for (uint32_t q = 0; q < 2; q++) {
            soa_st(W.la.v, slot + q, zero), soa_st(W.la.r, slot + q, zero);
            soa_st(W.la.proj.x, slot + q, id.x), soa_st(W.la.proj.y, slot + q, id.y), soa_st(W.la.proj.z, slot + q, id.z);
        }
    }
    // the slots that need a commitment, compacted (one atomic per wave; the order inside the list is irrelevant)
    const uint64_t m = __ballot(act);
    if (m) {
        const uint32_t lane = threadIdx.x & 63;
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(V.t1_cnt, 2u * (uint32_t)__popcll(m));
        base = __shfl(base, __builtin_ctzll(m), 64);
        if (act) {
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            V.t1_act[base + 2 * below] = slot, V.t1_act[base + 2 * below + 1] = slot + 1;
        }
    }
}
// derived commitments needed as hash inputs (pointAdd.ts:210-213,237,250): vd slot (p*VK+j)*5 + {C7, C9, C12, CintX, CintY}
ZK_DEV TomPt v_tom_from_plain(const St& xp, const St& yp) {
    TomPt r;
    Ft2 x = fe_to_mont(xp);
    r.y = fe_to_mont(yp);
    r.x = x * fe_const<ModT, 1>(TOM_S_M);
    r.t = r.x * r.y;
    r.z = fe_one_mont<ModT>().as<2>();
    return r;
}
ZK_DEV TomPt v_tom_from_bytes(const uint8_t* p72) {
    St x, y;
    ld_tom_bytes(p72, x, y);
    return v_tom_from_plain(x, y);
}
ZK_DEV TomPt ld_tom_proj3(const Soa3& a, uint32_t e) {  // (X:Y:Z) without T -> (XZ : YZ : XY : Z^2)
    Ft2 x = soa_ld<ModT, 2>(a.x, e), y = soa_ld<ModT, 2>(a.y, e), z = soa_ld<ModT, 2>(a.z, e);
    TomPt r;
    r.x = x * z, r.y = y * z, r.t = x * y, r.z = z * z;
    return r;
}
__global__ void __launch_bounds__(256) k_v_derived(Workspace W, VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    uint32_t t = gtid();
    if (t >= count * VK * 5) return;
    uint32_t sl = t / 5, k = t % 5, p = sl / VK, j = sl % VK;
    uint32_t iv = V.idx[sl], i = iv & 255, bit = iv >> 8;
    TomPt r = tom_identity();
    bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8) && V.exp_st[p] == ZK_OK;
    if (good && !bit) {
        const uint8_t* pr = proofs + off[first + p];
        const uint8_t* rep = pr + rep_offset(V.hbits + 4 * p, i);
        uint32_t la = p * (2 + 2 * W.sec) + 2 * j;
        // C7 = Px - T1x | C9 = Py - T1y | C12 = T1x - Tx | CintX = Tx + T1x + Px | CintY = C4 + C6 = T1y + Ty: one T1 coordinate
        // commitment and one point of the proof, either of them negated, selected by k so that there is one code path
        uint32_t lai = la + (k == 1 || k == 4 ? 1 : 0);
        TomPt a = ld_tom_proj3(W.la.proj, lai);   // T1x / T1y as k_tom_commit left them: only their sums are hashed, so list A is never normalised
        TomPt b = v_tom_from_bytes(k == 0 ? pr + 160 : k == 1 ? pr + 232 : k == 4 ? rep + 136 : rep + 64);
        if (k < 2) a = tom_neg(a);
        if (k == 2) b = tom_neg(b);
        r = tom_add(a, b);
        if (k == 3) r = tom_add(r, v_tom_from_bytes(pr + 160));
    }
    soa_st(V.vd.proj.x, t, r.x), soa_st(V.vd.proj.y, t, r.y), soa_st(V.vd.proj.z, t, r.z);
}
// the six sub-proof challenges of a zero-bit slot (mult.ts:156, equality.ts:101); thread = h * nslots + slot
__global__ void __launch_bounds__(256) k_v_padd_hash(DevParams P, Workspace W, VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    __shared__ uint32_t lds[16 * 256];
    uint32_t t = gtid(), nsl = count * VK;
    bool live = t < nsl * 6;
    if (!live) t = nsl * 6 - 1;
    uint32_t h = t / nsl, sl = t % nsl, p = sl / VK;
    uint32_t iv = V.idx[sl], i = iv & 255, bit = iv >> 8;
    bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8) && V.exp_st[p] == ZK_OK && !bit;
    ShaStream s;
    s.init(lds, threadIdx.x, 256);
    if (good) {
        const uint8_t* pa = proofs + off[first + p] + rep_offset(V.hbits + 4 * p, i) + ZK_REP_HEAD;
        const uint8_t *c8 = pa, *c10 = pa + 72, *c11 = pa + 144, *c13 = pa + 216;
        uint32_t d = sl * 5;
        if (h < 4) {
            const uint8_t* pts = pa + 288 + 656 * h;
            if (h == 0) {
                absorb_tom_soa(s, V.vd.ax, V.vd.ay, d + 0), absorb_tom_bytes(s, c8);
                uint32_t w[9];
                s.put_byte(4);
                words_from_limbs<9>(w, P.tom_g_aff);
                s.put_be<33>(w);
                words_from_limbs<9>(w, P.tom_g_aff + 9);
                s.put_be<33>(w);
            } else if (h == 1) absorb_tom_bytes(s, c8), absorb_tom_soa(s, V.vd.ax, V.vd.ay, d + 1), absorb_tom_bytes(s, c10);
            else if (h == 2) absorb_tom_bytes(s, c10), absorb_tom_bytes(s, c10), absorb_tom_bytes(s, c11);
            else absorb_tom_bytes(s, c10), absorb_tom_soa(s, V.vd.ax, V.vd.ay, d + 2), absorb_tom_bytes(s, c13);
            for (uint32_t k = 0; k < 6; k++) absorb_tom_bytes(s, pts + 72 * k);
        } else {
            const uint8_t* e = pa + (h == 4 ? 2912 : 3152);
            absorb_tom_bytes(s, h == 4 ? c11 : c13);
            absorb_tom_soa(s, V.vd.ax, V.vd.ay, d + (h == 4 ? 3 : 4));
            absorb_tom_bytes(s, e), absorb_tom_bytes(s, e + 72);
        }
    }
    uint32_t dg[8], c[4];
    s.finish(dg);
    v_challenge_words(dg, c);
    if (live) {
        uint32_t* o = V.vc + ((size_t)sl * 6 + h) * 3;
        o[0] = c[0], o[1] = c[1], o[2] = c[2];
    }
}

// The same six digests per zero-bit slot for a call of a few proofs (<= V_PH_MAXP): one lane per point writes the padded messages -- the points of the proof are
// copied as they stand (a 72-byte encoding holds two 36-byte big-endian coordinates; hashPoints takes the low 33 bytes of each) -- and k_hash.hip's schedule
// and two-lane rounds kernels hash them: 123 -> ~35 us on the chain one verification waits for.  Message m = slot * 6 + h, element k at byte 67 k.
ZK_DEV void ph_tom_bytes(uint8_t* b, const uint32_t* x9, const uint32_t* y9) {   // the 66 coordinate bytes of hashPoints' encoding (33 big-endian each)
    uint32_t w[9];
    words_from_limbs<9>(w, x9);
#pragma unroll
    for (int i = 0; i < 33; i++) b[i] = (uint8_t)(w[(32 - i) >> 2] >> (8 * ((32 - i) & 3)));
    words_from_limbs<9>(w, y9);
#pragma unroll
    for (int i = 0; i < 33; i++) b[33 + i] = (uint8_t)(w[(32 - i) >> 2] >> (8 * ((32 - i) & 3)));
}
__global__ void __launch_bounds__(256) k_v_padd_msg(DevParams P, Workspace W, VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    const uint32_t t = gtid(), nsl = count * VK;
    if (t >= nsl * 6 * 10) return;
    const uint32_t m = t / 10, k = t % 10, sl = m / 6, h = m % 6, p = sl / VK;
    const uint32_t iv = V.idx[sl], i = iv & 255, bit = iv >> 8;
    const bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8) && V.exp_st[p] == ZK_OK && !bit;
    const uint32_t ne = !good ? 0 : h < 4 ? 9 : 4;
    uint8_t* msg = V.ph_msg + (size_t)m * V_PH_BLOCKS * 64;
    if (k == 9) {   // padding: 0x80, zeros, the bit length in eight bytes
        const uint32_t len = 67 * ne, nb = (len + 9 + 63) / 64;
        V.ph_nblk[m] = (uint8_t)nb;
        msg[len] = 0x80;
        for (uint32_t j = len + 1; j < nb * 64 - 8; j++) msg[j] = 0;
        const uint64_t bits = (uint64_t)len * 8;
        for (int j = 0; j < 8; j++) msg[nb * 64 - 1 - j] = (uint8_t)(bits >> (8 * j));
        return;
    }
    if (k >= ne) return;
    const uint8_t* pa = proofs + off[first + p] + rep_offset(V.hbits + 4 * p, i) + ZK_REP_HEAD;
    const uint8_t *c8 = pa, *c10 = pa + 72, *c11 = pa + 144, *c13 = pa + 216;
    const uint32_t d = sl * 5;
    const uint8_t* src = nullptr;   // a point of the proof ...
    uint32_t vd = 0xffffffffu;      // ... or a derived commitment (V.vd) ...
    bool g = false;                 // ... or the generator
    if (h < 4) {
        if (k >= 3) src = pa + 288 + 656 * h + 72 * (k - 3);
        else if (h == 0) {
            if (k == 0) vd = d + 0;
            else if (k == 1) src = c8;
            else g = true;
        } else if (h == 1) {
            if (k == 0) src = c8;
            else if (k == 1) vd = d + 1;
            else src = c10;
        } else if (h == 2) src = k < 2 ? c10 : c11;
        else {
            if (k == 0) src = c10;
            else if (k == 1) vd = d + 2;
            else src = c13;
        }
    } else {
        const uint8_t* e = pa + (h == 4 ? 2912 : 3152);
        if (k == 0) src = h == 4 ? c11 : c13;
        else if (k == 1) vd = d + (h == 4 ? 3 : 4);
        else src = e + 72 * (k - 2);
    }
    uint8_t b[66];   // every load before the first store (the compiler cannot know that the proof and the message do not overlap)
    if (src) {
#pragma unroll
        for (int j = 0; j < 33; j++) b[j] = src[3 + j], b[33 + j] = src[39 + j];
    } else if (g) ph_tom_bytes(b, P.tom_g_aff, P.tom_g_aff + 9);
    else ph_tom_bytes(b, soa_ld<ModT, 1>(V.vd.ax, vd).l, soa_ld<ModT, 1>(V.vd.ay, vd).l);
    uint8_t* __restrict__ o = msg + 67 * k;
    o[0] = 4;
#pragma unroll
    for (int j = 0; j < 66; j++) o[1 + j] = b[j];
}

// ------------------------------------------------------------------ GK total (gk.ts:239-250), fold form:
// layer' [i] = (x - f_j) * layer[2i] + f_j * layer[2i+1]; tile of 2^T ring elements per workgroup, levels through LDS
#define VGK_T 13   // 256 lanes x 32 elements per tile
__global__ void __launch_bounds__(256) k_v_gk_fg(VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    uint32_t t = gtid();
    if (t >= count * V.n) return;
    uint32_t p = t / V.n, j = t % V.n;
    Sq f = fe_zero<ModQ>(), g = fe_zero<ModQ>();
    if (V.st[p] == ZK_OK && !(V.okflags[p] & 8)) {
        const uint8_t* sc = v_gk_base(V, proofs + off[first + p], p) + 4 * 72 * V.n;
        f = ld_scalar_q(sc + 32 * j);
        uint32_t xw[8] = {V.gkx[3 * p], V.gkx[3 * p + 1], V.gkx[3 * p + 2], 0, 0, 0, 0, 0};
        Sq x;
        limbs_from_words<8>(x.l, xw);
        g = fe_sub_mod(x, f);
    }
    // gk_f <- rho_j = f_j / g_j (Montgomery), gk_g <- scale factor g_j; if g_j = 0 (x = f_j): rho_j = 0 with the roles of
    // even/odd swapped (flag in limb 8 bit 29 of gk_g is not needed: rho = 0 and scale = f_j, see v_gk_pair)
    bool gz = fe_is_zero(g);
    Fe<ModQ, 2> gm = fe_to_mont(gz ? f : g), fm = fe_to_mont(f);
    Fe<ModQ, 2> rho = gz ? fe_zero<ModQ>().as<2>() : fm * fe_inv<ModQ>(gm);
    soa_st(V.gk_f, j * V.C + p, rho);
    soa_st(V.gk_g, j * V.C + p, gm);
    V.gk_swap[j * V.C + p] = gz ? 1u : 0u;
}
// one pair: (x - f) ev + f od = g (ev + rho od); the common factor g_j of every level is applied once at the end.
// Values stay lazily reduced: each level adds < 2q, so after n <= 28 levels the bound is < 64 q.
ZK_DEV Fe<ModQ, 64> v_gk_pair(const Fe<ModQ, 64>& ev, const Fe<ModQ, 64>& od, const Fe<ModQ, 2>& rho, bool swap) {
    Fe<ModQ, 64> r;
    if (swap) return od;  // g_j = 0: the even branch vanishes, scale = f_j
    auto t = rho * od + ev;  // < 66q by type, < 2 (j+1) q by induction
#pragma unroll
    for (int l = 0; l < NLIMB; l++) r.l[l] = t.l[l];
    return r;
}
// total = (prod_j scale_j) * folded value   (scale_j = g_j, or f_j where g_j = 0)
ZK_DEV Sq v_gk_scale(const VWork& V, uint32_t p, const Fe<ModQ, 64>& r) {
    Fe<ModQ, 2> acc = fe_reduce(r);  // plain value, < 2q
    for (uint32_t j = 0; j < V.n; j++) acc = acc * soa_ld<ModQ, 2>(V.gk_g, j * V.C + p);  // Montgomery factor: stays plain
    return fe_canon(acc);
}
ZK_DEV void v_gk_lds_levels(uint32_t*& A, uint32_t*& B, uint32_t cnt, uint32_t j0, uint32_t j1, const VWork& V, uint32_t p, uint32_t stride) {
    for (uint32_t j = j0; j < j1; j++) {
        uint32_t nout = cnt >> 1;
        Fe<ModQ, 2> rho = soa_ld<ModQ, 2>(V.gk_f, j * V.C + p);
        bool swap = V.gk_swap[j * V.C + p] != 0;
        for (uint32_t m = threadIdx.x; m < nout; m += blockDim.x) {
            Fe<ModQ, 64> ev, od;
            for (int l = 0; l < NLIMB; l++) ev.l[l] = A[l * stride + 2 * m], od.l[l] = A[l * stride + 2 * m + 1];
            auto r = v_gk_pair(ev, od, rho, swap);
            for (int l = 0; l < NLIMB; l++) B[l * stride + m] = r.l[l];
        }
        __syncthreads();
        uint32_t* tmp = A;
        A = B, B = tmp;
        cnt = nout;
    }
}
// depth-first register fold of 2^LEV consecutive ring elements (one modmul per pair)
// The contraction sum_i v_i prod_j w_{j, bit_j(i)} can fold the index bits in any order.  A lane's 2^LEV elements are
// `stride` apart (lanes read consecutive elements: coalesced), so the register phase folds index bits bit0 .. bit0+LEV-1.
template <int LEV>
struct VFold {
    static ZK_DEV Fe<ModQ, 64> run(const VWork& V, const Soa& ring, uint32_t p, uint32_t base, uint32_t stride, uint32_t bit0) {
        Fe<ModQ, 64> ev = VFold<LEV - 1>::run(V, ring, p, base, stride, bit0);
        Fe<ModQ, 64> od = VFold<LEV - 1>::run(V, ring, p, base + (stride << (LEV - 1)), stride, bit0);
        uint32_t j = bit0 + LEV - 1;
        return v_gk_pair(ev, od, soa_ld<ModQ, 2>(V.gk_f, j * V.C + p), V.gk_swap[j * V.C + p] != 0);
    }
};
template <>
struct VFold<0> {
    static ZK_DEV Fe<ModQ, 64> run(const VWork&, const Soa& ring, uint32_t, uint32_t base, uint32_t, uint32_t) { return soa_ld<ModQ, 1>(ring, base).as<64>(); }
};
template <int RL>
__global__ void __launch_bounds__(256) k_v_gk_tile(VWork V, Soa ring, uint32_t T, uint32_t ntiles, Soa res) {
    __shared__ uint32_t bufA[NLIMB * 256];
    __shared__ uint32_t bufB[NLIMB * 256];
    uint32_t p = blockIdx.x / ntiles, tile = blockIdx.x % ntiles, t = threadIdx.x;
    uint32_t lanes = 1u << (T - RL), stride = 256;
    if (t < lanes) {
        Fe<ModQ, 64> v = VFold<RL>::run(V, ring, p, (tile << T) + t, lanes, T - RL);
        for (int l = 0; l < NLIMB; l++) bufA[l * stride + t] = v.l[l];
    }
    __syncthreads();
    uint32_t *A = bufA, *B = bufB;
    v_gk_lds_levels(A, B, lanes, 0, T - RL, V, p, stride);  // adjacent lanes differ in index bit 0, then 1, ...
    if (t == 0) {
        Fe<ModQ, 64> r;
        for (int l = 0; l < NLIMB; l++) r.l[l] = A[l * stride];
        if (ntiles == 1) soa_st(V.gk_total, p, v_gk_scale(V, p, r));
        else soa_st(res, p * ntiles + tile, fe_canon(fe_reduce(r)));
    }
}
// finish pass: workgroup (proof, group) folds gsz <= 512 consecutive tile values through log2(gsz) levels (dynamic LDS)
#define VGK_FIN 512u
__global__ void __launch_bounds__(256) k_v_gk_finish(VWork V, uint32_t Tin, uint32_t npoly, uint32_t gsz, Soa src, Soa dst) {
    extern __shared__ uint32_t v_lds_dyn[];  // two planes of gsz elements (9 limbs each): sized by the launch
    uint32_t* bufA = v_lds_dyn;
    uint32_t* bufB = v_lds_dyn + NLIMB * gsz;
    uint32_t ngroups = npoly / gsz;
    uint32_t p = blockIdx.x / ngroups, g = blockIdx.x % ngroups, stride = gsz;
    for (uint32_t m = threadIdx.x; m < gsz; m += blockDim.x) {
        Sq c = soa_ld<ModQ, 1>(src, p * npoly + g * gsz + m);
        for (int l = 0; l < NLIMB; l++) bufA[l * stride + m] = c.l[l];
    }
    __syncthreads();
    uint32_t lv = 0;
    while ((1u << lv) < gsz) lv++;
    uint32_t *A = bufA, *B = bufB;
    v_gk_lds_levels(A, B, gsz, Tin, Tin + lv, V, p, stride);
    if (threadIdx.x == 0) {
        Fe<ModQ, 64> r;
        for (int l = 0; l < NLIMB; l++) r.l[l] = A[l * stride];
        if (ngroups == 1) soa_st(V.gk_total, p, v_gk_scale(V, p, r));
        else soa_st(dst, p * ngroups + g, fe_canon(fe_reduce(r)));
    }
}
// tiny rings (n < 3): one thread per proof
__global__ void k_v_gk_small(VWork V, Soa ring, uint32_t count) {
    uint32_t p = gtid();
    if (p >= count) return;
    Fe<ModQ, 64> v[4];
    uint32_t N = 1u << V.n;
    for (uint32_t i = 0; i < 4; i++) v[i] = soa_ld<ModQ, 1>(ring, i < N ? i : 0).as<64>();
    for (uint32_t j = 0; j < V.n; j++) {
        Fe<ModQ, 2> rho = soa_ld<ModQ, 2>(V.gk_f, j * V.C + p);
        bool swap = V.gk_swap[j * V.C + p] != 0;
        for (uint32_t i = 0; i < (N >> (j + 1)); i++) v[i] = v_gk_pair(v[2 * i], v[2 * i + 1], rho, swap);
    }
    soa_st(V.gk_total, p, v_gk_scale(V, p, v[0]));
}
void launch_v_gk_total(hipStream_t s, const VWork& V, const Soa& ring, const uint32_t* etab, const int8_t* kdig, uint32_t count, uint32_t N, const uint8_t* proofs, const uint64_t* off, uint64_t first, const Soa& res, const Soa& res2) {
    uint32_t nt = count * V.n;
    hipLaunchKernelGGL(k_v_gk_fg, dim3((nt + 255) / 256), dim3(256), 0, s, V, count, proofs, off, first);
    if (V.n < 3) {
        hipLaunchKernelGGL(k_v_gk_small, dim3((count + 63) / 64), dim3(64), 0, s, V, ring, count);
        return;
    }
    uint32_t T = etab ? 8 : V.n < VGK_T ? V.n : VGK_T, ntiles = N >> T;
    if (etab && kdig && V.n >= GKM_MINN) launch_v_gk_block_mfma(s, V, kdig, ntiles, count, (int8_t*)V.gk_csub, res);   // the same sums as int8 matrix products (k_gk_mfma.hip)
    else if (etab) launch_v_gk_block_stage(s, V, etab, ntiles, count, V.gk_csub, res);  // 8 low index bits through table E (k_gk.hip)
    else if (V.n >= 5) hipLaunchKernelGGL(k_v_gk_tile<5>, dim3(count * ntiles), dim3(256), 0, s, V, ring, T, ntiles, res);
    else hipLaunchKernelGGL(k_v_gk_tile<3>, dim3(count * ntiles), dim3(256), 0, s, V, ring, T, ntiles, res);
    Soa src = res, dst = res2;
    while (ntiles > 1) {
        uint32_t gsz = ntiles < VGK_FIN ? ntiles : VGK_FIN, ngroups = ntiles / gsz, lv = 0;
        while ((1u << lv) < gsz) lv++;
        hipLaunchKernelGGL(k_v_gk_finish, dim3(count * ngroups), dim3(256), sizeof(uint32_t) * 2 * NLIMB * gsz, s, V, T, ntiles, gsz, src, dst);
        Soa tmp = src;
        src = dst, dst = tmp, T += lv, ntiles = ngroups;
    }
}

// ------------------------------------------------------------------ term construction
// A Tom term = niels form of the point on the a=1 image (x', y, d'x'y, Montgomery) + a plain scalar.  Group gidx owns
// terms [k * ngroups + gidx]; the first n256 terms of a group have 256-bit scalars, the rest 128-bit.
ZK_DEV void vt_st_at(uint32_t* pts, uint32_t idx, const Ft2& x, const Ft2& y, const Ft2& dt) {
    uint32_t w[28];
#pragma unroll
    for (int l = 0; l < 9; l++) w[l] = x.l[l], w[9 + l] = y.l[l], w[18 + l] = dt.l[l];
    w[27] = 0;
    uint4* q = (uint4*)(pts + (size_t)idx * VT_ENTRY_WORDS);
#pragma unroll
    for (int i = 0; i < 7; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
ZK_DEV void vt_st(const VTerms& L, uint32_t idx, const Ft2& x, const Ft2& y, const Ft2& dt) { vt_st_at(L.pts, idx, x, y, dt); }
// The same store by a whole wave (every lane of every wave of the workgroup must call it; valid = false: nothing of this lane's is stored): the 64 entries go
// through LDS and leave as FULL 128-byte lines, eight lanes per entry -- one lane's seven 16-byte stores touch 64 different lines per instruction, which cost
// k_v_slot_points 0.6 ms per 32 768 proofs.  stage: 64 * 29 words of LDS of this wave.
ZK_DEV void vt_st_wave(uint32_t* pts, uint32_t idx, bool valid, const Ft2& x, const Ft2& y, const Ft2& dt, uint32_t* stage) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t* m = stage + lane * 29;
#pragma unroll
    for (int l = 0; l < 9; l++) m[l] = x.l[l], m[9 + l] = y.l[l], m[18 + l] = dt.l[l];
    m[27] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < 8; r++) {
        const uint32_t j = r * 8 + (lane >> 3), part = lane & 7;
        const uint32_t jidx = (uint32_t)__shfl((int)idx, (int)j, 64);
        const bool jvalid = __shfl((int)valid, (int)j, 64) != 0;
        const uint32_t* e = stage + j * 29 + part * 4;
        const uint4 v = part < 7 ? make_uint4(e[0], e[1], e[2], e[3]) : make_uint4(0, 0, 0, 0);
        if (jvalid) ((uint4*)(pts + (size_t)jidx * VT_ENTRY_WORDS))[part] = v;
    }
    __syncthreads();
}
ZK_DEV void vt_st_identity(const VTerms& L, uint32_t idx) { vt_st(L, idx, fe_zero<ModT>().as<2>(), fe_one_mont<ModT>().as<2>(), fe_zero<ModT>().as<2>()); }
ZK_DEV void vt_ld_xy(const VTerms& L, uint32_t idx, Ft2& x, Ft2& y) {
    const uint4* q = (const uint4*)(L.pts + (size_t)idx * VT_ENTRY_WORDS);
    uint32_t w[20];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint4 v = q[i];
        w[4 * i] = v.x, w[4 * i + 1] = v.y, w[4 * i + 2] = v.z, w[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int l = 0; l < 9; l++) x.l[l] = w[l], y.l[l] = w[9 + l];
}
ZK_DEV void put_term(const VTerms& L, uint32_t idx, const St& xp, const St& yp, bool negate, const Sq& sc) {
    Ft2 x = fe_to_mont(xp) * fe_const<ModT, 1>(TOM_S_M);
    Ft2 y = fe_to_mont(yp);
    if (negate) x = fe_reduce(fe_neg(x));
    Ft2 dt = (x * y) * fe_const<ModT, 1>(TOM_D1_M);
    vt_st(L, idx, x, y, dt), soa_st(L.sc, idx, sc);
}
ZK_DEV void put_term_bytes(const VTerms& L, uint32_t idx, const uint8_t* p72, bool negate, const Sq& sc) {
    St x, y;
    ld_tom_bytes(p72, x, y);
    put_term(L, idx, x, y, negate, sc);
}
ZK_DEV void put_term_point_null(const VTerms& L, uint32_t idx) {  // identity; the scalar is written by the scalar kernel
    vt_st_identity(L, idx);
}
// niels form of the 72-byte point at p72 (nullptr: the identity)
ZK_DEV void term_point(const uint8_t* p72, bool negate, Ft2& x, Ft2& y, Ft2& dt) {
    if (!p72) {
        x = fe_zero<ModT>().as<2>(), y = fe_one_mont<ModT>().as<2>(), dt = fe_zero<ModT>().as<2>();
        return;
    }
    St xp, yp;
    ld_tom_bytes(p72, xp, yp);
    x = fe_to_mont(xp) * fe_const<ModT, 1>(TOM_S_M);
    y = fe_to_mont(yp);
    if (negate) x = fe_reduce(fe_neg(x));
    dt = (x * y) * fe_const<ModT, 1>(TOM_D1_M);
}
ZK_DEV void put_term_point(const VTerms& L, uint32_t idx, const uint8_t* p72, bool negate) {
    Ft2 x, y, dt;
    term_point(p72, negate, x, y, dt);
    vt_st(L, idx, x, y, dt);
}
ZK_DEV void put_term_null(const VTerms& L, uint32_t idx) {  // identity with scalar 0
    vt_st_identity(L, idx), soa_st(L.sc, idx, fe_zero<ModQ>());
}
// mod-q product between two scheduling fences: the term kernels are chains of independent products, and without the fences the
// compiler interleaves them until the live set spills (256 VGPRs + scratch); in order they fit two or more waves per SIMD.
ZK_DEV Sq mulq(const Sq& a, const Sq& b) {
    Sq x = a;
    for (int l = 0; l < NLIMB; l++) asm volatile("" : "+v"(x.l[l]));
    Sq r = fe_mul_mod(x, b);
    for (int l = 0; l < NLIMB; l++) asm volatile("" : "+v"(r.l[l]));
    return r;
}
ZK_DEV Sq addq(const Sq& a, const Sq& b) { return fe_add_mod(a, b); }
ZK_DEV Sq chalq(const uint32_t* c3) {
    uint32_t w[8] = {c3[0], c3[1], c3[2], 0, 0, 0, 0, 0};
    Sq r;
    limbs_from_words<8>(r.l, w);
    return r;
}
struct SlotAcc {  // coefficients accumulated for shared points of one slot
    Sq g, h, c8, c10, c11, c13, w7, w9, w12, wX, wY;
};
// aggregateMult (mult.ts:158-173) with randomisers r[0..4]; point coefficients returned through references.
// pts: C4, Ax, Ay, Az, A41, A42 then 7 scalars.  Term slots: 256-bit C4 at i256, 128-bit A's at i128..i128+4.  Scalars only: the
// points of the same terms are converted by k_v_slot_points.
ZK_DEV void v_mult(const VTerms& L, uint32_t gidx, uint32_t ng, uint32_t i256, uint32_t i128, const uint8_t* pi, const Sq& c, const Sq* r,
                   Sq& wCx, Sq& wCy, Sq& wCz, Sq& g, Sq& h) {
    const uint8_t* sc = pi + 432;
    {
        Sq tx = ld_scalar_q(sc);
        wCx = addq(wCx, mulq(r[0], c));
        wCy = addq(wCy, addq(mulq(r[1], c), mulq(r[4], tx)));
        wCz = addq(wCz, mulq(r[2], c));
        g = addq(g, mulq(r[0], tx));
    }
    g = addq(g, mulq(r[1], ld_scalar_q(sc + 32)));
    g = addq(g, mulq(addq(r[2], r[3]), ld_scalar_q(sc + 64)));
    h = addq(h, mulq(r[0], ld_scalar_q(sc + 96)));
    h = addq(h, mulq(r[1], ld_scalar_q(sc + 128)));
    h = addq(h, mulq(r[2], ld_scalar_q(sc + 160)));
    h = addq(h, mulq(r[3], ld_scalar_q(sc + 192)));
    soa_st(L.sc, i256 * ng + gidx, mulq(addq(r[3], r[4]), c));  // C4: (r4 + r5) c
    for (int k = 0; k < 5; k++) soa_st(L.sc, (i128 + k) * ng + gidx, r[k]);  // r_k * (-A)
}
ZK_DEV void v_eq(const VTerms& L, uint32_t gidx, uint32_t ng, uint32_t i128, const uint8_t* pi, const Sq& c, const Sq* r, Sq& wC1, Sq& wC2, Sq& g, Sq& h) {
    const uint8_t* sc = pi + 144;
    wC1 = addq(wC1, mulq(r[0], c));
    wC2 = addq(wC2, mulq(r[1], c));
    g = addq(g, mulq(addq(r[0], r[1]), ld_scalar_q(sc)));
    h = addq(h, mulq(r[0], ld_scalar_q(sc + 32)));
    h = addq(h, mulq(r[1], ld_scalar_q(sc + 64)));
    soa_st(L.sc, i128 * ng + gidx, r[0]);
    soa_st(L.sc, (i128 + 1) * ng + gidx, r[1]);
}
// Layout of a slot group (group gidx = slot): 256-bit terms 0..9 = C8, C10, C11, C13, Tx, Ty, C4 x4; 128-bit terms 10..35 =
// 4 x (Ax, Ay, Az, A41, A42), pix A1, A2, piy A1, A2, then Tx, Ty of a bit-1 slot.
//
// Points: one thread per (term, slot), term-major; a wave stores its 64 entries together (vt_st_wave).  Three Tom products per thread.
__global__ void __launch_bounds__(256) k_v_slot_points(VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    __shared__ uint32_t stage[4][64 * 29];
    const uint32_t t0 = gtid(), ns = count * VK, ng = V.C * VK;
    const bool valid = t0 < ns * V_SLOT_TERMS;
    const uint32_t t = valid ? t0 : ns * V_SLOT_TERMS - 1;   // (the last workgroup's spare lanes mirror the last term: the store below is the whole wave's)
    uint32_t k = t / ns, sl = t % ns, p = sl / VK;
    uint32_t iv = V.idx[sl], i = iv & 255, bit = iv >> 8;
    const VTerms& L = V.slot_terms;
    bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8) && V.exp_st[p] == ZK_OK;
    const uint8_t* src = nullptr;
    bool neg = false;
    if (good) {
        const uint8_t* rep = proofs + off[first + p] + rep_offset(V.hbits + 4 * p, i);
        const uint8_t* pa = rep + ZK_REP_HEAD;
        if (bit) {
            if (k >= 34) src = rep + (k == 34 ? 64 : 136), neg = true;
        } else if (k < 4) src = pa + 72 * k;
        else if (k < 6) src = rep + (k == 4 ? 64 : 136);
        else if (k < 10) src = pa + 288 + 656 * (k - 6);
        else if (k < 30) src = pa + 288 + 656 * ((k - 10) / 5) + 72 * ((k - 10) % 5 + 1), neg = true;
        else if (k < 34) src = pa + (k < 32 ? 2912 : 3152) + 72 * (k & 1), neg = true;
    }
    Ft2 x, y, dt;
    term_point(src, neg, x, y, dt);
    vt_st_wave(L.pts, k * ng + sl, valid, x, y, dt, stage[threadIdx.x >> 6]);
}
// Scalars: one thread per checked slot: the scalars of the slot's terms, partial sums for shared points, and the slot's P-256
// contribution.  Only mod-q arithmetic here; sums that are complete are stored at once so that few values stay live.
__global__ void __launch_bounds__(64, 2) k_v_slot_terms(Workspace W, VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, const uint8_t* vseeds, uint64_t first) {
    uint32_t sl = gtid(), ng = V.C * VK;
    if (sl >= count * VK) return;
    uint32_t p = sl / VK, j = sl % VK;
    uint32_t iv = V.idx[sl], i = iv & 255, bit = iv >> 8;
    const VTerms& L = V.slot_terms;
    bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8) && V.exp_st[p] == ZK_OK;
    Sq zero = fe_zero<ModQ>();
    uint32_t pa_idx = p * VK + j;  // P-256 A-term index
    // slot classes for k_v_straus: a zero-bit rep has 36 live terms, every other slot only terms 34, 35 (128-bit)
    V.slot_class[sl] = good && !bit ? 1 : 0;
    if (!good) {
        soa_st(V.pa_x, pa_idx, fe_const<ModQ, 2>(P256_GX_M)), soa_st(V.pa_y, pa_idx, fe_const<ModQ, 2>(P256_GY_M)), soa_st(V.pa_sc, pa_idx, fe_zero<ModN>());
        for (uint32_t k = 0; k < V_SLOT_TERMS; k++) soa_st(L.sc, k * ng + sl, zero);
        soa_st(V.sSg, sl, zero), soa_st(V.sSh, sl, zero), soa_st(V.sSkx, sl, zero), soa_st(V.sSky, sl, zero);
        soa_st(V.sSR, sl, fe_zero<ModN>()), soa_st(V.sSH, sl, fe_zero<ModN>()), soa_st(V.sSL, sl, fe_zero<ModN>());
        return;
    }
    const uint8_t* pr = proofs + off[first + p];
    const uint8_t* rep = pr + rep_offset(V.hbits + 4 * p, i);
    uint64_t gp = first + p;
    const uint32_t tag = j << 8;
    // ---- P-256 relation (exp.ts:270-276 / 305-317) with randomiser r[24] (mod n): the T term is (rho * s) * R
    {
        Sq r24[1];
        v_rho_range<24, 1>(vseeds, gp, tag, r24);
        Sn rn;
        for (int l = 0; l < NLIMB; l++) rn.l[l] = r24[0].l[l];
        soa_st(V.sSR, sl, fe_mul_mod(rn, ld_scalar_n(rep + 208)));
        soa_st(V.sSH, sl, fe_mul_mod(rn, ld_scalar_n(rep + 240)));
        soa_st(V.sSL, sl, bit ? fe_zero<ModN>() : rn);
        // -rho * A  ==  rho * (-A)
        uint32_t xw[8], yw[8];
        load_be32(rep, xw);
        load_be32(rep + 32, yw);
        Fq2 ax = fe_to_mont(fe_from_words256_reduce<ModQ>(xw));
        Fq2 ay = fe_reduce(fe_neg(fe_to_mont(fe_from_words256_reduce<ModQ>(yw))));
        soa_st(V.pa_x, pa_idx, ax), soa_st(V.pa_y, pa_idx, ay), soa_st(V.pa_sc, pa_idx, rn);
    }
    Sq Sg, Sh;
    if (bit) {
        // relTx, relTy (exp.ts:287-297): r0 (sx g + beta2 h - Tx), r1 (sy g + beta3 h - Ty); sx, sy = affine T
        Sq r[2];
        v_rho_range<0, 2>(vseeds, gp, tag, r);
        Sg = addq(mulq(r[0], soa_ld<ModQ, 1>(W.Tx, sl)), mulq(r[1], soa_ld<ModQ, 1>(W.Ty, sl)));
        Sh = addq(mulq(r[0], ld_scalar_q(rep + 272)), mulq(r[1], ld_scalar_q(rep + 304)));
        for (uint32_t k = 0; k < 34; k++) soa_st(L.sc, k * ng + sl, zero);
        soa_st(L.sc, 34 * ng + sl, r[0]);
        soa_st(L.sc, 35 * ng + sl, r[1]);
        soa_st(V.sSkx, sl, zero), soa_st(V.sSky, sl, zero);
    } else {
        const uint8_t* pa = rep + ZK_REP_HEAD;
        const uint32_t* cw = V.vc + (size_t)sl * 18;
        Sq w7 = zero, w8 = zero, w9 = zero, w10 = zero, w11 = zero, w12 = zero, w13 = zero, w14 = zero, wX = zero, wY = zero;
        Sg = zero, Sh = zero;
        // pointAdd.ts:215-255
        Sq r[5];
        v_rho_range<0, 5>(vseeds, gp, tag, r);
        v_mult(L, sl, ng, 6, 10, pa + 288, chalq(cw), r, w7, w8, w14, Sg, Sh);                      // pi8 : (C7, C8, C14 = g)
        Sg = addq(Sg, w14);
        v_rho_range<5, 5>(vseeds, gp, tag, r);
        v_mult(L, sl, ng, 7, 15, pa + 288 + 656, chalq(cw + 3), r, w8, w9, w10, Sg, Sh);             // pi10: (C8, C9, C10)
        soa_st(L.sc, 0 * ng + sl, w8);
        {                                                                                             // pi11: (C10, C10, C11)
            Sq wa = zero, wb = zero;
            v_rho_range<10, 5>(vseeds, gp, tag, r);
            v_mult(L, sl, ng, 8, 20, pa + 288 + 2 * 656, chalq(cw + 6), r, wa, wb, w11, Sg, Sh);
            w10 = addq(w10, addq(wa, wb));
        }
        v_rho_range<15, 2>(vseeds, gp, tag, r);
        v_eq(L, sl, ng, 30, pa + 2912, chalq(cw + 12), r, w11, wX, Sg, Sh);                            // pix : (C11, C3 + C1 + C2)
        soa_st(L.sc, 2 * ng + sl, w11);
        v_rho_range<17, 5>(vseeds, gp, tag, r);
        v_mult(L, sl, ng, 9, 25, pa + 288 + 3 * 656, chalq(cw + 9), r, w10, w12, w13, Sg, Sh);         // pi13: (C10, C12, C13)
        soa_st(L.sc, 1 * ng + sl, w10);
        v_rho_range<22, 2>(vseeds, gp, tag, r);
        v_eq(L, sl, ng, 32, pa + 3152, chalq(cw + 15), r, w13, wY, Sg, Sh);                            // piy : (C13, C4 + C6)
        soa_st(L.sc, 3 * ng + sl, w13);
        // redistribute the derived commitments: C7 = Px - T1x, C9 = Py - T1y, C12 = T1x - Tx, CintX = Tx + T1x + Px,
        // CintY = T1y + Ty, C14 = g, T1x = sx g + r1 h, T1y = sy g + r2 h; sx, sy = T1 + Q of the slot
        soa_st(L.sc, 4 * ng + sl, fe_sub_mod(wX, w12));  // Tx
        soa_st(L.sc, 5 * ng + sl, wY);                   // Ty
        soa_st(L.sc, 34 * ng + sl, zero), soa_st(L.sc, 35 * ng + sl, zero);
        soa_st(V.sSkx, sl, addq(w7, wX)), soa_st(V.sSky, sl, w9);
        Sq u1 = addq(fe_sub_mod(w12, w7), wX), u2 = fe_sub_mod(wY, w9);
        Sg = addq(Sg, addq(mulq(u1, soa_ld<ModQ, 1>(W.Tx, sl)), mulq(u2, soa_ld<ModQ, 1>(W.Ty, sl))));
        Sh = addq(Sh, addq(mulq(u1, ld_scalar_q(rep + 272)), mulq(u2, ld_scalar_q(rep + 304))));
    }
    soa_st(V.sSg, sl, Sg), soa_st(V.sSh, sl, Sh);
}
// Class-sorted slot order of a range of slots that goes to the per-proof sums (k_v_straus): local ids, class 1 from the front,
// class 0 from the back, so that waves are homogeneous.
__global__ void __launch_bounds__(256) k_v_slot_perm(const uint8_t* __restrict__ slot_class, uint32_t nslots, uint32_t* perm, uint32_t* cnt) {
    uint32_t sl = gtid();
    if (sl >= nslots) return;
    if (slot_class[sl]) perm[atomicAdd(&cnt[0], 1u)] = sl;
    else perm[nslots - 1 - atomicAdd(&cnt[1], 1u)] = sl;
}
// (cnt: zeroed by k_v_proof_sums -- a memset node between two kernels of a stream costs a call of one proof 30-50 us of idle GPU)
void launch_v_slot_perm(hipStream_t s, const uint8_t* slot_class, uint32_t nslots, uint32_t* perm, uint32_t* cnt) {
    if (!nslots) return;
    hipLaunchKernelGGL(k_v_slot_perm, dim3((nslots + 255) / 256), dim3(256), 0, s, slot_class, nslots, perm, cnt);
}
// GK relations (gk.ts:223-259) -> groups of the gk list (4 per pair of bit positions: cl, cd 256-bit; ca, cb 128-bit) and the
// per-proof totals for the shared points.
// gk group q (q < ceil(n/2)) holds i = 2q, 2q+1: 256-bit terms {cl_i, cd_i} x2 = 0..3, 128-bit {ca_i, cb_i} x2 = 4..7.
// misc group (index = p): 256-bit terms: 0 = Px (membership coefficient), 1 = Px (Exp), 2 = Py (Exp).
//
// Points: one thread per (term, group), term-major.  which: bit 0 = the membership proof's points (the gk groups and misc term 0: nothing of verifyExp in them),
// bit 1 = misc terms 1, 2 (Px, Py of the Exp relations: they follow exp_st) -- a small chunk converts the former beside the P-256 front end.
__global__ void __launch_bounds__(256) k_v_proof_points(VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first, uint32_t which) {
    uint32_t t = gtid(), n = V.n, nq = (n + 1) / 2, ngk = V.C * nq, nm = V.C, ngl = count * nq;
    if (t >= ngl * 8 + count * 3) return;
    if (!((which >> (t < ngl * 8 + count ? 0 : 1)) & 1)) return;
    const uint8_t* src = nullptr;
    bool neg = false;
    uint32_t idx;
    if (t < ngl * 8) {
        uint32_t k = t / ngl, g = t % ngl, p = g / nq, i = 2 * (g % nq) + ((k & 3) >> 1);
        if (V.st[p] == ZK_OK && !(V.okflags[p] & 8) && i < n) {
            const uint8_t* gk = v_gk_base(V, proofs + off[first + p], p);
            // cl_i, -cd_i | ca_i, cb_i
            src = gk + 72 * ((k < 4 ? (k & 1 ? 3 * n : 0) : (k & 1 ? 2 * n : n)) + i), neg = k < 4 && (k & 1);
        }
        idx = k * ngk + g;
    } else {
        uint32_t u = t - ngl * 8, k = u / count, p = u % count;
        bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8);
        if (good && (k == 0 || V.exp_st[p] == ZK_OK)) src = proofs + off[first + p] + (k == 2 ? 232 : 160);
        idx = k * nm + p;
    }
    Ft2 x, y, dt;
    term_point(src, neg, x, y, dt);
    vt_st_at(t < ngl * 8 ? V.gk_terms.pts : V.misc_terms.pts, idx, x, y, dt);
}
// Scalars of the Exp relations' shared points: one thread per proof sums its slots' parts (k_v_slot_terms).
__global__ void __launch_bounds__(64, 2) k_v_proof_sums(Workspace W, VWork V, uint32_t count) {
    uint32_t p = gtid();
    if (p >= count) return;
    for (uint32_t i = p; i < 2 * (MSM_G_MAX + 1); i += count) V.slot_cnt[i] = 0;   // k_v_slot_perm's counters of stage 2, per range
    uint32_t n = V.n, nm = V.C;
    bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8);
    Sq zero = fe_zero<ModQ>();
    {
        Sq eg = zero, eh = zero, ekx = zero, eky = zero;
        Sn SR = fe_zero<ModN>(), SH = fe_zero<ModN>(), SL = fe_zero<ModN>();
#pragma unroll 1
        for (uint32_t j = 0; j < VK; j++) {
            uint32_t sl = p * VK + j;
            eg = addq(eg, soa_ld<ModQ, 1>(V.sSg, sl)), eh = addq(eh, soa_ld<ModQ, 1>(V.sSh, sl));
            ekx = addq(ekx, soa_ld<ModQ, 1>(V.sSkx, sl)), eky = addq(eky, soa_ld<ModQ, 1>(V.sSky, sl));
            SR = fe_add_mod(SR, soa_ld<ModN, 1>(V.sSR, sl)), SH = fe_add_mod(SH, soa_ld<ModN, 1>(V.sSH, sl)), SL = fe_add_mod(SL, soa_ld<ModN, 1>(V.sSL, sl));
        }
        // fixed-base parts: list C slots p*4n + {0: membership, 1: Exp}
        soa_st(W.lc.v, p * 4 * n + 1, eg), soa_st(W.lc.r, p * 4 * n + 1, eh);
        soa_st(V.pSR, p, SR), soa_st(V.pSH, p, SH), soa_st(V.pSL, p, SL);
        bool e = good && V.exp_st[p] == ZK_OK;
        soa_st(V.misc_terms.sc, 1 * nm + p, e ? ekx : zero);
        soa_st(V.misc_terms.sc, 2 * nm + p, e ? eky : zero);
    }
}
// Scalars of the membership proof's terms: one thread per proof.  Needs the membership challenge and total (k_v_challenges, k_v_gk_*), nothing of verifyExp.
__global__ void __launch_bounds__(64, 2) k_v_proof_terms(Workspace W, VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, const uint8_t* vseeds, uint64_t first) {
    uint32_t p = gtid();
    if (p >= count) return;
    uint32_t n = V.n, nq = (n + 1) / 2, ngk = V.C * nq, nm = V.C;
    bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8);
    Sq zero = fe_zero<ModQ>();
    Sq mg = zero, mh = zero;    // membership g, h coefficients
    if (good) {
        const uint8_t* pr = proofs + off[first + p];
        const uint8_t* sc = v_gk_base(V, pr, p) + 4 * 72 * n;
        uint64_t gp = first + p;
        Sq x = chalq(V.gkx + 3 * p);
        Sq rF, dummy;
        v_rho_pair(vseeds, gp, 0x10000u, rF, dummy);
        Sq xp = fe_zero<ModQ>();
        xp.l[0] = 1;  // x^i
#pragma unroll 1
        for (uint32_t i = 0; i < n; i++) {
            Sq r0, r1;
            v_rho_pair(vseeds, gp, 0x10001u + i, r0, r1);
            uint32_t grp = p * nq + (i >> 1), o = (i & 1) * 2;
            // rel0: x cl + ca - f g - za h ; rel1: (x - f) cl + cb - zb h
            {
                Sq f = ld_scalar_q(sc + 32 * i);
                soa_st(V.gk_terms.sc, (o + 0) * ngk + grp, addq(mulq(r0, x), mulq(r1, fe_sub_mod(x, f))));
                mg = addq(mg, mulq(r0, f));
            }
            soa_st(V.gk_terms.sc, (4 + o) * ngk + grp, r0);
            soa_st(V.gk_terms.sc, (5 + o) * ngk + grp, r1);
            // relFinal: -x^i cd_i
            soa_st(V.gk_terms.sc, (o + 1) * ngk + grp, mulq(rF, xp));
            mh = addq(mh, mulq(r0, ld_scalar_q(sc + 32 * (n + i))));
            mh = addq(mh, mulq(r1, ld_scalar_q(sc + 32 * (2 * n + i))));
            xp = mulq(xp, x);
        }
        if (n & 1)
            for (uint32_t k = 2; k < 8; k++)
                if (k != 4 && k != 5) soa_st(V.gk_terms.sc, k * ngk + p * nq + nq - 1, zero);
        mg = addq(mg, mulq(rF, soa_ld<ModQ, 1>(V.gk_total, p)));
        mh = addq(mh, mulq(rF, ld_scalar_q(sc + 32 * 3 * n)));
        mg = fe_sub_mod(zero, mg), mh = fe_sub_mod(zero, mh);
        soa_st(V.misc_terms.sc, 0 * nm + p, mulq(rF, xp));  // x^n com
    } else {
        for (uint32_t q = 0; q < nq; q++)
            for (uint32_t k = 0; k < 8; k++) soa_st(V.gk_terms.sc, k * ngk + p * nq + q, zero);
        soa_st(V.misc_terms.sc, 0 * nm + p, zero);
    }
    soa_st(W.lc.v, p * 4 * n, mg), soa_st(W.lc.r, p * 4 * n, mh);
}

// ------------------------------------------------------------------ windowed Straus over a group of terms
// Scalars are recoded into SIGNED 4-bit digits (65 windows for a 256-bit scalar, 33 for a 128-bit one, digits in
// [-7, 8]); per term a table {1P..8P} (extended coordinates with d'*T premultiplied) is built once.  The group's lane
// then runs the windows top down: 4 shared doublings + one table addition per term (9 modmuls); a negative digit
// negates X and d'T of the entry on load.  (Unsigned 3-bit windows needed 86 / 43 additions per term.)
#define VW_ENT 8
#define VW_NW256 65
#define VW_NW128 33
// table storage is AoS: entry e of term idx = 36 contiguous words (X, Y, d'T, Z) at tab[(idx*8 + e)*36], so that a lane's
// digit-dependent lookup is one contiguous 144-byte read
ZK_DEV void st_tab(const VTerms& L, uint32_t e, uint32_t idx, const TomPt& a) {
    uint4* q = (uint4*)(L.tab + ((size_t)idx * VW_ENT + e) * 36);
    Ft2 dt = a.t * fe_const<ModT, 1>(TOM_D1_M);
    uint32_t w[36];
#pragma unroll
    for (int l = 0; l < 9; l++) w[l] = a.x.l[l], w[9 + l] = a.y.l[l], w[18 + l] = dt.l[l], w[27 + l] = a.z.l[l];
#pragma unroll
    for (int i = 0; i < 9; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
// thread t: term k = t / ngroups of group g = t % ngroups, at index k * ng_stride + g (+ blockIdx.y * ystride: several lists in one launch)
__global__ void __launch_bounds__(256, 2) k_v_term_tables(VTerms L, uint32_t ngroups, uint32_t ng_stride, uint32_t nt, uint32_t ystride) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ngroups * nt) return;
    uint32_t idx = (t / ngroups) * ng_stride + t % ngroups + blockIdx.y * ystride;
    Sq sc = soa_ld<ModQ, 1>(L.sc, idx);
    uint32_t kw[8];
    words_from_limbs<8>(kw, sc.l);
    uint32_t carry = 0;
#pragma unroll 1
    for (uint32_t w = 0; w < VW_NW256; w++) {  // signed recoding: d in [-7, 8]
        uint32_t d = (kw[0] & 15) + carry;
        shr256<4>(kw);
        bool neg = d > 8;
        carry = neg ? 1 : 0;
        if (neg) d = 16 - d;
        L.dig[(size_t)w * L.cap + idx] = (uint8_t)(d | (neg ? 0x80u : 0u));
    }
    if (fe_is_zero(sc)) return;  // null term: every digit is 0 and its table is never used
    TomPt p;
    vt_ld_xy(L, idx, p.x, p.y);
    p.t = p.x * p.y, p.z = fe_one_mont<ModT>().as<2>();
    TomPt m = p;
    st_tab(L, 0, idx, m);
#pragma unroll 1
    for (uint32_t e = 1; e < VW_ENT; e++) {
        m = e == 1 ? tom_dbl(p) : tom_add(m, p);
        st_tab(L, e, idx, m);
    }
}
// addition with a table entry (X2, Y2, d'T2, Z2): 9 modmuls
template <int KX>
ZK_DEV TomPt tom_add_tab(const TomPt& p, const Fe<ModT, KX>& x2, const Ft2& y2, const Fe<ModT, KX>& dt2, const Ft2& z2) {
    auto A = p.x * x2;
    auto B = p.y * y2;
    auto C = p.t * dt2;
    auto D = p.z * z2;
    auto E = ((p.x + p.y) * (x2 + y2) - A) - B;
    auto F = D - C;
    auto G = D + C;
    auto H = B - A;
    TomPt r;
    r.x = E * F, r.y = G * H, r.t = E * H, r.z = F * G;
    return r;
}
// -v for a coordinate < 2t: 4t - v < 4t ... kept below 4t, which the products of tom_add_tab accept (2 * 4 <= kmax)
ZK_DEV Fe<ModT, 4> ft_neg_sel(const Ft2& v, bool neg) {
    Fe<ModT, 4> r;
#pragma unroll
    for (int i = 0; i < NLIMB; i++) r.l[i] = neg ? ModT::sub4[i] - v.l[i] : v.l[i];
    limbs_normalize(r.l);
    return r;
}
// Lane i works on group perm[i] (identity without perm).  Lanes below cnt[0] are full groups: terms [0, n256) have 256-bit
// scalars (windows 64..0), terms [n256, n256 + n128) 128-bit scalars (windows 32..0).  The remaining lanes (slots of
// one-bit repetitions or of rejected proofs) only own the last two 128-bit terms, so they start at window 32 and add
// two entries per window; sorting the slots keeps waves homogeneous.
// tsplit > 1: a group's terms are dealt round-robin to tsplit lanes, each with its own accumulator (own doublings), written to
// out[g * ostride + part]: a lane's chain of 65 windows x 36 additions is ~12 ms long on its own, so when only a few thousand
// slots are re-checked (one failing group of the batched check) four short chains finish in a third of the time.
// blockIdx.y: list y of several equally shaped ones, its terms and accumulators ystride entries further on (the three one-term sums of a proof).
__global__ void __launch_bounds__(256) k_v_straus(VTerms L, uint32_t ngroups, uint32_t ng_stride, uint32_t n256, uint32_t n128, Soa4 out,
                                                  const uint32_t* __restrict__ perm, const uint32_t* __restrict__ cnt, uint32_t tsplit, uint32_t ostride, uint32_t ystride) {
    uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= ngroups * tsplit) return;
    const uint32_t yo = blockIdx.y * ystride;
    const uint32_t gi = lane / tsplit, part = lane % tsplit;
    const uint32_t g = perm ? perm[gi] : gi;
    const bool full = !perm || gi < cnt[0];
    TomPt acc = tom_identity();
    const uint32_t nt = n256 + n128;
    const uint32_t klo = full ? part : nt - 2, kstep = full ? tsplit : 1;
    const bool idle = !full && part != 0;   // a light slot's two terms stay with part 0
#pragma unroll 1
    for (int w = idle ? -1 : (full && n256 > part ? VW_NW256 - 1 : VW_NW128 - 1); w >= 0; w--) {
        acc = tom_dbl(tom_dbl(tom_dbl(tom_dbl(acc))));
        uint32_t kmax = w >= VW_NW128 ? n256 : nt;
#pragma unroll 1
        for (uint32_t k = klo; k < kmax; k += kstep) {
            uint32_t idx = k * ng_stride + g + yo;
            uint32_t db = L.dig[(size_t)w * L.cap + idx];
            uint32_t d = db & 15;
            bool neg = (db & 0x80u) != 0;
            const uint4* q = (const uint4*)(L.tab + ((size_t)idx * VW_ENT + (d ? d - 1 : 0)) * 36);
            uint32_t tw[36];
#pragma unroll
            for (int i = 0; i < 9; i++) {
                uint4 v = q[i];
                tw[4 * i] = v.x, tw[4 * i + 1] = v.y, tw[4 * i + 2] = v.z, tw[4 * i + 3] = v.w;
            }
            Ft2 x2, y2, dt2, z2;
#pragma unroll
            for (int l = 0; l < 9; l++) x2.l[l] = tw[l], y2.l[l] = tw[9 + l], dt2.l[l] = tw[18 + l], z2.l[l] = tw[27 + l];
            TomPt s = tom_add_tab(acc, ft_neg_sel(x2, neg), y2, ft_neg_sel(dt2, neg), z2);
            bool on = d != 0;
            acc.x = fe_select(on, s.x, acc.x), acc.y = fe_select(on, s.y, acc.y);
            acc.t = fe_select(on, s.t, acc.t), acc.z = fe_select(on, s.z, acc.z);
        }
    }
    const uint32_t o = g * ostride + part + yo;
    soa_st(out.x, o, acc.x), soa_st(out.y, o, acc.y), soa_st(out.z, o, acc.z), soa_st(out.t, o, acc.t);
}
void launch_v_straus(hipStream_t s, const VTerms& L, uint32_t ngroups, uint32_t ng_stride, uint32_t n256, uint32_t n128, const Soa4& out,
                     const uint32_t* perm, const uint32_t* cnt, uint32_t tsplit, uint32_t ostride, uint32_t ny, uint32_t ystride) {
    if (!ngroups) return;
    const uint32_t nterms = ngroups * (n256 + n128);
    hipLaunchKernelGGL(k_v_term_tables, dim3((nterms + 255) / 256, ny), dim3(256), 0, s, L, ngroups, ng_stride, n256 + n128, ystride);
    if ((uint64_t)ngroups * tsplit * ny <= ZK_COOP_MAX_CHAINS && !zk_one_lane_chains()) {   // few chains: their length is the cost -- a cooperating wave per chain (k_coop.hip)
        launch_v_straus_co(s, L, ngroups, ng_stride, n256, n128, out, perm, cnt, tsplit, ostride, ny, ystride);
        return;
    }
    hipLaunchKernelGGL(k_v_straus, dim3((ngroups * tsplit + 255) / 256, ny), dim3(256), 0, s, L, ngroups, ng_stride, n256, n128, out, perm, cnt, tsplit, ostride, ystride);
}
// Sum of `width` consecutive accumulators per outer index, one wave each: src[o * width + q] -> dst[o * dstride], followed by `fill` identities
// (the consumer adds a fixed number of accumulators per proof).  Lane q adds entries q, q + NT, ... and the NT partial sums fold in log2 NT steps in LDS
// (NT = 256 for the 720 term accumulators of a proof's slots: 2 + 8 additions in a row).
template <uint32_t NT>
__global__ void __launch_bounds__(NT) k_v_acc_tree(Soa4 src, uint32_t width, Soa4 dst, uint32_t dstride, uint32_t fill) {
    __shared__ uint32_t sh[36][NT];
    const uint32_t o = blockIdx.x, q = threadIdx.x;
    TomPt acc = tom_identity();
#pragma unroll 1
    for (uint32_t k = q; k < width; k += NT) {
        const uint32_t e = o * width + k;
        TomPt a;
        a.x = soa_ld<ModT, 2>(src.x, e), a.y = soa_ld<ModT, 2>(src.y, e), a.z = soa_ld<ModT, 2>(src.z, e), a.t = soa_ld<ModT, 2>(src.t, e);
        acc = k == q ? a : tom_add(acc, a);
    }
#pragma unroll 1
    for (uint32_t half = NT / 2; half >= 1; half >>= 1) {
        if (q >= half && q < 2 * half) {
#pragma unroll
            for (int l = 0; l < 9; l++) sh[l][q] = acc.x.l[l], sh[9 + l][q] = acc.y.l[l], sh[18 + l][q] = acc.z.l[l], sh[27 + l][q] = acc.t.l[l];
        }
        __syncthreads();
        if (q < half) {
            TomPt b;
#pragma unroll
            for (int l = 0; l < 9; l++) b.x.l[l] = sh[l][q + half], b.y.l[l] = sh[9 + l][q + half], b.z.l[l] = sh[18 + l][q + half], b.t.l[l] = sh[27 + l][q + half];
            acc = tom_add(acc, b);
        }
        __syncthreads();
    }
    if (q <= fill) {
        const TomPt r = q == 0 ? acc : tom_identity();
        const uint32_t e = o * dstride + q;
        soa_st(dst.x, e, r.x), soa_st(dst.y, e, r.y), soa_st(dst.z, e, r.z), soa_st(dst.t, e, r.t);
    }
}
void launch_v_acc_tree(hipStream_t s, const Soa4& src, uint32_t nouter, uint32_t width, const Soa4& dst, uint32_t dstride, uint32_t fill) {
    if (!nouter) return;
    if (width > 128) hipLaunchKernelGGL(k_v_acc_tree<256>, dim3(nouter), dim3(256), 0, s, src, width, dst, dstride, fill);
    else hipLaunchKernelGGL(k_v_acc_tree<64>, dim3(nouter), dim3(64), 0, s, src, width, dst, dstride, fill);
}
// P-256: sum of rho_j * (-A_j) over the 20 checked repetitions, 5 terms per thread, 128-bit randomisers.  Signed 4-bit
// windows like the Tom side: every thread first recodes its scalars (33 digits in [-7, 8]) and builds {1A..8A} for its
// terms in global memory (projective, rtab.h entry format), then runs 33 windows of 4 doublings + 5 complete additions
// (the bit-serial version computed 128 doublings + 640 additions per thread).
#define VP_NW 33
#define VP_NW_CL 35   // SL * Clambda: SL is the sum of 20 128-bit values, < 2^133
// one thread per term: digits of its scalar and the multiples {1A..8A}.  Threads [0, count * VK): the A terms; [count * VK, count * (VK + 1)): Clambda of
// proof t - count * VK with the scalar SL (until round 4 a 136-step double-and-add inside k_v_final: 272 dependent additions where the window walk has 175)
__global__ void __launch_bounds__(256, 2) k_v_p256_tables(VWork V, uint32_t count) {
    uint32_t t = gtid();
    if (t >= count * (VK + 1)) return;
    const bool is_cl = t >= count * VK;
    const uint32_t idx = is_cl ? t - count * VK : t;
    const uint32_t cap = is_cl ? V.C : V.C * VK, nw = is_cl ? VP_NW_CL : VP_NW;
    uint8_t* dig = is_cl ? V.cl_dig : V.pa_dig;
    uint32_t kw[8];
    words_from_limbs<8>(kw, soa_ld<ModN, 1>(is_cl ? V.pSL : V.pa_sc, idx).l);
    uint32_t carry = 0;
#pragma unroll 1
    for (uint32_t w = 0; w < nw; w++) {
        uint32_t d = (kw[0] & 15) + carry;
        shr256<4>(kw);
        bool neg = d > 8;
        carry = neg ? 1 : 0;
        if (neg) d = 16 - d;
        dig[(size_t)w * cap + idx] = (uint8_t)(d | (neg ? 0x80u : 0u));
    }
    P256Aff a;
    a.x = soa_ld<ModQ, 2>(is_cl ? V.clx : V.pa_x, idx), a.y = soa_ld<ModQ, 2>(is_cl ? V.cly : V.pa_y, idx);
    P256Pt b = p256_from_affine(a), m = b;
    uint32_t* e = (is_cl ? V.cl_tab : V.pa_tab) + (size_t)idx * 8 * RTAB_ENTRY_WORDS;
    st_rtab(e, m);
    m = p256_dbl(b);
    st_rtab(e + RTAB_ENTRY_WORDS, m);
#pragma unroll 1
    for (uint32_t d = 2; d < 8; d++) {   // doubling and additions in separate regions: one live set at a time
        m = p256_add(m, b);
        st_rtab(e + d * RTAB_ENTRY_WORDS, m);
    }
}
// thread (p, q), q < parts: windowed sum over the A terms [q * per, q * per + per) of proof p; q == parts - 1: SL * Clambda.  per = 5 (five lanes per proof) for
// chunks, per = 1 (21 lanes per proof) for small batches, where the length of a lane's chain is all that counts.  Result in pacc[p * parts + q].
__global__ void __launch_bounds__(256) k_v_p256_straus(VWork V, uint32_t count, uint32_t per) {
    uint32_t t = gtid();
    const uint32_t parts = VK / per + 1;
    if (t >= count * parts) return;
    uint32_t p = t / parts, q = t % parts;
    const bool is_cl = q == parts - 1;
    const uint32_t cap = is_cl ? V.C : V.C * VK, nterms = is_cl ? 1 : per;
    const uint32_t i0 = is_cl ? p : p * VK + q * per;
    const uint8_t* dig = is_cl ? V.cl_dig : V.pa_dig;
    const uint32_t* tab = is_cl ? V.cl_tab : V.pa_tab;
    P256Pt acc = p256_identity();
#pragma unroll 1
    for (int w = (is_cl ? VP_NW_CL : VP_NW) - 1; w >= 0; w--) {
#pragma unroll 1
        for (int d = 0; d < 4; d++) acc = p256_dbl(acc);
#pragma unroll 1
        for (uint32_t k = 0; k < nterms; k++) {
            uint32_t idx = i0 + k;
            uint32_t db = dig[(size_t)w * cap + idx], d = db & 15;
            P256Pt e = ld_rtab(tab + ((size_t)idx * 8 + (d ? d - 1 : 0)) * RTAB_ENTRY_WORDS);
            e.y = fe_select((db & 0x80u) != 0, fq8_neg(e.y), e.y);
            P256Pt s = p256_add(acc, e);
            acc = p256_select(d != 0, s, acc);
        }
    }
    soa_st(V.pacc.x, t, acc.x), soa_st(V.pacc.y, t, acc.y), soa_st(V.pacc.z, t, acc.z);
}
// The P-256 relation of a proof: SR * R + SH * h_NIST + SL * Clambda + sum(-rho A) is the identity (weier.ts:117-119) -> p256_ok[p].  R through the
// proof's own window table, h_NIST through its comb, the other `parts` sums from k_v_p256_straus.  Its own kernel since round 4: for a handful of
// proofs it runs beside the Tom-256 sums on another stream.
__global__ void __launch_bounds__(64, 2) k_v_p256_total(DevParams P, Workspace W, VWork V, uint32_t count, uint32_t parts) {
    uint32_t p = gtid();
    if (p >= count) return;
    uint32_t ok = 0;
    if (V.st[p] == ZK_OK && !(V.okflags[p] & 8)) {
        uint32_t kw[8];
        words_from_limbs<8>(kw, soa_ld<ModN, 1>(V.pSR, p).l);
        P256Pt acc = p256_rtab_mul(W.rtab + (size_t)p * rtab_words(RTAB_VERIFY_BITS), kw, RTAB_VERIFY_BITS);
        words_from_limbs<8>(kw, soa_ld<ModN, 1>(V.pSH, p).l);
#pragma unroll 1
        for (int w = 0; w < PFIX_NWIN; w++) {
            uint32_t d = kw[0] & (PFIX_WIN_SIZE - 1);
            shr256<PFIX_WIN_BITS>(kw);
            const uint32_t* en = P.pfix_H + (size_t)PFIX_ENTRY_WORDS * (w * PFIX_WIN_SIZE + d);
            P256Aff a;
            for (int l = 0; l < 9; l++) a.x.l[l] = en[l], a.y.l[l] = en[9 + l];
            P256Pt s = p256_add_mixed(acc, a);
            acc = p256_select(d != 0, s, acc);
        }
#pragma unroll 1
        for (uint32_t q = 0; q < parts; q++) {
            P256Pt a;
            a.x = soa_ld<ModQ, 8>(V.pacc.x, p * parts + q), a.y = soa_ld<ModQ, 8>(V.pacc.y, p * parts + q), a.z = soa_ld<ModQ, 8>(V.pacc.z, p * parts + q);
            acc = p256_add(acc, a);
        }
        ok = fe_is_zero(fe_reduce(acc.x)) && fe_is_zero(fe_reduce(acc.z)) && !fe_is_zero(fe_reduce(acc.y));  // weier.ts:117-119
    }
    V.p256_ok[p] = ok;
}

// The same for a handful of proofs, EIGHT lanes each: four take a quarter of the 65 windows of R's table, four a quarter of h_NIST's comb, each adds its share
// of the `parts` partial sums of k_v_p256_straus, and the eight points meet through the wave's cross-lane moves: 17 + 3 + 3 additions in a row instead of
// 65 + 13 + 21.
// MODE 1 / 2: the two halves of it as kernels of their own for a small chunk (api_verify.hip stage2a) -- the table walks (1: SR * R + SH * h_NIST, kept at
// pacc[count * parts + p]) need nothing from k_v_p256_straus and run beside it, 2 adds the partial sums and gives the verdict: 263 + 50 us in a row instead of 263 + 194.
template <int MODE>
__global__ void __launch_bounds__(256) k_v_p256_total_wide(DevParams P, Workspace W, VWork V, uint32_t count, uint32_t parts) {
    const uint32_t tt = gtid();
    const bool live = tt < count * 8;
    const uint32_t p = live ? tt >> 3 : count - 1, sub = tt & 7;   // dead lanes of the last wave mirror the last proof: the cross-lane moves need every lane of a group
    const bool good = V.st[p] == ZK_OK && !(V.okflags[p] & 8);
    P256Pt acc = p256_identity();
    if (good) {
        uint32_t kw[8];
        if (MODE != 2) {
            if (sub < 4) {
                words_from_limbs<8>(kw, soa_ld<ModN, 1>(V.pSR, p).l);
                const uint32_t per = (rtab_nwin(RTAB_VERIFY_BITS) + 3) / 4;
                acc = p256_rtab_mul_range(W.rtab + (size_t)p * rtab_words(RTAB_VERIFY_BITS), kw, RTAB_VERIFY_BITS, sub * per, per);
            } else {
                words_from_limbs<8>(kw, soa_ld<ModN, 1>(V.pSH, p).l);
                constexpr uint32_t gper = (PFIX_NWIN + 3) / 4;
                acc = p256_fixed_mul_range(acc, P.pfix_H, kw, (sub - 4) * gper, gper);
            }
        }
        if (MODE != 1) {
#pragma unroll 1
            for (uint32_t q = sub; q < parts + (MODE == 2 ? 1 : 0); q += 8) {
                const uint32_t e = q < parts ? p * parts + q : count * parts + p;   // (MODE 2: the table walks' sum after the straus lanes')
                P256Pt a;
                a.x = soa_ld<ModQ, 8>(V.pacc.x, e), a.y = soa_ld<ModQ, 8>(V.pacc.y, e), a.z = soa_ld<ModQ, 8>(V.pacc.z, e);
                acc = p256_add(acc, a);
            }
        }
    }
    acc = p256_quad_sum(acc);
    acc = p256_add(acc, p256_shfl_xor(acc, 4));
    if (!live || sub) return;
    if (MODE == 1) {
        const uint32_t e = count * parts + p;
        soa_st(V.pacc.x, e, acc.x), soa_st(V.pacc.y, e, acc.y), soa_st(V.pacc.z, e, acc.z);
        return;
    }
    V.p256_ok[p] = good && fe_is_zero(fe_reduce(acc.x)) && fe_is_zero(fe_reduce(acc.z)) && !fe_is_zero(fe_reduce(acc.y)) ? 1u : 0u;  // weier.ts:117-119
}

// ------------------------------------------------------------------ final sums and verdict
ZK_DEV TomPt ld_tom4(const Soa4& a, uint32_t e) {
    TomPt r;
    r.x = soa_ld<ModT, 2>(a.x, e), r.y = soa_ld<ModT, 2>(a.y, e), r.z = soa_ld<ModT, 2>(a.z, e), r.t = soa_ld<ModT, 2>(a.t, e);
    return r;
}
ZK_DEV bool tom_is_identity(const TomPt& a) {  // edwards.ts:117-125 on the a=1 image
    return fe_is_zero(a.x) && fe_eq(a.y, a.z) && !fe_is_zero(a.z);
}
// grp_ok[p / gsz] != 0: the batched check (k_msm.hip) found the Tom-256 total of the proof's group to be the identity, i.e. the
// membership and Exp/Tom sums of every proof of the group are (their per-proof accumulators were not computed)
__global__ void __launch_bounds__(64, 2) k_v_final(Workspace W, VWork V, uint32_t count, uint8_t* ok_out, int32_t* status_out, uint64_t first, VGroupFlags gf, uint32_t gsz) {
    uint32_t p = gtid();
    if (p >= count) return;
    const uint32_t flag = gf.v[p / gsz];          // 1: the group passed the batched check; else V_RECHECK | tsplit of its slot sums (| V_FOLDED: one sum per proof)
    const bool tom_all_ok = flag == 1, folded = flag & V_FOLDED;
    const uint32_t tsplit = flag & 0xff;
    int32_t st = V.st[p];  // structural errors; W.st additionally carries a late "T is at infinity"
    uint8_t ok = 0;
    if (V.st[p] == ZK_OK && !(V.okflags[p] & 8)) {
        uint32_t n = V.n, nq = (n + 1) / 2;
        // membership (gk.ts:261)
        bool memb = true;
        if (!tom_all_ok) {
            TomPt m = ld_tom_proj3(W.lc.proj, p * 4 * n);
            for (uint32_t q = 0; q < (folded ? 1 : nq); q++) m = tom_add(m, ld_tom4(V.gk_acc, p * nq + q));
            m = tom_add(m, ld_tom4(V.misc_acc, 0 * V.C + p));
            memb = tom_is_identity(m);
        }
        if (memb) {
            // exceptions of verifyExp only surface when membership passed (zkpAttestList.ts:165-183)
            int32_t est = v_exp_exception(V.exp_st[p], V.exp_jz[p], V.exp_jm[p]);
            if (est == ZK_OK && W.st[p] != ZK_OK) est = W.st[p];
            if (est != ZK_OK) st = est;
            else {
                bool okW = true;
                if (!tom_all_ok) {
                    TomPt e = ld_tom_proj3(W.lc.proj, p * 4 * n + 1);
                    for (uint32_t j = 0; j < (folded ? 1 : VK); j++)
                        for (uint32_t q = 0; q < tsplit; q++) e = tom_add(e, ld_tom4(V.slot_acc, (p * VK + j) * V_SLOT_SPLIT + q));
                    e = tom_add(e, ld_tom4(V.misc_acc, 1 * V.C + p));
                    e = tom_add(e, ld_tom4(V.misc_acc, 2 * V.C + p));
                    okW = tom_is_identity(e);
                }
                const bool okN = V.p256_ok[p] != 0;   // k_v_p256_total
                ok = (okW && okN) ? 1 : 0;
            }
        }
    }
    ok_out[first + p] = ok;
    status_out[first + p] = st;
}
// Clambda = comS1 from the proof (Montgomery affine), for k_v_final
__global__ void k_v_clambda(VWork V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    uint32_t p = gtid();
    if (p >= count) return;
    Fq2 x = fe_const<ModQ, 2>(P256_GX_M), y = fe_const<ModQ, 2>(P256_GY_M);
    if (V.st[p] == ZK_OK && !(V.okflags[p] & 8)) {
        const uint8_t* pr = proofs + off[first + p];
        uint32_t xw[8], yw[8];
        load_be32(pr + 96, xw);
        load_be32(pr + 128, yw);
        x = fe_to_mont(fe_from_words256_reduce<ModQ>(xw)), y = fe_to_mont(fe_from_words256_reduce<ModQ>(yw));
    }
    soa_st(V.clx, p, x), soa_st(V.cly, p, y);
}

// ------------------------------------------------------------------ launch wrappers
#define L1(kern, n, bs, ...) hipLaunchKernelGGL(kern, dim3(((n) + (bs)-1) / (bs)), dim3(bs), 0, s, __VA_ARGS__)
void launch_v_header_validate(hipStream_t s, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    L1(k_v_header, count, 256, V, count, proofs, off, first);
    {
        uint32_t per = (2 + V.sec) + 2 + 4 * V.n + 2 * V.sec + 32 * V.sec;   // = v_validate_threads(V.sec, V.n)
        hipLaunchKernelGGL(k_v_validate, dim3(count * ((per + 255) / 256)), dim3(256), 0, s, V, count, proofs, off, first);
    }
}
void launch_v_front_r(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    L1(k_v_front_r, count, 64, W, V, count, proofs, off, first);
}
void launch_v_front_q(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, const uint8_t* msg, uint64_t first) {
    L1(k_v_front_q, count, 64, P, W, V, count, proofs, off, msg, first);
    L1(k_v_clambda, count, 64, V, count, proofs, off, first);
}
void launch_v_challenges(hipStream_t s, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, const uint8_t* msg, uint64_t first, uint32_t parts) {
    L1(k_v_challenges, count, 64, V, count, proofs, off, msg, first, parts);
}
void launch_v_exp_challenge_small(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    L1(k_v_exph_msg, count * (2 + 3 * V.sec + 1), 256, W, V, count, proofs, off, first);
    launch_exph_hash(s, W, count, V.chal);
}
void launch_v_sample(hipStream_t s, const VWork& V, uint32_t count, const uint8_t* vseeds, uint64_t first) {   // needs the header only
    L1(k_v_sample_fills, count * VS_KMAX, 256, V, count, vseeds, first);
    L1(k_v_sample, count, 64, V, count, vseeds, first);
}
void launch_v_sample_check(hipStream_t s, const VWork& V, uint32_t count) {   // needs the recomputed challenge and the sampled slots
    L1(k_v_sample_check, count, 64, V, count);
}
void launch_v_exp_points(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first, uint32_t split) {
    if (split == 4 && (uint64_t)count * VK * 4 <= ZK_COOP_MAX_CHAINS && !zk_one_lane_chains()) launch_v_exp_points_co(s, W, V, count, proofs, off, first);   // k_coop.hip
    else L1(k_v_exp_points, count * VK * split, 256, W, V, count, proofs, off, first, split);
}
void launch_v_exp_status(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, bool have_jm) {   // needs the points, and k_v_sample_check's verdict if have_jm
    L1(k_v_exp_status, count, 64, W, V, count, have_jm);
}
void launch_v_t1_scalars(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    L1(k_v_t1_scalars, count * VK, 256, W, V, count, proofs, off, first);   // (t1_cnt: zeroed by k_v_exp_status)
}
void launch_v_derived(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    L1(k_v_derived, count * VK * 5, 256, W, V, count, proofs, off, first);
}
void launch_v_padd_hash(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {
    if (count <= V_PH_MAXP && !zk_one_lane_chains()) {
        L1(k_v_padd_msg, count * VK * 6 * 10, 256, P, W, V, count, proofs, off, first);
        Workspace Wb = W;
        Wb.exph_msg = V.ph_msg, Wb.exph_wk = V.ph_wk;
        launch_sha_msgs(s, Wb, count * VK * 6, V.vc, V_PH_BLOCKS, 3, V.ph_nblk);
        return;
    }
    L1(k_v_padd_hash, count * VK * 6, 256, P, W, V, count, proofs, off, first);
}
// The term lists of a chunk in four pieces (a large chunk runs them in this order on one stream, `which` = 3 in one launch; a small chunk runs the membership
// piece beside the P-256 front end and the slots' points beside the commitments T1 -- api_verify.hip stage1).
void launch_v_slot_points(hipStream_t s, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first) {   // needs idx, exp_st
    L1(k_v_slot_points, count * VK * V_SLOT_TERMS, 256, V, count, proofs, off, first);
}
void launch_v_slot_terms(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, const uint8_t* vseeds, uint64_t first) {
    L1(k_v_slot_terms, count * VK, 64, W, V, count, proofs, off, vseeds, first);   // needs T, the derived commitments' challenges (k_v_padd_hash)
}
void launch_v_proof_points(hipStream_t s, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, uint64_t first, uint32_t which) {
    L1(k_v_proof_points, count * ((V.n + 1) / 2 * 8 + 3), 256, V, count, proofs, off, first, which);
}
void launch_v_proof_terms(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, const uint8_t* proofs, const uint64_t* off, const uint8_t* vseeds, uint64_t first) {
    L1(k_v_proof_terms, count, 64, W, V, count, proofs, off, vseeds, first);   // needs the membership total
}
void launch_v_proof_sums(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count) {   // needs k_v_slot_terms
    L1(k_v_proof_sums, count, 64, W, V, count);
}
void launch_v_p256_straus(hipStream_t s, const VWork& V, uint32_t count, uint32_t per) {
    L1(k_v_p256_tables, count * (VK + 1), 256, V, count);
    if (per == 1 && (uint64_t)count * (VK + 1) <= ZK_COOP_MAX_CHAINS && !zk_one_lane_chains()) launch_v_p256_straus_co(s, V, count);   // k_coop.hip
    else L1(k_v_p256_straus, count * (VK / per + 1), 256, V, count, per);
}
void launch_v_p256_total(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count, uint32_t per) {
    if (count * 8 <= ZK_WIDE_MAX_UNITS) L1(k_v_p256_total_wide<0>, count * 8, 256, P, W, V, count, VK / per + 1);   // a handful of proofs: eight lanes each
    else L1(k_v_p256_total, count, 64, P, W, V, count, VK / per + 1);
}
// a small chunk (<= V_SIDE_MAXP proofs, one term per lane in k_v_p256_straus): the table walks, and the sum of everything
void launch_v_p256_total_fixed(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count) {
    L1(k_v_p256_total_wide<1>, count * 8, 256, P, W, V, count, VK + 1);
}
void launch_v_p256_total_sum(hipStream_t s, const DevParams& P, const Workspace& W, const VWork& V, uint32_t count) {
    L1(k_v_p256_total_wide<2>, count * 8, 256, P, W, V, count, VK + 1);
}
void launch_v_final(hipStream_t s, const Workspace& W, const VWork& V, uint32_t count, uint8_t* ok, int32_t* status, uint64_t first, const VGroupFlags& gf, uint32_t gsz) {
    L1(k_v_final, count, 64, W, V, count, ok, status, first, gf, gsz);
}
