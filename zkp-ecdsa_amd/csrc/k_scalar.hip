// Scalar-side kernels of the prover: commitment openings (v, r) for every Pedersen commitment, Sigma-protocol
// responses, the Groth-Kohlweiss polynomial (fold form), stream compaction of the zero-bit reps and ZKA1 output.
//
// Draw indices follow the reference's consumption order of crypto.getRandomValues inside ONE proveSignatureList
// call:
//   0 comS1.r (mod n) | 1 pkX.r | 2 pkY.r | 3+4i.. alpha_i, r_i (mod n), Tx_i.r, Ty_i.r |
//   then for the rho-th zero challenge bit, d0 = 3 + 4 sec + 40 rho:
//     d0+0 T1x.r, +1 T1y.r, +2 C8.r, +3 C10.r, +4 C11.r, +5 C13.r,
//     pi8: +6 kx, +7 ky, +8 kz, +9 sx, +10 sy, +11 sz, +12 s4 | pi10: +13.. | pi11: +20.. |
//     pix: +27 k, +28 s1, +29 s2 | pi13: +30.. | piy: +37 k, +38 s1, +39 s2
//   then g0 = 3 + 4 sec + 40 z:  g0+5i.. r_i, a_i, s_i, t_i, rho_i   (gk.ts:117-123)
#include "engine.h"

typedef Fe<ModQ, 1> Sq;  // canonical plain scalar mod q
typedef Fe<ModN, 1> Sn;

ZK_DEV Sq drawq(const Workspace& W, uint32_t p, uint32_t k) { return rng_draw<ModQ>(W.rng, p, k); }
ZK_DEV Sn drawn(const Workspace& W, uint32_t p, uint32_t k) { return rng_draw<ModN>(W.rng, p, k); }
ZK_DEV void put_vr(const TomList& L, uint32_t slot, const Sq& v, const Sq& r) {
    soa_st(L.v, slot, v);
    soa_st(L.r, slot, r);
}
ZK_DEV Sq chal_scalar(const uint32_t* c3) {
    uint32_t w[8] = {c3[0], c3[1], c3[2], 0, 0, 0, 0, 0};
    Sq r;
    limbs_from_words<8>(r.l, w);
    return r;
}

// ---------------------------------------------------------------- list A openings (zkpAttestList.ts:139-140, exp.ts:154-155)
__global__ void __launch_bounds__(256) k_lista_scalars(Workspace W, uint32_t count) {
    uint32_t t = gtid();
    uint32_t per = 2 + 2 * W.sec;
    if (t >= count * per) return;
    uint32_t p = t / per, k = t % per;
    Sq v, r;
    if (k == 0) v = soa_ld<ModQ, 1>(W.pkx, p), r = drawq(W, p, 1);
    else if (k == 1) v = soa_ld<ModQ, 1>(W.pky, p), r = drawq(W, p, 2);
    else {
        uint32_t i = (k - 2) >> 1, e = p * (W.sec + 1) + i;
        if ((k & 1) == 0) v = soa_ld<ModQ, 1>(W.Tx, e), r = drawq(W, p, 3 + 4 * i + 2);
        else v = soa_ld<ModQ, 1>(W.Ty, e), r = drawq(W, p, 3 + 4 * i + 3);
    }
    put_vr(W.la, t, v, r);
}
void launch_lista_scalars(hipStream_t s, const Workspace& W, uint32_t count) {
    uint32_t n = count * (2 + 2 * W.sec);
    hipLaunchKernelGGL(k_lista_scalars, dim3((n + 255) / 256), dim3(256), 0, s, W, count);
}

// ---------------------------------------------------------------- compaction: sizes, offsets, item list
ZK_DEV uint32_t count_zero_bits(const uint32_t* chal, uint32_t sec) { return zeros_below(chal, sec); }

// Single workgroup.  256 threads, not 1024: the host WAITS for this kernel, and a 16-wave workgroup needs sixteen free wave
// slots on one CU at the same moment -- under the other lane's register-heavy commitment kernels it was seen to starve for
// 60 ms (tools/exp_io_timeline.py), leaving its lane idle; four waves slip in as soon as any wave retires.
#define SCAN_T 256
__global__ void __launch_bounds__(SCAN_T) k_scan(Workspace W, uint32_t count, uint64_t cur_in, uint64_t out_cap, uint64_t* out_off, int32_t* status_out,
                                               uint32_t* totals /* [0] items, [1] overflow flag, [2..3] bytes */, uint64_t first_proof) {
    __shared__ uint64_t sb[SCAN_T];
    __shared__ uint32_t si[SCAN_T];
    __shared__ uint64_t s_cur;
    uint32_t t = threadIdx.x;
    uint32_t per = (count + SCAN_T - 1) / SCAN_T;
    uint32_t lo = t * per, hi = lo + per < count ? lo + per : count;
    uint64_t bytes = 0;
    uint32_t items = 0;
    for (uint32_t p = lo; p < hi; p++) {
        int32_t st = W.st[p];
        uint32_t z = count_zero_bits(W.chal + 4 * p, W.sec);
        if (st == ZK_OK || st == ZK_ST_T_INF_LATE) {
            // a stream that is too short reads as zero fills, which can surface as T[i] = identity first: the caller error wins
            uint32_t need = 3 + 4 * W.sec + 40 * (st == ZK_OK ? z : 0) + 5 * W.n + W.rng.exc_cnt[p];
            if (W.rng.exc_cnt[p] > RNG_MAX_EXC || (W.rng.mode == 1 && need > W.rng.stride_blocks)) st = ZK_E_RNG_EXHAUSTED;
            else if (st == ZK_ST_T_INF_LATE) st = ZK_E_T_INF;
            W.st[p] = st;
        }
        // r = 0 mod n (k_front): the first zero-bit repetition throws "Points don't add up!" (pointAdd.ts:104-106) -- after the commit phase's
        // 'T[i] is at infinity' (above), before proveMembership's `which` (ZK_E_ARG)
        if (W.r_zero[p] && (st == ZK_OK || st == ZK_E_ARG) && z > 0) W.st[p] = st = ZK_E_POINTS_DONT_ADD;
        if (st != ZK_OK) z = 0;
        W.zcnt[p] = z;
        items += z;
        bytes += st == ZK_OK ? wire_proof_size(W.wire, W.sec, W.n, z) : 0;
    }
    sb[t] = bytes, si[t] = items;
    __syncthreads();
    if (t == 0) {
        uint64_t b = 0;
        uint32_t it = 0;
        for (int i = 0; i < SCAN_T; i++) {
            uint64_t nb = sb[i];
            uint32_t ni = si[i];
            sb[i] = b, si[i] = it;
            b += nb, it += ni;
        }
        uint64_t cur = cur_in;
        totals[0] = it;
        totals[1] = (cur + b > out_cap) ? 1u : 0u;
        totals[2] = (uint32_t)b, totals[3] = (uint32_t)(b >> 32);
        W.item_base[count] = it;
        W.out_base[count] = b;
        out_off[first_proof + count] = cur + b;
        s_cur = cur;
    }
    __syncthreads();
    uint64_t b = sb[t];
    uint32_t it = si[t];
    uint64_t base = s_cur;
    for (uint32_t p = lo; p < hi; p++) {
        W.item_base[p] = it;
        W.out_base[p] = b;
        out_off[first_proof + p] = base + b;
        status_out[first_proof + p] = W.st[p];
        it += W.zcnt[p];
        b += W.st[p] == ZK_OK ? wire_proof_size(W.wire, W.sec, W.n, W.zcnt[p]) : 0;
    }
}
// Small read-backs the host waits for (chunk totals, slice boundaries, the batched check's verdicts) are WRITTEN by a kernel
// into page-locked, device-mapped host memory instead of being copied: an asynchronous D2H copy is served by the same DMA
// engine as the proof bytes of the other lanes and queues behind them -- a 16-byte read-back was seen to wait 60 ms for five
// 700 MB slices (tools/exp_io_timeline.py).  The host only synchronises with the stream.
__global__ void k_words_to_host(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint32_t n) {
    uint32_t i = gtid();
    if (i < n) dst[i] = src[i];
}
void launch_words_to_host(hipStream_t s, void* dst_pinned, const void* src_dev, size_t nwords) {
    if (!nwords) return;
    hipLaunchKernelGGL(k_words_to_host, dim3((uint32_t)((nwords + 255) / 256)), dim3(256), 0, s, (uint32_t*)dst_pinned, (const uint32_t*)src_dev, (uint32_t)nwords);
}
void launch_scan(hipStream_t s, const Workspace& W, uint32_t count, uint64_t cursor, uint64_t out_cap, uint64_t* d_out_off, int32_t* d_status_out,
                 uint32_t* d_totals, uint64_t first_proof) {
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(SCAN_T), 0, s, W, count, cursor, out_cap, d_out_off, d_status_out, d_totals, first_proof);
}
__global__ void __launch_bounds__(256) k_items(Workspace W, uint32_t count) {
    uint32_t t = gtid();
    if (t >= count * W.sec) return;
    uint32_t p = t / W.sec, i = t % W.sec;
    if (W.st[p] != ZK_OK) return;
    const uint32_t* c = W.chal + 4 * p;
    if ((c[i >> 5] >> (i & 31)) & 1) return;
    uint32_t rank = zeros_below(c, i);
    uint32_t it = W.item_base[p] + rank;
    W.item_proof[it] = p, W.item_rep[it] = i, W.item_rank[it] = rank;
}
void launch_items(hipStream_t s, const Workspace& W, uint32_t count) {
    uint32_t n = count * W.sec;
    hipLaunchKernelGGL(k_items, dim3((n + 255) / 256), dim3(256), 0, s, W, count);
}
// copy the chunk's final per-proof status (late, cryptographically negligible errors included)
__global__ void k_status_out(Workspace W, uint32_t count, int32_t* status_out, uint64_t first_proof) {
    uint32_t p = gtid();
    if (p < count) status_out[first_proof + p] = W.st[p];
}
void launch_status_out(hipStream_t s, const Workspace& W, uint32_t count, int32_t* d_status_out, uint64_t first_proof) {
    hipLaunchKernelGGL(k_status_out, dim3((count + 255) / 256), dim3(256), 0, s, W, count, d_status_out, first_proof);
}

// ---------------------------------------------------------------- PointAdd witness (pointAdd.ts:107-136)
// openings of one proveMult instance (mult.ts:102-114): 6 commitments starting at `slot`
ZK_DEV void mult_openings(const Workspace& W, uint32_t p, uint32_t dm, uint32_t it, uint32_t k0, const Sq& x, const Sq& y, const Sq& ry) {
    Sq kx = drawq(W, p, dm), ky = drawq(W, p, dm + 1), kz = drawq(W, p, dm + 2);
    Sq sx = drawq(W, p, dm + 3), sy = drawq(W, p, dm + 4), sz = drawq(W, p, dm + 5), s4 = drawq(W, p, dm + 6);
    auto xm = fe_to_mont(x), kxm = fe_to_mont(kx);
    put_vr(W.lb, lbi(W, it, k0 + 0), fe_canon(xm * y), fe_canon(xm * ry));    // C4 = x * Cy
    put_vr(W.lb, lbi(W, it, k0 + 1), kx, sx);                                 // Ax
    put_vr(W.lb, lbi(W, it, k0 + 2), ky, sy);                                 // Ay
    put_vr(W.lb, lbi(W, it, k0 + 3), kz, sz);                                 // Az
    put_vr(W.lb, lbi(W, it, k0 + 4), kz, s4);                                 // A4_1
    put_vr(W.lb, lbi(W, it, k0 + 5), fe_canon(kxm * y), fe_canon(kxm * ry));  // A4_2 = kx * Cy
}
// i8 = 1 / (x2 - x1) (pointAdd.ts:131, invMod) for all items of the chunk with Montgomery's trick: k_padd_i7 parks
// x2 - x1 (Montgomery form) in T1proj.x -- free once T1 is normalised --, k_padd_inv runs one Fermat inversion per
// `per` items (prefix products in T1proj.z) and leaves the inverses in T1proj.y.  x2 = x1 gives 0 like fe_inv(0).
__global__ void __launch_bounds__(256) k_padd_i7(Workspace W, uint32_t items) {
    uint32_t it = gtid();
    if (it >= items) return;
    Sq d = fe_sub_mod(soa_ld<ModQ, 1>(W.pkx, W.item_proof[it]), soa_ld<ModQ, 1>(W.T1x, it));
    soa_st(W.T1proj.x, it, fe_to_mont(d));
}
__global__ void __launch_bounds__(256) k_padd_inv(Workspace W, uint32_t items, uint32_t nthreads, uint32_t per) {
    uint32_t t = gtid();
    if (t >= nthreads) return;
    Fq2 acc = fe_one_mont<ModQ>().as<2>();
    for (uint32_t j = 0; j < per; j++) {
        uint32_t e = t + j * nthreads;
        if (e >= items) break;
        Fq2 v = soa_ld<ModQ, 2>(W.T1proj.x, e);
        soa_st(W.T1proj.z, e, acc);
        if (!fe_is_zero(v)) acc = acc * v;
    }
    Fq2 inv = fe_inv<ModQ>(acc);
    for (int j = (int)per - 1; j >= 0; j--) {
        uint32_t e = t + (uint32_t)j * nthreads;
        if (e >= items) continue;
        Fq2 v = soa_ld<ModQ, 2>(W.T1proj.x, e);
        bool zero = fe_is_zero(v);
        Fq2 r = inv * soa_ld<ModQ, 2>(W.T1proj.z, e);
        if (!zero) inv = inv * v;
        soa_st(W.T1proj.y, e, zero ? fe_zero<ModQ>().as<2>() : r);
    }
}
// One thread per (part, item), part-major so that a wave runs one part: 0 = the six commitments T1x, T1y, C8, C10, C11, C13;
// 1..4 = openings of pi8, pi10, pi11, pi13; 5 = pix, piy.  Each part derives the witness values it needs from x1, y1, x2, y2, x3 and
// i8 itself (at most two extra products), so no part holds more than a few scalars.
#define PADD_PARTS 6
// The six parts of an item read overlapping draws (the item's blinders d0 + 0..5, the repetition's and the proof's) and their neighbours in the same cache
// lines.  ZK_PADD_TILED = 1 (default since round 4): a workgroup is six waves over the SAME 64 items, wave w = part w, so those lines are fetched once per
// CU instead of once per part of a grid-wide sweep (part = t / items: the parts of an item ran a whole grid apart).  0 = the old mapping, for A/Bs.
#ifndef ZK_PADD_TILED
#define ZK_PADD_TILED 1
#endif
#ifndef ZK_PADD_RESPOND_WAVES
#define ZK_PADD_RESPOND_WAVES 3   // waves per SIMD k_padd_respond is compiled for (168 VGPRs)
#endif
#if ZK_PADD_TILED
#define PADD_BLOCK (64 * PADD_PARTS)
#define PADD_GRID(items) dim3(((items) + 63) / 64)
#define PADD_MAP(items, part, it)                                                                   \
    const uint32_t part = threadIdx.x >> 6, it = blockIdx.x * 64 + (threadIdx.x & 63);             \
    if (it >= (items)) return
#else
#define PADD_BLOCK 256
#define PADD_GRID(items) dim3(((items) * PADD_PARTS + 255) / 256)
#define PADD_MAP(items, part, it)                                                                   \
    const uint32_t t_ = gtid();                                                                     \
    if (t_ >= (items) * PADD_PARTS) return;                                                         \
    const uint32_t part = t_ / (items), it = t_ % (items)
#endif
struct PaddIn {
    uint32_t p, i, d0;
    Sq x1, y1;
};
ZK_DEV PaddIn padd_in(const Workspace& W, uint32_t it) {
    PaddIn a;
    a.p = W.item_proof[it], a.i = W.item_rep[it];
    a.d0 = 3 + 4 * W.sec + 40 * W.item_rank[it];
    a.x1 = soa_ld<ModQ, 1>(W.T1x, it), a.y1 = soa_ld<ModQ, 1>(W.T1y, it);
    return a;
}
ZK_DEV Sq padd_i9(const Workspace& W, const PaddIn& a) { return fe_sub_mod(soa_ld<ModQ, 1>(W.pky, a.p), a.y1); }
ZK_DEV Sq padd_i12(const Workspace& W, const PaddIn& a) { return fe_sub_mod(a.x1, soa_ld<ModQ, 1>(W.Tx, a.p * (W.sec + 1) + a.i)); }
#ifndef ZK_PADD_SCALARS_WAVES
#define ZK_PADD_SCALARS_WAVES 4   // waves per SIMD k_padd_scalars is compiled for
#endif
__global__ void __launch_bounds__(PADD_BLOCK, ZK_PADD_SCALARS_WAVES) k_padd_scalars(Workspace W, uint32_t items) {
    PADD_MAP(items, part, it);
    PaddIn a = padd_in(W, it);
    uint32_t p = a.p, d0 = a.d0;
    Sq i8 = fe_from_mont(soa_ld<ModQ, 2>(W.T1proj.y, it));
    switch (part) {
        case 0: {
            put_vr(W.lb, lbi(W, it, 0), a.x1, drawq(W, p, d0 + 0));       // T1x   (exp.ts:196)
            put_vr(W.lb, lbi(W, it, 1), a.y1, drawq(W, p, d0 + 1));       // T1y
            put_vr(W.lb, lbi(W, it, 2), i8, drawq(W, p, d0 + 2));         // C8    (pointAdd.ts:138-143)
            Sq i10 = fe_mul_mod(i8, padd_i9(W, a));
            put_vr(W.lb, lbi(W, it, 3), i10, drawq(W, p, d0 + 3));        // C10
            put_vr(W.lb, lbi(W, it, 4), fe_mul_mod(i10, i10), drawq(W, p, d0 + 4));                // C11
            put_vr(W.lb, lbi(W, it, 5), fe_mul_mod(i10, padd_i12(W, a)), drawq(W, p, d0 + 5));     // C13
        } break;
        case 1:  // pi8 : x = i7, Cy = C8
            mult_openings(W, p, d0 + 6, it, 6, fe_sub_mod(soa_ld<ModQ, 1>(W.pkx, p), a.x1), i8, drawq(W, p, d0 + 2));
            break;
        case 2:  // pi10: x = i8, Cy = C9 = C5 - C4
            mult_openings(W, p, d0 + 13, it, 12, i8, padd_i9(W, a), fe_sub_mod(drawq(W, p, 2), drawq(W, p, d0 + 1)));
            break;
        case 3: {  // pi11: x = i10, Cy = C10
            Sq i10 = fe_mul_mod(i8, padd_i9(W, a));
            mult_openings(W, p, d0 + 20, it, 18, i10, i10, drawq(W, p, d0 + 3));
        } break;
        case 4: {  // pi13: x = i10, Cy = C12 = C1 - C3
            Sq i10 = fe_mul_mod(i8, padd_i9(W, a));
            mult_openings(W, p, d0 + 30, it, 24, i10, padd_i12(W, a), fe_sub_mod(drawq(W, p, d0 + 0), drawq(W, p, 3 + 4 * a.i + 2)));
        } break;
        default: {
            Sq k = drawq(W, p, d0 + 27);                                               // pix (equality.ts:66-68)
            put_vr(W.lb, lbi(W, it, 30), k, drawq(W, p, d0 + 28));
            put_vr(W.lb, lbi(W, it, 31), k, drawq(W, p, d0 + 29));
            k = drawq(W, p, d0 + 37);                                                  // piy
            put_vr(W.lb, lbi(W, it, 32), k, drawq(W, p, d0 + 38));
            put_vr(W.lb, lbi(W, it, 33), k, drawq(W, p, d0 + 39));
        }
    }
}
void launch_padd_scalars(hipStream_t s, const DevParams&, const Workspace& W, uint32_t items) {
    if (!items) return;
    hipLaunchKernelGGL(k_padd_i7, dim3((items + 255) / 256), dim3(256), 0, s, W, items);
    uint32_t per = 16, nthreads = (items + per - 1) / per;
    hipLaunchKernelGGL(k_padd_inv, dim3((nthreads + 255) / 256), dim3(256), 0, s, W, items, nthreads, per);
    hipLaunchKernelGGL(k_padd_scalars, PADD_GRID(items), dim3(PADD_BLOCK), 0, s, W, items);
}

// ---------------------------------------------------------------- PointAdd responses (mult.ts:122-130, equality.ts:73-77)
ZK_DEV void mult_respond(const Workspace& W, uint32_t p, uint32_t dm, const uint32_t* c3, uint8_t* o, const Sq& x, const Sq& y, const Sq& z,
                         const Sq& rx, const Sq& ry, const Sq& rz) {
    auto cm = fe_to_mont(chal_scalar(c3));
    Sq kx = drawq(W, p, dm), ky = drawq(W, p, dm + 1), kz = drawq(W, p, dm + 2);
    Sq sx = drawq(W, p, dm + 3), sy = drawq(W, p, dm + 4), sz = drawq(W, p, dm + 5), s4 = drawq(W, p, dm + 6);
    Sq r4 = fe_mul_mod(x, ry);
    store_scalar_be(o + 0, fe_sub_mod(kx, fe_canon(cm * x)));
    store_scalar_be(o + 32, fe_sub_mod(ky, fe_canon(cm * y)));
    store_scalar_be(o + 64, fe_sub_mod(kz, fe_canon(cm * z)));
    store_scalar_be(o + 96, fe_sub_mod(sx, fe_canon(cm * rx)));
    store_scalar_be(o + 128, fe_sub_mod(sy, fe_canon(cm * ry)));
    store_scalar_be(o + 160, fe_sub_mod(sz, fe_canon(cm * rz)));
    store_scalar_be(o + 192, fe_sub_mod(s4, fe_canon(cm * r4)));
}
ZK_DEV void eq_respond(const Workspace& W, uint32_t p, uint32_t de, const uint32_t* c3, uint8_t* o, const Sq& x, const Sq& rc1, const Sq& rc2) {
    auto cm = fe_to_mont(chal_scalar(c3));
    Sq k = drawq(W, p, de), s1 = drawq(W, p, de + 1), s2 = drawq(W, p, de + 2);
    store_scalar_be(o + 0, fe_sub_mod(k, fe_canon(cm * x)));
    store_scalar_be(o + 32, fe_sub_mod(s1, fe_canon(cm * rc1)));
    store_scalar_be(o + 64, fe_sub_mod(s2, fe_canon(cm * rc2)));
}
// Same split as k_padd_scalars: part 0 = the rep-level responses of a zero bit, 1..4 = pi8, pi10, pi11, pi13, 5 = pix, piy.
__global__ void __launch_bounds__(PADD_BLOCK, ZK_PADD_RESPOND_WAVES) k_padd_respond(Workspace W, uint32_t items, uint8_t* out) {
    PADD_MAP(items, part, it);
    uint32_t p = W.item_proof[it], i = W.item_rep[it];
    uint32_t d0 = 3 + 4 * W.sec + 40 * W.item_rank[it];
    const Wire& wr = W.wire;
    uint8_t* rep = out + W.out_base[p] + rep_offset_w(wr, W.chal + 4 * p, i);
    uint8_t* pa = rep + wr.rep_head;
    uint8_t* rsc = rep + 64 + 4 * wr.tc;   // the repetition's four response scalars (ZKA1: rep + 208)
    const uint32_t* c = W.padd_c + (size_t)it * 18;
    const uint32_t MS = 8 * wr.tc + 12 * wr.tc;  // scalars of MultProof m start behind C8..C13 (8 coordinates), m MultProofs and its own 6 points
    const uint32_t ES = 8 * wr.tc + 4 * wr.mult + 4 * wr.tc;   // scalars of EqualityProof e: + eq * e
    auto lbv = [&](uint32_t k) { return soa_ld<ModQ, 1>(W.lb.v, lbi(W, it, k)); };  // 0..5 = x1, y1, i8, i10, i11, i13
    // blinders: r1, r4, r8, r10, r11, r13 = draws d0 + 0..5; r2, r5 = draws 1, 2 (Px, Py); r3, r6 = draws of Tx, Ty of the rep
    switch (part) {
        case 0: {
            // rep-level response for a zero bit (exp.ts:186,221-225): z = alpha - s, z2 = r_i - Cs.r, r1 = T1x.r, r2 = T1y.r
            Sn alpha = drawn(W, p, 3 + 4 * i), ri = drawn(W, p, 3 + 4 * i + 1), r0 = drawn(W, p, 0);
            store_scalar_be(rsc, fe_sub_mod(alpha, soa_ld<ModN, 1>(W.s1, p)));
            store_scalar_be(rsc + 32, fe_sub_mod(ri, r0));
            store_scalar_be(rsc + 64, drawq(W, p, d0 + 0));
            store_scalar_be(rsc + 96, drawq(W, p, d0 + 1));
        } break;
        case 1: {  // pi8: (i7, i8, 1) with blinders (r2 - r1, r8, 0): C14 = g
            Sq one = fe_zero<ModQ>();
            one.l[0] = 1;
            mult_respond(W, p, d0 + 6, c + 0, pa + MS, fe_sub_mod(soa_ld<ModQ, 1>(W.pkx, p), lbv(0)), lbv(2), one,
                         fe_sub_mod(drawq(W, p, 1), drawq(W, p, d0 + 0)), drawq(W, p, d0 + 2), fe_zero<ModQ>());
        } break;
        case 2:  // pi10: (i8, i9, i10)
            mult_respond(W, p, d0 + 13, c + 3, pa + MS + wr.mult, lbv(2), fe_sub_mod(soa_ld<ModQ, 1>(W.pky, p), lbv(1)), lbv(3), drawq(W, p, d0 + 2),
                         fe_sub_mod(drawq(W, p, 2), drawq(W, p, d0 + 1)), drawq(W, p, d0 + 3));
            break;
        case 3: {  // pi11: (i10, i10, i11)
            Sq i10 = lbv(3), r10 = drawq(W, p, d0 + 3);
            mult_respond(W, p, d0 + 20, c + 6, pa + MS + 2 * wr.mult, i10, i10, lbv(4), r10, r10, drawq(W, p, d0 + 4));
        } break;
        case 4:  // pi13: (i10, i12, i13)
            mult_respond(W, p, d0 + 30, c + 9, pa + MS + 3 * wr.mult, lbv(3), fe_sub_mod(lbv(0), soa_ld<ModQ, 1>(W.Tx, p * (W.sec + 1) + i)), lbv(5),
                         drawq(W, p, d0 + 3), fe_sub_mod(drawq(W, p, d0 + 0), drawq(W, p, 3 + 4 * i + 2)), drawq(W, p, d0 + 5));
            break;
        default: {
            Sq r1 = drawq(W, p, d0 + 0), r4 = drawq(W, p, d0 + 1);
            // pix: Cint = C3 + C1 + C2; piy: Cint = C6 + C4
            eq_respond(W, p, d0 + 27, c + 12, pa + ES, lbv(4), drawq(W, p, d0 + 4), fe_add_mod(fe_add_mod(drawq(W, p, 3 + 4 * i + 2), r1), drawq(W, p, 1)));
            eq_respond(W, p, d0 + 37, c + 15, pa + ES + wr.eq, lbv(5), drawq(W, p, d0 + 5), fe_add_mod(drawq(W, p, 3 + 4 * i + 3), r4));
        }
    }
}
// Limit study (profiles/r04_ab_variants.txt (6)): -DZK_AB_SKIP_RESPOND=1 leaves the response / serialisation kernels out altogether -- the proofs are then
// garbage -- to measure what the whole family costs the OVERLAPPED step, i.e. the most any rewrite of these kernels could win.  Never set in a product build.
#ifndef ZK_AB_SKIP_RESPOND
#define ZK_AB_SKIP_RESPOND 0
#endif
void launch_padd_respond(hipStream_t s, const Workspace& W, uint32_t items, uint8_t* out) {
    if (!items || ZK_AB_SKIP_RESPOND) return;
    hipLaunchKernelGGL(k_padd_respond, PADD_GRID(items), dim3(PADD_BLOCK), 0, s, W, items, out);
}

// ---------------------------------------------------------------- ZKA1 fixed part and rep heads
ZK_DEV void put_tom_point(uint8_t* o, const TomList& L, uint32_t slot) {
    store_tomcoord_be(o, soa_ld<ModT, 1>(L.ax, slot));
    store_tomcoord_be(o + 36, soa_ld<ModT, 1>(L.ay, slot));
}
ZK_DEV void put_p256_point(uint8_t* o, const Soa& ax, const Soa& ay, uint32_t e) {
    store_scalar_be(o, soa_ld<ModQ, 1>(ax, e));
    store_scalar_be(o + 32, soa_ld<ModQ, 1>(ay, e));
}
// two consecutive Tom points of a list (slots a, b) at o (4-byte aligned): 2 x 72 bytes, or 132 packed
ZK_DEV void put_tom_pair(const Wire& wr, uint8_t* o, const TomList& L, uint32_t a, uint32_t b) {
    if (wr.tc == 36) {
        put_tom_point(o, L, a);
        put_tom_point(o + 72, L, b);
    } else {
        store_tom_pair_packed(o, soa_ld<ModT, 1>(L.ax, a), soa_ld<ModT, 1>(L.ay, a), soa_ld<ModT, 1>(L.ax, b), soa_ld<ModT, 1>(L.ay, b));
    }
}
__global__ void __launch_bounds__(256) k_write_fixed(Workspace W, uint32_t count, uint8_t* out) {
    uint32_t t = gtid();
    uint32_t per = W.sec + 1;
    if (t >= count * per) return;
    uint32_t p = t / per, j = t % per;
    if (W.st[p] != ZK_OK) return;
    const Wire& wr = W.wire;
    uint8_t* base = out + W.out_base[p];
    const uint32_t* c = W.chal + 4 * p;
    uint32_t la = p * (2 + 2 * W.sec), ea = p * per;
    if (j == W.sec) {
        uint32_t total = (uint32_t)(W.out_base[p + 1] - W.out_base[p]);
        uint32_t* h = (uint32_t*)base;
        h[0] = wr.magic;  // "ZKA1" / "ZK1P"
        h[1] = bswap32(total), h[2] = bswap32(W.sec), h[3] = bswap32(W.n);
        uint32_t bits[4] = {c[0], c[1], c[2], 0};
        for (uint32_t b = W.sec; b < 128; b++) bits[b >> 5] &= ~(1u << (b & 31));
        store_be<4>(base + 16, bits);
        put_p256_point(base + 32, W.Rx, W.Ry, p);
        put_p256_point(base + 96, W.Ax, W.Ay, ea + W.sec);  // comS1
        put_tom_pair(wr, base + 160, W.la, la + 0, la + 1);   // keyXcom, keyYcom
        return;
    }
    uint8_t* rep = base + rep_offset_w(wr, c, j);
    put_p256_point(rep, W.Ax, W.Ay, ea + j);
    put_tom_pair(wr, rep + 64, W.la, la + 2 + 2 * j, la + 3 + 2 * j);   // Tx_j, Ty_j
    if ((c[j >> 5] >> (j & 31)) & 1) {  // exp.ts:170-184: alpha, r, Tx.r, Ty.r
        uint8_t* rsc = rep + 64 + 4 * wr.tc;
        store_scalar_be(rsc, drawn(W, p, 3 + 4 * j));
        store_scalar_be(rsc + 32, drawn(W, p, 3 + 4 * j + 1));
        store_scalar_be(rsc + 64, drawq(W, p, 3 + 4 * j + 2));
        store_scalar_be(rsc + 96, drawq(W, p, 3 + 4 * j + 3));
    }
}
void launch_write_fixed(hipStream_t s, const Workspace& W, uint32_t count, uint8_t* out) {
    if (ZK_AB_SKIP_RESPOND) return;
    uint32_t n = count * (W.sec + 1);
    hipLaunchKernelGGL(k_write_fixed, dim3((n + 255) / 256), dim3(256), 0, s, W, count, out);
}
// The 32 Tom points of a PointAdd block (2 304 of its 3 392 bytes), staged through LDS: list B is item-fastest, the proof bytes are
// item-major.  A workgroup takes WP_ITEMS consecutive items; phase A reads the limbs coalesced (32 consecutive items per limb row) and
// parks the 18 big-endian words of each point in LDS, phase B walks an item's words in output order so that consecutive lanes write
// consecutive dwords of the runs of 4 / 6 / 2 points (round 1 wrote one 72-byte piece per lane, 3.4 KB apart: 1.8 TB/s).
#ifndef WP_ITEMS
#define WP_ITEMS 32
#endif
#define WP_WORDS (32 * 18)       // point words per item
#define WP_STRIDE (WP_WORDS + 1)  // LDS item stride: odd, so that the 32 items of a phase-A row fall into 32 banks
ZK_DEV uint32_t padd_point_off(const Wire& wr, uint32_t k) {  // byte offset of point k (0..31) inside the PointAdd block (pointAdd.ts:138-191 order)
    const uint32_t pt = 2 * wr.tc;
    return k < 4 ? pt * k : k < 28 ? 4 * pt + wr.mult * ((k - 4) / 6) + pt * ((k - 4) % 6) : k < 30 ? 4 * pt + 4 * wr.mult + pt * (k - 28) : 4 * pt + 4 * wr.mult + wr.eq + pt * (k - 30);
}
__global__ void __launch_bounds__(256) k_write_padd_points(Workspace W, uint32_t items, uint8_t* out) {
    __shared__ uint32_t words[WP_ITEMS * WP_STRIDE];
    __shared__ uint64_t base[WP_ITEMS];
    const Wire& wr = W.wire;
    const uint32_t t = threadIdx.x, it0 = blockIdx.x * WP_ITEMS;
    const uint32_t li = t % WP_ITEMS, kk = t / WP_ITEMS, it = it0 + li;
    if (it < items) {
        if (kk == 0) {
            uint32_t p = W.item_proof[it], i = W.item_rep[it];
            base[li] = W.out_base[p] + rep_offset_w(wr, W.chal + 4 * p, i) + wr.rep_head;
        }
#pragma unroll 1
        for (uint32_t k = kk; k < 32; k += 256 / WP_ITEMS) {
            uint32_t slot = lbi(W, it, 2 + k), w[9];
            uint32_t* d = words + li * WP_STRIDE + k * 18;
            words_from_limbs<9>(w, soa_ld<ModT, 1>(W.lb.ax, slot).l);
#pragma unroll
            for (int j = 0; j < 9; j++) d[j] = bswap32(w[8 - j]);
            words_from_limbs<9>(w, soa_ld<ModT, 1>(W.lb.ay, slot).l);
#pragma unroll
            for (int j = 0; j < 9; j++) d[9 + j] = bswap32(w[8 - j]);
        }
    }
    __syncthreads();
    const uint32_t nit = items - it0 < WP_ITEMS ? items - it0 : WP_ITEMS;
    if (wr.tc == 36) {
        const uint32_t n = nit * WP_WORDS;
        for (uint32_t e = t; e < n; e += 256) {
            uint32_t i = e / WP_WORDS, d = e % WP_WORDS, k = d / 18;
            *(uint32_t*)(out + base[i] + padd_point_off(wr, k) + 4 * (d % 18)) = words[i * WP_STRIDE + d];
        }
    } else {
        // ZKA1P: the 32 points of an item are 32 x 66 bytes; every run of points starts on a multiple of 132 of that string and at a 4-byte
        // aligned offset of the block, so output dword d holds bytes 4 d .. 4 d + 3 of the string and never straddles a run.  A coordinate's
        // 33 bytes are bytes 3 .. 35 of its 36-byte big-endian image in LDS.
        const uint8_t* lb8 = (const uint8_t*)words;
        const uint32_t PW = 32 * 66 / 4;   // 528 dwords per item
        const uint32_t n = nit * PW;
        for (uint32_t e = t; e < n; e += 256) {
            const uint32_t i = e / PW, q0 = 4 * (e % PW);
            uint32_t v = 0;
#pragma unroll
            for (uint32_t b = 0; b < 4; b++) {
                const uint32_t q = q0 + b, k = q / 66, r = q % 66;
                v |= (uint32_t)lb8[4 * (i * WP_STRIDE + k * 18 + (r >= 33 ? 9 : 0)) + 3 + (r >= 33 ? r - 33 : r)] << (8 * b);
            }
            const uint32_t k0 = q0 / 66;
            *(uint32_t*)(out + base[i] + padd_point_off(wr, k0) + q0 % 66) = v;
        }
    }
}
void launch_write_padd_points(hipStream_t s, const Workspace& W, uint32_t items, uint8_t* out) {
    if (!items || ZK_AB_SKIP_RESPOND) return;
    hipLaunchKernelGGL(k_write_padd_points, dim3((items + WP_ITEMS - 1) / WP_ITEMS), dim3(256), 0, s, W, items, out);
}

// ---------------------------------------------------------------- Groth-Kohlweiss (gk.ts:94-195)
// Polynomial: the reference evaluates d(w) = sum_i (v_l - v_i) p_i(w) at w = 0..n-1 (2 N n modmuls) and
// interpolates (interpolate.ts).  Since sum_i p_i(w) = w^n, d(w) = v_l w^n - P(w) with P(w) = sum_i v_i p_i(w), and
// P is obtained EXACTLY (same coefficients, deg d <= n-1) by folding the ring along the index bits with
// polynomial-valued entries:  new = w * (l_j ? odd : even) + a_j * (odd - even)   -- (j+1) modmuls per output,
// 2N in total.  The self-check of interpolate.ts:63-67 holds by construction.
ZK_DEV uint32_t gk_g0(const Workspace& W, uint32_t p) { return 3 + 4 * W.sec + 40 * W.zcnt[p]; }
__global__ void __launch_bounds__(256) k_gk_scalars(Workspace W, ChunkIn in, Soa am) {
    uint32_t t = gtid();
    if (t >= in.count * W.n) return;
    uint32_t p = t / W.n, j = t % W.n;
    uint32_t g0 = gk_g0(W, p) + 5 * j;
    uint32_t l = (in.which[p] >> j) & 1;
    Sq r = drawq(W, p, g0), a = drawq(W, p, g0 + 1), s = drawq(W, p, g0 + 2), tt = drawq(W, p, g0 + 3);
    Sq lv = fe_zero<ModQ>();
    lv.l[0] = l;
    uint32_t base = p * 4 * W.n;
    put_vr(W.lc, base + j, lv, r);                               // cl_j = Com(l_j; r_j)
    put_vr(W.lc, base + W.n + j, a, s);                          // ca_j = Com(a_j; s_j)
    put_vr(W.lc, base + 2 * W.n + j, l ? a : fe_zero<ModQ>(), tt);  // cb_j = Com(l_j a_j; t_j)
    soa_st(am, j * W.C + p, fe_to_mont(a));
}
// one fold level for a group of proofs.  in/out element index: (k * G + g) * npoly + m
__global__ void __launch_bounds__(256) k_gk_level(Workspace W, ChunkIn in, Soa am, uint32_t first, uint32_t G, uint32_t j, Soa src, Soa dst, uint32_t npoly_out) {
    uint32_t t = gtid();
    if (t >= G * npoly_out) return;
    uint32_t g = t / npoly_out, m = t % npoly_out, p = first + g;
    uint32_t npoly_in = 2 * npoly_out;
    bool l = (in.which[p] >> j) & 1;
    Fe<ModQ, 2> a = soa_ld<ModQ, 2>(am, j * W.C + p);
    Sq prev_sel = fe_zero<ModQ>();
    bool last = npoly_out == 1;
    for (uint32_t k = 0; k <= j; k++) {
        Sq ev, od;
        if (j == 0) ev = soa_ld<ModQ, 1>(W.ring, 2 * m), od = soa_ld<ModQ, 1>(W.ring, 2 * m + 1);
        else {
            uint32_t e = (k * G + g) * npoly_in + 2 * m;
            ev = soa_ld<ModQ, 1>(src, e), od = soa_ld<ModQ, 1>(src, e + 1);
        }
        Sq prod = fe_canon(a * fe_sub_mod(od, ev));
        Sq c = fe_add_mod(prod, prev_sel);
        prev_sel = l ? od : ev;
        if (last) soa_st(W.gk_coef, k * W.C + p, c);
        else soa_st(dst, (k * G + g) * npoly_out + m, c);
    }
    if (last) soa_st(W.gk_coef, (j + 1) * W.C + p, prev_sel);
    else soa_st(dst, ((j + 1) * G + g) * npoly_out + m, prev_sel);
}
__global__ void __launch_bounds__(256) k_gk_cd_scalars(Workspace W, uint32_t count) {
    uint32_t t = gtid();
    if (t >= count * W.n) return;
    uint32_t p = t / W.n, k = t % W.n;
    Sq c = soa_ld<ModQ, 1>(W.gk_coef, k * W.C + p);
    Sq d = fe_sub_mod(fe_zero<ModQ>(), c);                       // d_k = -P_k
    put_vr(W.lc, p * 4 * W.n + 3 * W.n + k, d, drawq(W, p, gk_g0(W, p) + 5 * k + 4));  // cd_k = Com(d_k; rho_k)
}
void launch_gk_cd_scalars(hipStream_t s, const Workspace& W, uint32_t count) {
    uint32_t n = count * W.n;
    hipLaunchKernelGGL(k_gk_cd_scalars, dim3((n + 255) / 256), dim3(256), 0, s, W, count);
}
// responses (gk.ts:181-194) and the GK section of the proof
__global__ void __launch_bounds__(64) k_gk_respond(Workspace W, ChunkIn in, uint8_t* out) {
    uint32_t p = gtid();
    if (p >= in.count || W.st[p] != ZK_OK) return;
    uint32_t n = W.n, g0 = gk_g0(W, p);
    uint8_t* gk = out + W.out_base[p] + W.wire.fixed + (uint64_t)W.wire.rep_head * W.sec + (uint64_t)W.wire.padd * W.zcnt[p];
    uint8_t* sc = gk + 8 * W.wire.tc * n;
    Sq x = chal_scalar(W.gk_x + 3 * p);
    auto xm = fe_to_mont(x);
    Sq rcom = drawq(W, p, 1);                                    // blinder of keyXcom
    Fe<ModQ, 2> xpow = fe_one_mont<ModQ>().as<2>();              // x^i (Montgomery)
    Sq acc = fe_zero<ModQ>();                                    // sum rho_i x^i
    for (uint32_t i = 0; i < n; i++) {
        uint32_t l = (in.which[p] >> i) & 1;
        Sq r = drawq(W, p, g0 + 5 * i), a = drawq(W, p, g0 + 5 * i + 1), s = drawq(W, p, g0 + 5 * i + 2);
        Sq tt = drawq(W, p, g0 + 5 * i + 3), rho = drawq(W, p, g0 + 5 * i + 4);
        Sq f = l ? fe_add_mod(x, a) : a;                         // f_i = l_i x + a_i
        Sq za = fe_add_mod(fe_canon(xm * r), s);                 // za_i = r_i x + s_i
        Sq zb = fe_add_mod(fe_mul_mod(r, fe_sub_mod(x, f)), tt); // zb_i = r_i (x - f_i) + t_i
        store_scalar_be(sc + 32 * i, f);
        store_scalar_be(sc + 32 * (n + i), za);
        store_scalar_be(sc + 32 * (2 * n + i), zb);
        acc = fe_add_mod(acc, fe_canon(xpow * rho));
        xpow = xpow * xm;
    }
    Sq zd = fe_sub_mod(fe_canon(xpow * rcom), acc);              // zd = r x^n - sum rho_i x^i
    store_scalar_be(sc + 32 * 3 * n, zd);
}
__global__ void __launch_bounds__(256) k_write_gk_points(Workspace W, uint32_t count, uint8_t* out) {   // one thread per PAIR of the 4 n points
    uint32_t t = gtid();
    if (t >= count * 2 * W.n) return;
    uint32_t p = t / (2 * W.n), k = t % (2 * W.n);
    if (W.st[p] != ZK_OK) return;
    uint8_t* gk = out + W.out_base[p] + W.wire.fixed + (uint64_t)W.wire.rep_head * W.sec + (uint64_t)W.wire.padd * W.zcnt[p];
    put_tom_pair(W.wire, gk + 4 * W.wire.tc * k, W.lc, p * 4 * W.n + 2 * k, p * 4 * W.n + 2 * k + 1);
}
void launch_gk_respond(hipStream_t s, const Workspace& W, const ChunkIn& in, uint8_t* out) {
    if (ZK_AB_SKIP_RESPOND) return;
    hipLaunchKernelGGL(k_gk_respond, dim3((in.count + 63) / 64), dim3(64), 0, s, W, in, out);
    uint32_t n = in.count * 2 * W.n;
    hipLaunchKernelGGL(k_write_gk_points, dim3((n + 255) / 256), dim3(256), 0, s, W, in.count, out);
}
// ---- plain fold (rings without table E, see k_gk.hip for the table path): a tile of 2^T ring elements per workgroup.
// RL levels run depth-first in registers (2^RL elements per lane), the others coefficient-parallel through LDS (one output coefficient = one modmul per lane), so a whole
// tile costs ~20 modmul latencies instead of one kernel launch per level.  LDS planes are limb-major (conflict-free).
#define GK_TMAX 12   // 256 lanes x 16 elements
struct LdsPlane {
    uint32_t* p;
    uint32_t stride;
};
typedef Fe<ModQ, 64> Sq64;  // lazily reduced fold coefficient: after level j the value is < 2(j+1) M (induction: c = prod(<2M) + sel)
ZK_DEV Sq64 lds_ld(const LdsPlane& a, uint32_t e) {
    Sq64 r;
#pragma unroll
    for (int l = 0; l < NLIMB; l++) r.l[l] = a.p[l * a.stride + e];
    return r;
}
template <int K>
ZK_DEV void lds_st(const LdsPlane& a, uint32_t e, const Fe<ModQ, K>& v) {
#pragma unroll
    for (int l = 0; l < NLIMB; l++) a.p[l * a.stride + e] = v.l[l];
}
template <int K>
ZK_DEV Sq64 as64(const Fe<ModQ, K>& v) {  // bound bookkeeping only (see Sq64)
    Sq64 r;
#pragma unroll
    for (int l = 0; l < NLIMB; l++) r.l[l] = v.l[l];
    return r;
}
// Depth-first register fold of 2^LEV consecutive ring elements: polynomial with LEV+1 coefficients, lazily reduced
// (coefficient bound K(LEV) = 2 LEV + 1 multiples of q: c = a*(od - ev) [< 2q] + sel [< K(LEV-1) q]).
// The index bits can be folded in any order (the p_i are products of per-bit linear factors): a lane's 2^LEV elements
// are `stride` apart, so lanes read consecutive ring elements (coalesced) and the register phase folds index bits
// bit0 .. bit0+LEV-1; the number of coefficients only depends on how many bits have been folded.
template <int LEV>
struct GkFold {
    static constexpr int K = 2 * LEV + 1;
    static ZK_DEV void run(const Soa& ring, const Soa& am, uint32_t C, uint32_t p, uint32_t which, uint32_t base, uint32_t stride, uint32_t bit0,
                           Fe<ModQ, K> (&out)[LEV + 1]) {
        Fe<ModQ, GkFold<LEV - 1>::K> ev[LEV], od[LEV];
        GkFold<LEV - 1>::run(ring, am, C, p, which, base, stride, bit0, ev);
        GkFold<LEV - 1>::run(ring, am, C, p, which, base + (stride << (LEV - 1)), stride, bit0, od);
        uint32_t jb = bit0 + LEV - 1;
        Fe<ModQ, 2> a = soa_ld<ModQ, 2>(am, jb * C + p);
        bool l = (which >> jb) & 1;
#pragma unroll
        for (int k = 0; k < LEV; k++) {
            Fe<ModQ, 2> prod = a * (od[k] - ev[k]);
            if (k == 0) out[0] = prod.template as<K>();
            else out[k] = (prod + fe_select(l, od[k - 1], ev[k - 1])).template as<K>();
        }
        out[LEV] = fe_select(l, od[LEV - 1], ev[LEV - 1]).template as<K>();
    }
};
template <>
struct GkFold<0> {
    static constexpr int K = 1;
    static ZK_DEV void run(const Soa& ring, const Soa&, uint32_t, uint32_t, uint32_t, uint32_t base, uint32_t, uint32_t, Fe<ModQ, 1> (&out)[1]) {
        out[0] = soa_ld<ModQ, 1>(ring, base);
    }
};
// levels j0..j1-1 of `npoly` polynomials (j0+1 coefs each) held in LDS plane A (element (k*npoly + m)); result left in
// the plane returned by reference (ping-pong with B).  All threads of the workgroup must call this.
// `deg0` = index bits already folded (polynomials have deg0+1 coefficients), the steps fold index bits bit0, bit0+1, ...
ZK_DEV void gk_lds_levels(LdsPlane& A, LdsPlane& B, uint32_t npoly, uint32_t deg0, uint32_t nsteps, uint32_t bit0, const Workspace& W, const Soa& am, uint32_t p, uint32_t which) {
    for (uint32_t st = 0; st < nsteps; st++) {
        uint32_t j = deg0 + st, jb = bit0 + st;
        uint32_t nout = npoly >> 1;
        Fe<ModQ, 2> a = soa_ld<ModQ, 2>(am, jb * W.C + p);
        bool l = (which >> jb) & 1;
        uint32_t items = nout * (j + 2);
        for (uint32_t it = threadIdx.x; it < items; it += blockDim.x) {
            uint32_t k = it / nout, m = it % nout;
            Sq64 c;
            if (k <= j) {
                Sq64 ev = lds_ld(A, k * npoly + 2 * m), od = lds_ld(A, k * npoly + 2 * m + 1);
                Fe<ModQ, 2> prod = a * (od - ev);
                if (k > 0) c = as64(prod + lds_ld(A, (k - 1) * npoly + 2 * m + (l ? 1 : 0)));
                else c = as64(prod);
            } else {
                c = lds_ld(A, j * npoly + 2 * m + (l ? 1 : 0));
            }
            lds_st(B, k * nout + m, c);
        }
        __syncthreads();
        LdsPlane t = A;
        A = B, B = t;
        npoly = nout;
    }
}
// grid = count * ntiles workgroups of 256 lanes; RL register levels (2^RL elements per lane), T >= RL levels per tile.
// Output: tile polynomial (T+1 coefs) at res[(k*C + p)*ntiles + tile], or straight into gk_coef when the tile is the ring.
template <int RL>
__global__ void __launch_bounds__(256) k_gk_tile(Workspace W, ChunkIn in, Soa am, uint32_t T, uint32_t ntiles, Soa res) {
    constexpr uint32_t A_EL = 256u * (RL + 1), B_EL = 128u * (RL + 2);
    __shared__ uint32_t ldsA[NLIMB * A_EL];
    __shared__ uint32_t ldsB[NLIMB * B_EL];
    uint32_t p = blockIdx.x / ntiles, tile = blockIdx.x % ntiles;
    uint32_t which = in.which[p];
    uint32_t lanes = 1u << (T - RL);  // active lanes in the register phase (<= 256)
    uint32_t t = threadIdx.x;
    LdsPlane A = {ldsA, A_EL}, B = {ldsB, B_EL};
    if (t < lanes) {
        Fe<ModQ, GkFold<RL>::K> poly[RL + 1];
        GkFold<RL>::run(W.ring, am, W.C, p, which, (tile << T) + t, lanes, T - RL, poly);
#pragma unroll
        for (int k = 0; k <= RL; k++) lds_st(A, k * lanes + t, poly[k]);
    }
    __syncthreads();
    gk_lds_levels(A, B, lanes, RL, T - RL, 0, W, am, p, which);  // adjacent lanes differ in index bit 0, then 1, ...
    bool whole = ntiles == 1;
    for (uint32_t k = t; k <= T; k += blockDim.x) {
        Sq c = fe_canon(fe_reduce(lds_ld(A, k)));
        if (whole) soa_st(W.gk_coef, k * W.C + p, c);
        else soa_st(res, (k * W.C + p) * ntiles + tile, c);
    }
}
// finish pass: workgroup (proof, group) folds `gsz` consecutive polynomials (Tin+1 coefs each, canonical) through
// log2(gsz) levels; the result goes to gk_coef when it is the whole ring, else to dst[(k*C + p)*ngroups + g].
__global__ void __launch_bounds__(256) k_gk_finish(Workspace W, ChunkIn in, Soa am, uint32_t Tin, uint32_t npoly, uint32_t gsz, Soa src, Soa dst) {
    // dynamic LDS: plane A holds gsz x (Tin+1) elements, plane B gsz/2 x (Tin+2); later levels need less
    extern __shared__ uint32_t lds_dyn[];
    const uint32_t a_el = gsz * (Tin + 1), b_el = (gsz / 2) * (Tin + 2);
    uint32_t* ldsA = lds_dyn;
    uint32_t* ldsB = lds_dyn + NLIMB * a_el;
    uint32_t ngroups = npoly / gsz;
    uint32_t p = blockIdx.x / ngroups, g = blockIdx.x % ngroups;
    uint32_t which = in.which[p];
    LdsPlane A = {ldsA, a_el}, B = {ldsB, b_el};
    for (uint32_t it = threadIdx.x; it < gsz * (Tin + 1); it += blockDim.x) {
        uint32_t k = it / gsz, m = it % gsz;
        lds_st(A, k * gsz + m, soa_ld<ModQ, 1>(src, (k * W.C + p) * npoly + g * gsz + m));
    }
    __syncthreads();
    uint32_t lv = 0;
    while ((1u << lv) < gsz) lv++;
    gk_lds_levels(A, B, gsz, Tin, lv, Tin, W, am, p, which);
    for (uint32_t k = threadIdx.x; k <= Tin + lv; k += blockDim.x) {
        Sq c = fe_canon(fe_reduce(lds_ld(A, k)));
        if (ngroups == 1) soa_st(W.gk_coef, k * W.C + p, c);
        else soa_st(dst, (k * W.C + p) * ngroups + g, c);
    }
}
// host-side driver of the fold
void launch_gk_scalars_fold(hipStream_t s, const Workspace& W, const ChunkIn& in, const Soa& am) {
    uint32_t nt = in.count * W.n;
    hipLaunchKernelGGL(k_gk_scalars, dim3((nt + 255) / 256), dim3(256), 0, s, W, in, am);
    if (W.n >= 3) {
        const uint32_t RL = W.n >= 4 ? 4 : 3;
        uint32_t T = W.gk_etab ? 8 : W.n < GK_TMAX ? W.n : GK_TMAX;
        uint32_t ntiles = W.N >> T;
        // tile results: (T+1) coefs x ntiles per proof, kept in gk_bufA (capacity checked by the workspace carver)
        Soa res = {W.gk_bufA, (uint32_t)((T + 1) * W.C * ntiles)};
        if (W.gk_etab) launch_gk_block_stage(s, W, in, am, res);  // 8 low index bits through the per-ring table (k_gk.hip)
        else if (RL == 4) hipLaunchKernelGGL(k_gk_tile<4>, dim3(in.count * ntiles), dim3(256), 0, s, W, in, am, T, ntiles, res);
        else hipLaunchKernelGGL(k_gk_tile<3>, dim3(in.count * ntiles), dim3(256), 0, s, W, in, am, T, ntiles, res);
        // finish passes: groups of <= 64 polynomials per workgroup (dynamic LDS sized to the group), ping-pong bufA/bufB
        Soa src = res;
        uint32_t* other = W.gk_bufB;
        while (ntiles > 1) {
            uint32_t gsz = gk_finish_gsz(T, ntiles), ngroups = ntiles / gsz;
            uint32_t lv = 0;
            while ((1u << lv) < gsz) lv++;
            Soa dst = {other, (uint32_t)((T + lv + 1) * W.C * ngroups)};
            size_t lds = gk_finish_lds(T, gsz);
            hipLaunchKernelGGL(k_gk_finish, dim3(in.count * ngroups), dim3(256), lds, s, W, in, am, T, ntiles, gsz, src, dst);
            other = src.p;
            src = dst, T += lv, ntiles = ngroups;
        }
        return;
    }
    // tiny rings (N < 8): one kernel per level through ping-pong buffers
    uint32_t cap = W.gk_group * W.N;
    Soa A = {W.gk_bufA, cap}, B = {W.gk_bufB, cap};
    for (uint32_t first = 0; first < in.count; first += W.gk_group) {
        uint32_t G = in.count - first < W.gk_group ? in.count - first : W.gk_group;
        Soa src = A, dst = B;
        for (uint32_t j = 0; j < W.n; j++) {
            uint32_t npoly_out = W.N >> (j + 1);
            uint32_t threads = G * npoly_out;
            hipLaunchKernelGGL(k_gk_level, dim3((threads + 255) / 256), dim3(256), 0, s, W, in, am, first, G, j, src, dst, npoly_out);
            Soa tmp = src;
            src = dst, dst = tmp;
        }
    }
}

// ---------------------------------------------------------------- misc conversions
__global__ void k_ring_load(const uint8_t* keys, uint64_t nkeys, uint64_t N, Soa ring) {
    uint64_t i = gtid();
    if (i >= N) return;
    uint32_t w[8];
    load_be32(keys + 32 * (i < nkeys ? i : 0), w);  // gk.ts:80-83 pads with element 0
    soa_st(ring, (uint32_t)i, fe_from_words256_reduce<ModQ>(w));
}
void launch_ring_load(hipStream_t s, const uint8_t* d_keys, uint64_t nkeys, uint64_t N, const Soa& ring) {
    hipLaunchKernelGGL(k_ring_load, dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, s, d_keys, nkeys, N, ring);
}
// keyToInt (zkpAttestList.ts:94-102): deserializePoint's curve check (weier.ts:74-89, mod p, no range check) + affine x
__global__ void k_keys_to_ints(const uint8_t* pk, uint64_t count, uint8_t* out, int32_t* st) {
    uint64_t i = gtid();
    if (i >= count) return;
    uint32_t xw[8], yw[8];
    load_be32(pk + 64 * i, xw);
    load_be32(pk + 64 * i + 32, yw);
    Sq x = fe_from_words256_reduce<ModQ>(xw);
    P256Aff a;
    a.x = fe_to_mont(x);
    a.y = fe_to_mont(fe_from_words256_reduce<ModQ>(yw));
    bool ok = p256_on_curve(a);
    st[i] = ok ? ZK_OK : ZK_E_POINT_NOT_IN_GROUP;
    store_scalar_be(out + 32 * i, ok ? x : fe_zero<ModQ>());
}
void launch_keys_to_ints(hipStream_t s, const uint8_t* d_pk, uint64_t count, uint8_t* d_out, int32_t* d_st) {
    hipLaunchKernelGGL(k_keys_to_ints, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, s, d_pk, count, d_out, d_st);
}
__global__ void k_bytes_to_scalars(const uint8_t* be, uint64_t count, Soa out) {
    uint64_t i = gtid();
    if (i >= count) return;
    uint32_t w[8];
    load_be32(be + 32 * i, w);
    soa_st(out, (uint32_t)i, fe_from_words256_reduce<ModQ>(w));
}
void launch_bytes_to_scalars(hipStream_t s, const uint8_t* d_be32, uint64_t count, const Soa& out) {
    hipLaunchKernelGGL(k_bytes_to_scalars, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, s, d_be32, count, out);
}
__global__ void k_affine_to_bytes(Soa ax, Soa ay, uint64_t count, int tom, uint8_t* out) {
    uint64_t i = gtid();
    if (i >= count) return;
    if (tom) {
        store_tomcoord_be(out + 72 * i, soa_ld<ModT, 1>(ax, (uint32_t)i));
        store_tomcoord_be(out + 72 * i + 36, soa_ld<ModT, 1>(ay, (uint32_t)i));
    } else {
        store_scalar_be(out + 64 * i, soa_ld<ModQ, 1>(ax, (uint32_t)i));
        store_scalar_be(out + 64 * i + 32, soa_ld<ModQ, 1>(ay, (uint32_t)i));
    }
}
void launch_affine_to_bytes(hipStream_t s, const Soa& ax, const Soa& ay, uint64_t count, int tom, uint8_t* d_out) {
    hipLaunchKernelGGL(k_affine_to_bytes, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, s, ax, ay, count, tom, d_out);
}

// ---------------------------------------------------------------- unit-test hook: field ops on 40-byte big-endian operands
template <class M>
ZK_DEV void test_field_one(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    uint32_t aw[10], bw[10];
    for (int i = 0; i < 10; i++) aw[i] = bswap32(((const uint32_t*)a)[9 - i]), bw[i] = bswap32(((const uint32_t*)b)[9 - i]);
    Fe<M, 1> x = fe_from_words<M, 9>(aw), y = fe_from_words<M, 9>(bw), r;
    if (op == 0) r = fe_mul_mod(x, y);
    else if (op == 1) r = fe_add_mod(x, y);
    else if (op == 2) r = fe_sub_mod(x, y);
    else if (op == 4) {  // x*y - x - y through the fused double subtraction (the E term of the Edwards addition)
        auto xm = fe_to_mont(x), ym = fe_to_mont(y);
        r = fe_from_mont(fe_sub2(xm * ym, xm, ym));
    } else if (op == 5) {  // (x + y)^2 through the dedicated squaring, at a lazy magnitude
        r = fe_from_mont(fe_sqr(fe_to_mont(x) + fe_to_mont(y)));
    } else r = fe_from_mont(fe_inv<M>(fe_to_mont(x)));
    uint32_t rw[10];
    words_from_limbs<9>(rw, r.l);
    rw[9] = 0;
    for (int i = 0; i < 10; i++) ((uint32_t*)out)[i] = bswap32(rw[9 - i]);
}
__global__ void k_test_field(int which, int op, uint64_t count, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    uint64_t i = gtid();
    if (i >= count) return;
    if (which == 0) test_field_one<ModQ>(op, a + 40 * i, b + 40 * i, out + 40 * i);
    else if (which == 1) test_field_one<ModN>(op, a + 40 * i, b + 40 * i, out + 40 * i);
    else test_field_one<ModT>(op, a + 40 * i, b + 40 * i, out + 40 * i);
}
void launch_test_field(hipStream_t s, int which, int op, uint64_t count, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    hipLaunchKernelGGL(k_test_field, dim3((uint32_t)((count + 63) / 64)), dim3(64), 0, s, which, op, count, a, b, out);
}
