// Fixed-base table construction (one-time, at zk_ctx_set_params).  The reference rebuilds a 16-entry window table
// inside every mul/dblmul call (src/curves/group.ts:105-112,139-143); bases g, h (Tom-256) and G, h_NIST (P-256)
// are fixed for a whole batch, so the engine precomputes d * 2^(W w) * P for every W-bit window w
// and digit d once and every commitment becomes 2 * ceil(256/W) table additions with no doublings.
#include "engine.h"

// ---------------------------------------------------------------- Tom-256 fixed bases
// Table of one base P for a W-bit comb: entry[w][d] = d * 2^(W w) * P in affine niels form.  Built by composition so
// that even the 24-bit tables (184 M entries per base) take a fraction of a second and no per-entry scratch:
//   1. k_tomtab_bases   2^(W w) P for every window (one thread, 256 doublings)
//   2. k_tomtab_sub     two half-width sub-tables per window, extended coordinates:
//                       Lo[w][i] = i * B_w (i < 2^lo), Hi[w][j] = j * 2^lo * B_w (j < 2^hi), lo = ceil(W/2), hi = W - lo
//   3. k_tomtab_compose entry[w][d] = Lo[w][d mod 2^lo] + Hi[w][d >> lo]: one addition, then to affine with
//                       Montgomery's trick over 8 entries per thread (prefix products parked in the entries
//                       themselves, the sums recomputed on the way back: 2 x 9 + 3 + 4 modmuls + 1/8 inversion per entry).
// scratch layout (words): window bases [nwin][36], then per window Lo [2^lo][36] and Hi [2^hi][36]
// digits run over [0, E), E = tom_win_entries(bits) (2^bits, or 2^(bits-1) + 1 for the signed widths): split d = i + j * 2^lo
static inline uint32_t tom_lo_bits(uint32_t bits) { return ((tom_signed(bits) ? bits - 1 : bits) + 1) / 2; }
static inline uint32_t tom_n_hi(uint32_t bits) { return ((tom_win_entries(bits) - 1) >> tom_lo_bits(bits)) + 1; }
size_t tom_table_scratch_words(uint32_t bits) {
    return (size_t)tom_nwin(bits) * 36 * (1 + ((size_t)1 << tom_lo_bits(bits)) + tom_n_hi(bits));
}
size_t pfix_table_scratch_words() { return (size_t)PFIX_NWIN * 36 + (size_t)PFIX_NWIN * PFIX_WIN_SIZE * 36; }

ZK_DEV void st_tompt(uint32_t* p, const TomPt& a) {
#pragma unroll
    for (int l = 0; l < 9; l++) p[l] = a.x.l[l], p[9 + l] = a.y.l[l], p[18 + l] = a.t.l[l], p[27 + l] = a.z.l[l];
}
ZK_DEV TomPt ld_tompt(const uint32_t* p) {
    TomPt a;
#pragma unroll
    for (int l = 0; l < 9; l++) a.x.l[l] = p[l], a.y.l[l] = p[9 + l], a.t.l[l] = p[18 + l], a.z.l[l] = p[27 + l];
    return a;
}
__global__ void __launch_bounds__(64) k_tomtab_bases(const uint32_t* xy, uint32_t* scratch, int32_t* ok, uint32_t bits, uint32_t nwin) {
    if (gtid() != 0) return;
    uint32_t xw[9], yw[9];
    for (int i = 0; i < 9; i++) xw[i] = xy[i], yw[i] = xy[9 + i];
    TomPt p;
    bool good = tom_from_affine_words(p, xw, yw);
    if (!good) *ok = 0;
    for (uint32_t w = 0; w < nwin; w++) {
        st_tompt(scratch + 36 * w, p);
        for (uint32_t i = 0; i < bits; i++) p = tom_dbl(p);
    }
}
__global__ void __launch_bounds__(64) k_tomtab_sub(uint32_t* scratch, uint32_t nwin, uint32_t lo, uint32_t n_hi) {
    uint32_t t = gtid();
    uint32_t per_win = (1u << lo) + n_hi;
    uint32_t hi = 32 - __clz(n_hi > 1 ? n_hi - 1 : 1);  // bits of the largest Hi index
    if (t >= nwin * per_win) return;
    uint32_t w = t / per_win, i = t % per_win;
    bool is_hi = i >= (1u << lo);
    uint32_t d = is_hi ? i - (1u << lo) : i;
    TomPt base = ld_tompt(scratch + 36 * w);
    TomPt acc = tom_identity();
    for (int b = (int)(is_hi ? hi : lo) - 1; b >= 0; b--) {
        acc = tom_dbl(acc);
        TomPt s = tom_add(acc, base);
        bool bit = (d >> b) & 1;
        acc.x = fe_select(bit, s.x, acc.x), acc.y = fe_select(bit, s.y, acc.y);
        acc.t = fe_select(bit, s.t, acc.t), acc.z = fe_select(bit, s.z, acc.z);
    }
    if (is_hi)
        for (uint32_t b = 0; b < lo; b++) acc = tom_dbl(acc);
    st_tompt(scratch + (size_t)36 * nwin + (size_t)36 * t, acc);
}
#define TOMTAB_PER 8
__global__ void __launch_bounds__(256) k_tomtab_compose(const uint32_t* scratch, uint32_t* tab, uint32_t ent, uint32_t nwin, uint32_t lo, uint32_t n_hi, uint32_t nthreads) {
    uint32_t t = gtid();
    if (t >= nthreads) return;
    const uint32_t per_win = (1u << lo) + n_hi;
    const uint32_t* sub = scratch + (size_t)36 * nwin;
    const uint64_t total = (uint64_t)nwin * ent;
    auto entry_sum = [&](uint64_t e) {
        uint32_t w = (uint32_t)(e / ent), d = (uint32_t)(e % ent);
        const uint32_t* sw = sub + (size_t)36 * per_win * w;
        return tom_add(ld_tompt(sw + (size_t)36 * (d & ((1u << lo) - 1))), ld_tompt(sw + (size_t)36 * ((1u << lo) + (d >> lo))));
    };
    Ft2 acc = fe_one_mont<ModT>().as<2>();
    for (int j = 0; j < TOMTAB_PER; j++) {
        uint64_t e = t + (uint64_t)j * nthreads;  // consecutive lanes own consecutive entries: coalesced 128-byte stores
        if (e >= total) break;
        uint32_t* pe = tab + (size_t)TOM_ENTRY_WORDS * e;  // the prefix product is parked in the entry it belongs to
#pragma unroll
        for (int l = 0; l < 9; l++) pe[l] = acc.l[l];
        acc = acc * entry_sum(e).z;
    }
    Ft2 inv = fe_inv<ModT>(acc);
    const auto d1 = fe_const<ModT, 1>(TOM_D1_M);
    for (int j = TOMTAB_PER - 1; j >= 0; j--) {
        uint64_t e = t + (uint64_t)j * nthreads;
        if (e >= total) continue;
        TomPt s = entry_sum(e);
        Ft2 pre;
#pragma unroll
        for (int l = 0; l < 9; l++) pre.l[l] = tab[(size_t)TOM_ENTRY_WORDS * e + l];
        Ft2 zi = inv * pre;
        inv = inv * s.z;
        Ft2 x = s.x * zi, y = s.y * zi;
        Ft2 dt = (x * y) * d1;
        uint4* q = (uint4*)(tab + (size_t)TOM_ENTRY_WORDS * e);
        uint32_t w[28];
#pragma unroll
        for (int l = 0; l < 9; l++) w[l] = x.l[l], w[9 + l] = y.l[l], w[18 + l] = dt.l[l];
        w[27] = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
    }
}
void launch_build_tom_table(hipStream_t s, const uint32_t* xy, uint32_t bits, uint32_t* tab, uint32_t* scratch, int32_t* ok) {
    uint32_t nwin = tom_nwin(bits), lo = tom_lo_bits(bits), n_hi = tom_n_hi(bits), ent = tom_win_entries(bits);
    hipLaunchKernelGGL(k_tomtab_bases, dim3(1), dim3(64), 0, s, xy, scratch, ok, bits, nwin);
    uint32_t nsub = nwin * ((1u << lo) + n_hi);
    hipLaunchKernelGGL(k_tomtab_sub, dim3((nsub + 63) / 64), dim3(64), 0, s, scratch, nwin, lo, n_hi);
    uint64_t total = (uint64_t)nwin * ent;
    uint32_t nthreads = (uint32_t)((total + TOMTAB_PER - 1) / TOMTAB_PER);
    hipLaunchKernelGGL(k_tomtab_compose, dim3((nthreads + 255) / 256), dim3(256), 0, s, scratch, tab, ent, nwin, lo, n_hi, nthreads);
}

// ---------------------------------------------------------------- P-256 fixed bases
ZK_DEV void st_ppt(uint32_t* p, const P256Pt& a) {
#pragma unroll
    for (int l = 0; l < 9; l++) p[l] = a.x.l[l], p[9 + l] = a.y.l[l], p[18 + l] = a.z.l[l];
}
ZK_DEV P256Pt ld_ppt(const uint32_t* p) {
    P256Pt a;
#pragma unroll
    for (int l = 0; l < 9; l++) a.x.l[l] = p[l], a.y.l[l] = p[9 + l], a.z.l[l] = p[18 + l];
    return a;
}
__global__ void k_pfix_bases(const uint32_t* xy, uint32_t* scratch, int32_t* ok) {
    if (gtid() != 0) return;
    P256Aff a;
    if (xy) {
        uint32_t xw[8], yw[8];
        for (int i = 0; i < 8; i++) xw[i] = xy[i], yw[i] = xy[8 + i];
        a.x = fe_to_mont(fe_from_words256_reduce<ModQ>(xw));
        a.y = fe_to_mont(fe_from_words256_reduce<ModQ>(yw));
        if (!p256_on_curve(a)) *ok = 0;
    } else {
        a.x = fe_const<ModQ, 2>(P256_GX_M);
        a.y = fe_const<ModQ, 2>(P256_GY_M);
    }
    P256Pt p = p256_from_affine(a);
    for (int w = 0; w < PFIX_NWIN; w++) {
        st_ppt(scratch + 36 * w, p);
        for (int i = 0; i < PFIX_WIN_BITS; i++) p = p256_dbl(p);
    }
}
__global__ void __launch_bounds__(64) k_pfix_fill(uint32_t* scratch) {
    uint32_t t = gtid();
    if (t >= PFIX_NWIN * PFIX_WIN_SIZE) return;
    uint32_t w = t >> PFIX_WIN_BITS, d = t & (PFIX_WIN_SIZE - 1);
    P256Pt base = ld_ppt(scratch + 36 * w);
    P256Pt acc = p256_identity();
    for (int b = PFIX_WIN_BITS - 1; b >= 0; b--) {
        acc = p256_dbl(acc);
        P256Pt s = p256_add(acc, base);
        acc = p256_select((d >> b) & 1, s, acc);
    }
    st_ppt(scratch + PFIX_NWIN * 36 + (size_t)36 * t, acc);
}
__global__ void k_pfix_affine(const uint32_t* scratch, uint32_t* tab) {
    uint32_t t = gtid();
    if (t >= PFIX_NWIN * PFIX_WIN_SIZE) return;
    P256Pt a = ld_ppt(scratch + PFIX_NWIN * 36 + (size_t)36 * t);
    Fq2 zi = fe_inv<ModQ>(fe_reduce(a.z));  // identity (digit 0) gives 0 -> entry (0,0), never used
    Fq2 x = a.x * zi, y = a.y * zi;
    uint32_t* e = tab + (size_t)PFIX_ENTRY_WORDS * t;
#pragma unroll
    for (int l = 0; l < 9; l++) e[l] = x.l[l], e[9 + l] = y.l[l];
    e[18] = 0, e[19] = 0;
}
void launch_build_pfix_table(hipStream_t s, const uint32_t* xy, uint32_t* tab, uint32_t* scratch, int32_t* ok) {
    hipLaunchKernelGGL(k_pfix_bases, dim3(1), dim3(64), 0, s, xy, scratch, ok);
    hipLaunchKernelGGL(k_pfix_fill, dim3((PFIX_NWIN * PFIX_WIN_SIZE + 63) / 64), dim3(64), 0, s, scratch);
    hipLaunchKernelGGL(k_pfix_affine, dim3((PFIX_NWIN * PFIX_WIN_SIZE + 63) / 64), dim3(64), 0, s, scratch, tab);
}
