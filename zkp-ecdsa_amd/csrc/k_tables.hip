// Fixed-base table construction (one-time, at zk_ctx_set_params).  The reference rebuilds a 16-entry window table
// inside every mul/dblmul call (src/curves/group.ts:105-112,139-143); bases g, h (Tom-256) and G, h_NIST (P-256)
// are fixed for a whole batch, so the engine precomputes d * 2^(W w) * P for every W-bit window w
// and digit d once (W = 16 by default) and every commitment becomes 2 * 256/W table additions with no doublings.
#include "engine.h"

// scratch layout (words): window bases [NWIN][36] extended/projective, then entries [NWIN * 2^W][36]
size_t table_scratch_words() { return (size_t)TOM_NWIN * 36 + (size_t)TOM_NWIN * TOM_WIN_SIZE * 36 + (size_t)PFIX_NWIN * 36 + (size_t)PFIX_NWIN * PFIX_WIN_SIZE * 36; }

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
__global__ void k_tomtab_bases(const uint32_t* xy, uint32_t* scratch, int32_t* ok) {
    if (gtid() != 0) return;
    uint32_t xw[9], yw[9];
    for (int i = 0; i < 9; i++) xw[i] = xy[i], yw[i] = xy[9 + i];
    TomPt p;
    bool good = tom_from_affine_words(p, xw, yw);
    if (!good) *ok = 0;
    for (int w = 0; w < TOM_NWIN; w++) {
        st_tompt(scratch + 36 * w, p);
        for (int i = 0; i < TOM_WIN_BITS; i++) p = tom_dbl(p);
    }
}
__global__ void k_tomtab_fill(uint32_t* scratch) {
    uint32_t t = gtid();
    if (t >= TOM_NWIN * TOM_WIN_SIZE) return;
    uint32_t w = t >> TOM_WIN_BITS, d = t & (TOM_WIN_SIZE - 1);
    TomPt base = ld_tompt(scratch + 36 * w);
    TomPt acc = tom_identity();
    for (int b = TOM_WIN_BITS - 1; b >= 0; b--) {
        acc = tom_dbl(acc);
        TomPt s = tom_add(acc, base);
        bool bit = (d >> b) & 1;
        acc.x = fe_select(bit, s.x, acc.x), acc.y = fe_select(bit, s.y, acc.y);
        acc.t = fe_select(bit, s.t, acc.t), acc.z = fe_select(bit, s.z, acc.z);
    }
    st_tompt(scratch + TOM_NWIN * 36 + (size_t)36 * t, acc);
}
__global__ void k_tomtab_affine(const uint32_t* scratch, uint32_t* tab) {
    uint32_t t = gtid();
    if (t >= TOM_NWIN * TOM_WIN_SIZE) return;
    TomPt a = ld_tompt(scratch + TOM_NWIN * 36 + (size_t)36 * t);
    Ft2 zi = fe_inv<ModT>(a.z);
    Ft2 x = a.x * zi, y = a.y * zi;
    Ft2 dt = (x * y) * fe_const<ModT, 1>(TOM_D1_M);
    uint32_t* e = tab + (size_t)TOM_ENTRY_WORDS * t;
#pragma unroll
    for (int l = 0; l < 9; l++) e[l] = x.l[l], e[9 + l] = y.l[l], e[18 + l] = dt.l[l];
    e[27] = 0;
}
void launch_build_tom_table(hipStream_t s, const uint32_t* xy, uint32_t* tab, uint32_t* scratch, int32_t* ok) {
    hipLaunchKernelGGL(k_tomtab_bases, dim3(1), dim3(64), 0, s, xy, scratch, ok);
    hipLaunchKernelGGL(k_tomtab_fill, dim3((TOM_NWIN * TOM_WIN_SIZE + 63) / 64), dim3(64), 0, s, scratch);
    hipLaunchKernelGGL(k_tomtab_affine, dim3((TOM_NWIN * TOM_WIN_SIZE + 63) / 64), dim3(64), 0, s, scratch, tab);
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
__global__ void k_pfix_fill(uint32_t* scratch) {
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
