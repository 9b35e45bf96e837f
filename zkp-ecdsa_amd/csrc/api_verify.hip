// zk_verify_batch / zk_verify_batch_device: host-side phase pipeline of the verifier (kernels in k_verify.hip).
#include "ctx.h"
#include "jobs.h"

static size_t vcarve(VWork& V, Soa& res, Soa& res2, MsmBuf& M, PMsmBuf& PM, uint8_t* base, uint32_t C, uint32_t sec, uint32_t n, uint64_t N, bool want_msm, bool want_pm,
                     uint32_t want_groups) {
    Carver k(base);
    V.C = C, V.sec = sec, V.n = n;
    V.st = (int32_t*)k.take(4 * (size_t)C);
    V.exp_st = (int32_t*)k.take(4 * (size_t)C);
    V.exp_jm = (uint32_t*)k.take(4 * (size_t)C), V.exp_jz = (uint32_t*)k.take(4 * (size_t)C);
    V.okflags = (uint32_t*)k.take(4 * (size_t)C);
    V.zcnt = (uint32_t*)k.take(4 * (size_t)C);
    V.hbits = (uint32_t*)k.take(16 * (size_t)C);
    V.chal = (uint32_t*)k.take(16 * (size_t)C);
    V.gkx = (uint32_t*)k.take(12 * (size_t)C);
    size_t ns = (size_t)C * VK;
    V.idx = (uint32_t*)k.take(4 * ns);
    V.t1_act = (uint32_t*)k.take(8 * ns), V.t1_cnt = (uint32_t*)k.take(256);
    V.vbytes = (uint8_t*)k.take(V_SAMPLE_FILLS * (size_t)C);
    V.vc = (uint32_t*)k.take(4 * 18 * ns);
    {
        const size_t nm = std::min<size_t>(C, V_PH_MAXP) * VK * 6;
        V.ph_msg = (uint8_t*)k.take(nm * V_PH_BLOCKS * 64), V.ph_wk = (uint32_t*)k.take(nm * V_PH_BLOCKS * 256), V.ph_nblk = (uint8_t*)k.take(nm);
    }
    V.vd = k.list(ns * 5);
    V.gk_f = k.soa((size_t)n * C), V.gk_g = k.soa((size_t)n * C);
    V.gk_total = k.soa(C);
    V.gk_csub = (uint32_t*)k.take(n >= GK_ETAB_MINN && n <= GK_ETAB_MAXN ? std::max<size_t>(36 * 256 * (size_t)C, n >= GKM_MINN ? gkm_coef_frag_bytes(C) : 0) : 16);
    V.gk_swap = (uint32_t*)k.take(4 * (size_t)n * C);
    size_t nq = (n + 1) / 2;
    // the points of the three term lists are ONE array of 128-byte entries (slot terms, membership terms, the three per proof): a term's number in it is
    // its id in the bucket pass (k_msm.hip), which gathers these entries as they are
    const size_t tcap = ns * V_SLOT_TERMS + (size_t)C * nq * 8 + (size_t)C * 3;
    uint32_t* pts = (uint32_t*)k.take(tcap * VT_ENTRY_WORDS * 4);
    size_t pts_off = 0;
    auto terms = [&](size_t cnt) {
        VTerms t{pts ? pts + pts_off * VT_ENTRY_WORDS : nullptr, k.soa(cnt), (uint32_t*)k.take(cnt * 8 * 36 * 4), (uint8_t*)k.take(cnt * 65), (uint32_t)cnt};
        pts_off += cnt;
        return t;
    };
    auto soa4 = [&](size_t cnt) { return Soa4{k.soa(cnt), k.soa(cnt), k.soa(cnt), k.soa(cnt)}; };
    V.slot_terms = terms(ns * V_SLOT_TERMS);
    V.slot_class = (uint8_t*)k.take(ns), V.slot_perm = (uint32_t*)k.take(4 * ns), V.slot_cnt = (uint32_t*)k.take(8 * (MSM_G_MAX + 1));
    V.gk_terms = terms((size_t)C * nq * 8);
    V.misc_terms = terms((size_t)C * 3);
    V.slot_acc = soa4(ns * V_SLOT_SPLIT), V.gk_acc = soa4((size_t)C * nq), V.misc_acc = soa4((size_t)C * 3);
    V.wide_acc = soa4(std::min<size_t>(C, V_WIDE_MAXP) * (VK * V_SLOT_TERMS + nq * V_WIDE_GK));
    V.sSg = k.soa(ns), V.sSh = k.soa(ns), V.sSkx = k.soa(ns), V.sSky = k.soa(ns), V.sSR = k.soa(ns), V.sSH = k.soa(ns), V.sSL = k.soa(ns);
    V.pSR = k.soa(C), V.pSH = k.soa(C), V.pSL = k.soa(C);
    V.pa_x = k.soa(ns), V.pa_y = k.soa(ns), V.pa_sc = k.soa(ns);
    V.pa_tab = (uint32_t*)k.take(ns * 8 * RTAB_ENTRY_WORDS * 4), V.pa_dig = (uint8_t*)k.take(ns * 33);
    V.pacc = k.soa3(std::max<size_t>((size_t)C * (VK / 5 + 1), std::min<size_t>(C, V_SIDE_MAXP) * (VK + 2)));
    V.clx = k.soa(C), V.cly = k.soa(C);
    V.cl_tab = (uint32_t*)k.take((size_t)C * 8 * RTAB_ENTRY_WORDS * 4), V.cl_dig = (uint8_t*)k.take((size_t)C * 35), V.p256_ok = (uint32_t*)k.take(4 * (size_t)C);
    PM = PMsmBuf{};
    if (want_pm) {   // cross-proof P-256 pass (k_pmsm.hip): it runs INSTEAD of the per-proof window tables, so it lives in their memory where that is large enough
        const size_t need = pmsm_carve(nullptr, nullptr, C, want_groups), have = ns * 8 * RTAB_ENTRY_WORDS * 4;
        uint8_t* at = need <= have ? (uint8_t*)V.pa_tab : (uint8_t*)k.take(need);
        if (base) pmsm_carve(&PM, at, C, want_groups);
    }
    M = MsmBuf{};
    if (want_msm) {   // batched Tom check (k_msm.hip): ~1.5 GB per lane, only where chunks are large enough to use it
        size_t cap = ns * V_SLOT_TERMS + (size_t)C * nq * 8 + (size_t)C * 3;
        M.cap = (uint32_t)cap;
        M.aos = pts;   // (== V.slot_terms.pts: no copy)
        // sized for either shape of the pass: 16 windows x (8 groups x 2^16 digits) or 20 windows x (64 groups x 2^13 digits)
        const size_t NW = want_groups == 64 ? 20 : 16, NBG = (size_t)1 << 19, NWG = NW * want_groups;
        M.pairs = (uint2*)k.take(cap * 8 * NW);
        M.vals_out = (uint32_t*)k.take(cap * 4 * NW);
        M.start = (uint32_t*)k.take(4 * NW * NBG), M.end = (uint32_t*)k.take(4 * NW * NBG);
        M.ord_id = (uint32_t*)k.take(4 * NW * NBG);
        M.counters = (uint32_t*)k.take(256), M.flag = (uint32_t*)k.take(256), M.big_list = (uint32_t*)k.take(4 * 4096), M.big_part = (uint32_t*)k.take((size_t)4096 * 128 * 144);
        // the buckets are written after the last reader of the (key, id) pairs has run (k_msm_binsort): where the pairs' memory is large enough they share it
        if (cap * 8 * NW >= NW * NBG * 144) M.buckets = (uint32_t*)M.pairs;
        else M.buckets = (uint32_t*)k.take(NW * NBG * 144);
        M.red = (uint32_t*)k.take(msm_red_words(want_groups) * 4);
        M.Tw = (uint32_t*)k.take(NWG * 144);
        M.sort_tmp_bytes = msm_workspace_bytes((uint32_t)cap);
        M.sort_tmp = k.take(M.sort_tmp_bytes);
        M.one = k.list(MSM_G_MAX);
    }
    uint32_t T = n >= GK_ETAB_MINN && n <= GK_ETAB_MAXN ? 8 : std::min<uint32_t>(n, 13);  // block path: one value per 256 keys
    res = k.soa((size_t)C * (N >> T));
    res2 = k.soa((size_t)C * std::max<uint64_t>(1, (N >> T) / 512));
    return k.off + 256;
}
zk_status ensure_vworkspace(zk_ctx* c, uint32_t C, uint32_t nlanes) {
    uint32_t sec = c->P.sec, n = c->n;
    const bool want_msm = c->verify_batch_min && C >= c->verify_batch_min;   // the chunk-wide sums never run on smaller chunks
    const bool want_pm = want_msm && c->p256_batch_min && C >= c->p256_batch_min;
    if (!(c->vs_C == C && c->vs_sec == sec && c->vs_n == n && c->vs_msm == want_msm && c->vs_pm == want_pm && c->vs_groups == c->verify_groups)) {
        for (auto& L : c->vl) L.ready = false;
        c->vs_C = C, c->vs_sec = sec, c->vs_n = n, c->vs_msm = want_msm, c->vs_pm = want_pm, c->vs_groups = c->verify_groups;
    }
    for (uint32_t l = 0; l < nlanes && l < ZK_MAX_LANES; l++) {
        auto& L = c->vl[l];
        if (L.ready) continue;
        size_t need = vcarve(L.V, L.res, L.res2, L.M, L.PM, nullptr, C, sec, n, c->N, want_msm, want_pm, c->verify_groups);
        if (need > L.arena_bytes) {
            if (L.arena) HIPCHK(c, hipFree(L.arena));
            L.arena = nullptr, L.arena_bytes = 0;
            HIPCHK(c, malloc_or_shed(c, &L.arena, need));
            L.arena_bytes = need;
        }
        vcarve(L.V, L.res, L.res2, L.M, L.PM, (uint8_t*)L.arena, C, sec, n, c->N, want_msm, want_pm, c->verify_groups);
        if (!L.h_msm) HIPCHK(c, hipHostMalloc((void**)&L.h_msm, 1024, hipHostMallocMapped | hipHostMallocCoherent));
        if (!L.aux_fork) HIPCHK(c, hipEventCreateWithFlags(&L.aux_fork, hipEventDisableTiming));
        for (int i = 0; i < V_AUX_STREAMS; i++) {
            if (!L.aux[i]) HIPCHK(c, hipStreamCreateWithFlags(&L.aux[i], hipStreamNonBlocking));
            if (!L.aux_done[i]) HIPCHK(c, hipEventCreateWithFlags(&L.aux_done[i], hipEventDisableTiming));
        }
        L.M.host = L.h_msm;
        L.ready = true;
    }
    for (uint32_t l = 0; l < nlanes && l < ZK_MAX_LANES; l++) c->vl[l].V.hardened = c->mode == ZK_MODE_HARDENED, c->vl[l].V.ring_digest = c->ring_digest;
    return ZK_OK;
}

static bool os_random(uint8_t* out, size_t n) {
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f) return false;
    size_t got = fread(out, 1, n, f);
    fclose(f);
    return got == n;
}
struct VMaster {
    uint32_t w[8];
};
__global__ void k_default_vseeds(uint64_t B, VMaster master, uint8_t* out) {
    // verifier seeds when the caller supplies none: SHA-256("zkv\x01" || be64(b) || master), master = 32 bytes of OS
    // randomness drawn for this call (the reference's verifier draws from crypto.getRandomValues: the checked subset and the
    // multipliers must not be predictable from public data)
    uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t m[16], h[8];
    m[0] = 0x7a6b7600u | 0x01, m[1] = (uint32_t)(b >> 32), m[2] = (uint32_t)b;
    for (int i = 0; i < 8; i++) m[3 + i] = bswap32(master.w[i]);
    m[11] = 0x80000000u, m[12] = 0, m[13] = 0, m[14] = 0, m[15] = 44 * 8;
    sha256_iv(h);
    sha256_compress(h, m);
    for (int i = 0; i < 8; i++) ((uint32_t*)out)[8 * b + i] = bswap32(h[i]);
}

zk_status VerifyJob::plan_unpack() {
    if (!d_packed) return ZK_OK;
    const uint64_t nchunks = plan.size();
    std::vector<uint64_t> pfirst(nchunks);
    if (!host_off) {   // a device-side offset array was never seen by the host: the expansion's sizing rests on it being non-decreasing (host arrays are checked by their callers)
        DevBuf flag;
        uint32_t bad = 0;
        HIPCHK(c, hipMalloc(&flag.p, 4));
        HIPCHK(c, hipMemsetAsync(flag.p, 0, 4, c->stream));
        launch_offsets_monotonic(c->stream, d_poff, B, flag.as<uint32_t>());
        HIPCHK(c, hipMemcpyAsync(&bad, flag.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (bad) {
            c->err = "proof offsets must be non-decreasing and 4-byte aligned";
            return ZK_E_ARG;
        }
    }
    for (uint64_t k = 0; k < nchunks; k++) {
        if (host_off) pfirst[k] = host_off[plan[k].first];
        else HIPCHK(c, hipMemcpy(&pfirst[k], d_poff + plan[k].first, 8, hipMemcpyDeviceToHost));   // device-pointer call: one word per chunk
    }
    ubase.resize(nchunks);
    for (uint64_t k = 0; k < nchunks; k++)   // expansion is at most 36 / 33 per proof, and 32 bytes for a proof that is shorter than its header
        ubase[k] = (((pfirst[k] * 12 + 10) / 11 + 32 * plan[k].first + 256 * k) + 255) & ~(uint64_t)255;
    return ZK_OK;
}
zk_status VerifyJob::enqueue_h2d() {
    if (!host_src) return ZK_OK;
    uint8_t* dst = d_packed ? (uint8_t*)d_packed : (uint8_t*)d_proofs;
    const uint64_t nchunks = plan.size();
    arrived.assign(nchunks, nullptr);
    for (uint64_t k = 0; k < nchunks; k++) {
        uint64_t b0 = host_off[plan[k].first], b1 = host_off[plan[k].first + plan[k].cnt];
        if (hipEventCreateWithFlags(&arrived[k], hipEventDisableTiming) != hipSuccess ||
            (b1 > b0 && hipMemcpyAsync(dst + b0, host_src + b0, b1 - b0, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess) ||
            hipEventRecord(arrived[k], c->copy_stream) != hipSuccess) {
            c->err = "host-to-device copy of the proofs failed";
            return ZK_E_DEVICE;
        }
    }
    return ZK_OK;
}
// Stage 1 (everything up to the term lists) of chunk k+1 is enqueued on the other stream before the host blocks on the
// batched Tom check of chunk k (k_msm.hip reads counters and the verdict back), so neither stream runs dry.
zk_status VerifyJob::stage1(uint64_t chunk_no) {
    const DevParams& P = c->P;
    const uint64_t first = plan[chunk_no].first;
    const uint32_t cnt = plan[chunk_no].cnt;
    const uint32_t lane = lane_of(chunk_no);
    Workspace& W = c->pl[lane].W;
    VWork& V = c->vl[lane].V;
    hipStream_t s = c->pl[lane].stream;
    const Soa& vres = c->vl[lane].res;
    const Soa& vres2 = c->vl[lane].res2;
    if (inputs_ready) HIPCHK(c, hipStreamWaitEvent(s, inputs_ready, 0));
    if (host_src && hipStreamWaitEvent(s, arrived[chunk_no], 0) != hipSuccess) {
        c->err = "hipStreamWaitEvent failed";
        return ZK_E_DEVICE;
    }
    const uint64_t* d_off = off_of(chunk_no);   // shadows the member: the chunk's own offsets when the input is packed
    {
        MaybeScope t(timed, c, "v_parse_validate", s);
        if (d_packed) launch_v_unpack(s, V.sec, cnt, d_packed, d_poff, first, ubase[chunk_no], (uint8_t*)d_proofs, d_uoff + first + chunk_no);
        launch_v_header_validate(s, V, cnt, d_proofs, d_off, first);
    }
    // A small chunk (per_proof_range below) is a chain of latencies: the two challenge hashes (one lane per proof, 16 KB each) and the membership
    // total need nothing from the P-256 front end (R's window table: 256 doublings in a row) and run beside it on an auxiliary stream.
    const bool small = side_streams(cnt);
    if (small && timed) c->timing_forked = true;
    auto& A = c->vl[lane];
    if (small) {
        // Host order = the longest chain first: a launch costs about 5 us of host time, and the main stream's chain (R's table: 256 doublings in a row, then the sampled
        // points) is what the call waits for -- with the side streams' fourteen launches in front of it the table started 0.14 ms late (profiles/r06_ab_variants.txt (13)).
        HIPCHK(c, hipEventRecord(A.aux_fork, s));   // header read and validated
        HIPCHK(c, hipStreamWaitEvent(A.aux[0], A.aux_fork, 0));
        HIPCHK(c, hipStreamWaitEvent(A.aux[1], A.aux_fork, 0));
        hipStream_t sq = A.aux[2];
        {
            MaybeScope t(timed, c, "v_p256_front_rtab", s);
            launch_v_front_r(s, W, V, cnt, d_proofs, d_off, first);
            HIPCHK(c, hipEventRecord(A.aux_fork, s));   // R
            HIPCHK(c, hipStreamWaitEvent(sq, A.aux_fork, 0));
            launch_rtab(s, W, cnt, RTAB_VERIFY_BITS);
        }
        {   // aux 2: Q = (z / R.x) G (an inversion and a comb walk) beside R's table, then the sampled repetitions as the header's bits give them (k_v_sample)
            MaybeScope t(timed, c, "v_p256_front_rtab", sq);
            launch_v_front_q(sq, P, W, V, cnt, d_proofs, d_off, d_msg, first);
        }
        {
            MaybeScope t(timed, c, "v_hash", sq);
            launch_v_sample(sq, V, cnt, d_vseeds, first);
        }
        HIPCHK(c, hipEventRecord(A.aux_done[2], sq));
        {   // aux 0: the Exp challenge (three kernels where the schedule buffer holds the chunk): 0.45 ms in one lane, needed only to CHECK the header's bits
            MaybeScope t(timed, c, "v_hash", A.aux[0]);
            if (cnt <= W.exph_cap && W.exph_wk) launch_v_exp_challenge_small(A.aux[0], W, V, cnt, d_proofs, d_off, first);
            else launch_v_challenges(A.aux[0], V, cnt, d_proofs, d_off, d_msg, first, 1);
            HIPCHK(c, hipStreamWaitEvent(A.aux[0], A.aux_done[2], 0));
            launch_v_sample_check(A.aux[0], V, cnt);   // waited for in stage 2, in front of k_v_final
        }
        HIPCHK(c, hipEventRecord(A.aux_done[0], A.aux[0]));
        {   // aux 1: the membership challenge and total
            MaybeScope t(timed, c, "v_gk_total", A.aux[1]);
            launch_v_challenges(A.aux[1], V, cnt, d_proofs, d_off, d_msg, first, 2);
            launch_v_gk_total(A.aux[1], V, W.ring, W.gk_etab, W.gk_kdig, cnt, W.N, d_proofs, d_off, first, vres, vres2);
        }
        {   // ... and the membership proof's terms (points and scalars), which need nothing else
            MaybeScope t(timed, c, "v_terms", A.aux[1]);
            launch_v_proof_points(A.aux[1], V, cnt, d_proofs, d_off, first, 1);
            launch_v_proof_terms(A.aux[1], W, V, cnt, d_proofs, d_off, d_vseeds, first);
        }
        HIPCHK(c, hipEventRecord(A.aux_done[1], A.aux[1]));
        HIPCHK(c, hipStreamWaitEvent(s, A.aux_done[2], 0));   // Q and the sampled repetitions
    } else {
        MaybeScope t(timed, c, "v_p256_front_rtab", s);
        launch_v_front_r(s, W, V, cnt, d_proofs, d_off, first);
        launch_v_front_q(s, P, W, V, cnt, d_proofs, d_off, d_msg, first);
        launch_rtab(s, W, cnt, RTAB_VERIFY_BITS);
    }
    if (!small) {
        MaybeScope t(timed, c, "v_hash", s);
        // (a chunk of this size keeps one wave per SIMD busy with one lane per proof already: the three-kernel path of the small chunks measured 1.9 ms per
        // 32 768 proofs against this kernel's 1.86, profiles/r05_ab_variants.txt)
        launch_v_challenges(s, V, cnt, d_proofs, d_off, d_msg, first, 3);
        launch_v_sample(s, V, cnt, d_vseeds, first);
        launch_v_sample_check(s, V, cnt);
    }
    {
        MaybeScope t(timed, c, "v_p256_exp_points", s);
        launch_v_exp_points(s, W, V, cnt, d_proofs, d_off, first, small ? 4 : 1);
        // a small chunk does not wait for the recomputed challenge here: a proof whose header disagrees with it is carried along as the header reads (defined data all the
        // way) and k_v_final, behind the wait in stage 2, gives it verifyExp's exception
        launch_v_exp_status(s, W, V, cnt, !small);
        if (small) {   // the slots' term points are read off the proof: beside the commitments T1 and the derived points, on Q's stream
            HIPCHK(c, hipEventRecord(A.aux_fork, s));
            HIPCHK(c, hipStreamWaitEvent(A.aux[2], A.aux_fork, 0));
            {
                MaybeScope t2(timed, c, "v_terms", A.aux[2]);
                launch_v_slot_points(A.aux[2], V, cnt, d_proofs, d_off, first);
            }
            HIPCHK(c, hipEventRecord(A.aux_done[2], A.aux[2]));
        }
        launch_p256_normalize(s, W.Tproj, cnt * VK, W.Tx, W.Ty, W.st, VK, 0, nullptr);   // identities were given their status by k_v_exp_status
    }
    {
        MaybeScope t(timed, c, "v_tom_fixed", s);
        launch_v_t1_scalars(s, W, V, cnt, d_proofs, d_off, first);
        // only the zero-bit repetitions have a T1 (exp.ts:299-330): k_v_t1_scalars lists their slots (about half of the 2 VK per proof) and gives the others the
        // identity directly; a small chunk keeps the plain launch (its commitments run four lanes wide)
        if (cnt * 2 * VK > ZK_WIDE_MAX_UNITS) launch_tom_commit_list(s, P, W.la, V.t1_act, V.t1_cnt, cnt * 2 * VK);
        else launch_tom_commit(s, P, W.la, cnt * 2 * VK, 2 * VK, 2 + 2 * W.sec);
        launch_v_derived(s, W, V, cnt, d_proofs, d_off, first);
        launch_tom_normalize(s, V.vd, cnt * VK * 5, 0, 1, 1);
    }
    {
        MaybeScope t(timed, c, "v_hash", s);
        launch_v_padd_hash(s, P, W, V, cnt, d_proofs, d_off, first);
    }
    if (!small) {
        MaybeScope t(timed, c, "v_gk_total", s);
        launch_v_gk_total(s, V, W.ring, W.gk_etab, W.gk_kdig, cnt, W.N, d_proofs, d_off, first, vres, vres2);
    }
    {
        MaybeScope t(timed, c, "v_terms", s);
        if (!small) launch_v_slot_points(s, V, cnt, d_proofs, d_off, first);
        launch_v_slot_terms(s, W, V, cnt, d_proofs, d_off, d_vseeds, first);
        if (small) launch_v_proof_points(s, V, cnt, d_proofs, d_off, first, 2);
        else {
            launch_v_proof_points(s, V, cnt, d_proofs, d_off, first, 3);
            launch_v_proof_terms(s, W, V, cnt, d_proofs, d_off, d_vseeds, first);
        }
        launch_v_proof_sums(s, W, V, cnt);
    }
    if (small) {   // the membership proof's total and terms (aux 1), the slots' points (aux 2)
        HIPCHK(c, hipStreamWaitEvent(s, A.aux_done[1], 0));
        HIPCHK(c, hipStreamWaitEvent(s, A.aux_done[2], 0));
    }
    if (p256_batched(cnt, lane)) {   // a large chunk sums its P-256 relations across proofs (k_pmsm.hip): entries, digits, counting sort and the R parts here
        MaybeScope t(timed, c, "v_msm_p256", s);
        pmsm_prepare(s, W, V, cnt, A.PM, c->vs_groups);
    } else if (!small) {   // (a small chunk's P-256 sums run in stage 2, one term per lane, beside its Tom-256 sums)
        MaybeScope t(timed, c, "v_straus_p256", s);
        launch_v_p256_straus(s, V, cnt, 5);
    }
    return ZK_OK;
}
// Per-proof sums (windowed Straus + the two fixed-base commitments) of proofs [p0, p1) of a chunk: the unchanged kernels on
// views of the term lists / accumulators that start at the range's first slot, gk group and proof.
// At most V_WIDE_MAXP proofs (every call of a few proofs: the reference's own shape is ONE, zkpAttestList.ts:150-190): what the caller waits for is the
// chain of dependent point operations of a lane, so every term gets a lane of its own (65 windows x (4 doublings + 1 addition) instead of x 13),
// k_v_acc_tree folds the accumulators into the places k_v_final reads, and the independent sums run side by side on the lane's auxiliary streams.
static zk_status host_wait(zk_ctx* c, hipEvent_t ev) {   // the host spins until `ev` has happened (a blocking call: its thread has nothing else to do)
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return ZK_OK;
        if (e != hipErrorNotReady) {
            c->err = std::string("hipEventQuery failed: ") + hipGetErrorString(e);
            return ZK_E_DEVICE;
        }
    }
}
static zk_status small_chunk_p256(zk_ctx* c, bool timed, uint32_t lane, uint32_t cnt, bool one_stream = false);
// release: the event behind stage 1 (VerifyJob::host_release) -- the auxiliary streams do not wait for it, the host does before it launches their kernels
static zk_status per_proof_range(zk_ctx* c, bool timed, hipStream_t s, uint32_t lane, const Workspace& W, const VWork& V, uint32_t p0, uint32_t p1, uint32_t range_no, uint32_t tsplit,
                                 hipEvent_t release = nullptr, uint32_t chunk_cnt = 0) {
    const DevParams& P = c->P;
    const uint32_t nq = (c->n + 1) / 2;
    const uint32_t np = p1 - p0;
    auto terms_at = [](VTerms t, size_t o) {
        t.pts += o * VT_ENTRY_WORDS, t.sc.p += o, t.tab += o * 8 * 36, t.dig += o;
        return t;
    };
    auto acc_at = [](Soa4 a, size_t o) {
        a.x.p += o, a.y.p += o, a.z.p += o, a.t.p += o;
        return a;
    };
    TomList lc = W.lc;
    {
        const size_t o = (size_t)p0 * 4 * W.n;
        for (Soa* a : {&lc.v, &lc.r, &lc.proj.x, &lc.proj.y, &lc.proj.z, &lc.ax, &lc.ay}) a->p += o;
    }
    if (np <= V_WIDE_MAXP) {
        auto& A = c->vl[lane];
        MaybeScope t(timed, c, "v_straus_tom", s);
        const bool forked = A.stage2_forked;
        if (!forked) {   // (a small chunk's streams left the main one in stage2a, which covers its FIRST range: a later range's sums reuse wide_acc,
            HIPCHK(c, hipEventRecord(A.aux_fork, s));   // so its streams start behind the range before it)
            for (int i = 0; i < 3; i++) HIPCHK(c, hipStreamWaitEvent(A.aux[i], A.aux_fork, 0));
        }
        A.stage2_forked = false;
        // host order: the longest chain first (the slots' 720 terms per proof on the main stream), the short ones last -- a launch costs about 5 us of host time
        const size_t so = (size_t)p0 * VK;
        uint32_t* perm = V.slot_perm + so;
        uint32_t* pc = V.slot_cnt + 2 * range_no;
        launch_v_slot_perm(s, V.slot_class + so, np * VK, perm, pc);
        // one term per chain while the chains fit the cooperating waves (k_coop.hip: <= ZK_COOP_MAX_CHAINS per launch, about 20 proofs); beyond that nine terms per
        // chain keep a call of up to ~200 proofs on them (4 doublings + 9 additions per window on a wave: 0.5 ms a chain, against 1.3-1.9 ms for one term in one lane)
        const uint64_t slots = (uint64_t)np * VK;
        const uint32_t ts = slots * V_SLOT_TERMS > ZK_COOP_MAX_CHAINS && slots * V_SLOT_SPLIT <= ZK_COOP_MAX_CHAINS && !zk_one_lane_chains() ? V_SLOT_SPLIT : V_SLOT_TERMS;
        launch_v_straus(s, terms_at(V.slot_terms, so), np * VK, V.C * VK, 10, 26, V.wide_acc, perm, pc, ts, ts);
        const Soa4 wgk = acc_at(V.wide_acc, (size_t)np * VK * V_SLOT_TERMS);
        if (forked && release) {   // the main stream's chain is queued to its end; the other streams' kernels follow when stage 1 is over
            launch_v_acc_tree(s, V.wide_acc, np, VK * ts, acc_at(V.slot_acc, so * V_SLOT_SPLIT), VK * V_SLOT_SPLIT, 0);
            if (zk_status zr = host_wait(c, release)) return zr;
            // host order from here: the longest chain first -- the P-256 relation (tables, sums, total: ~0.35 ms), then the membership sums, the rest
            if (!A.p256_launched) {
                if (zk_status zr = small_chunk_p256(c, timed, lane, chunk_cnt)) return zr;
                A.p256_launched = true;
            }
        }
        launch_v_straus(A.aux[0], terms_at(V.gk_terms, (size_t)p0 * nq), np * nq, V.C * nq, 4, 4, wgk, nullptr, nullptr, V_WIDE_GK, V_WIDE_GK);
        launch_v_straus(A.aux[1], terms_at(V.misc_terms, p0), np, 3 * V.C, 1, 0, acc_at(V.misc_acc, p0), nullptr, nullptr, 1, 1, 3, V.C);
        if (!(forked && release)) launch_v_acc_tree(s, V.wide_acc, np, VK * ts, acc_at(V.slot_acc, so * V_SLOT_SPLIT), VK * V_SLOT_SPLIT, 0);   // a proof's VK * ts accumulators are consecutive
        launch_v_acc_tree(A.aux[0], wgk, np, nq * V_WIDE_GK, acc_at(V.gk_acc, (size_t)p0 * nq), nq, 0);   // one sum per proof (V_FOLDED)
        launch_tom_commit(A.aux[2], P, lc, np * 2, 2, 4 * W.n);
        for (int i = 0; i < 3; i++) {
            HIPCHK(c, hipEventRecord(A.aux_done[i], A.aux[i]));
            HIPCHK(c, hipStreamWaitEvent(s, A.aux_done[i], 0));
        }
        return ZK_OK;
    }
    {
        MaybeScope t(timed, c, "v_straus_tom", s);
        const size_t so = (size_t)p0 * VK;
        uint32_t* perm = V.slot_perm + so;
        uint32_t* pc = V.slot_cnt + 2 * range_no;
        launch_v_slot_perm(s, V.slot_class + so, np * VK, perm, pc);
        launch_v_straus(s, terms_at(V.slot_terms, so), np * VK, V.C * VK, 10, 26, acc_at(V.slot_acc, so * V_SLOT_SPLIT), perm, pc, tsplit, V_SLOT_SPLIT);
        launch_v_straus(s, terms_at(V.gk_terms, (size_t)p0 * nq), np * nq, V.C * nq, 4, 4, acc_at(V.gk_acc, (size_t)p0 * nq), nullptr, nullptr);
        for (uint32_t k = 0; k < 3; k++)
            launch_v_straus(s, terms_at(V.misc_terms, (size_t)k * V.C + p0), np, 3 * V.C, 1, 0, acc_at(V.misc_acc, (size_t)k * V.C + p0), nullptr, nullptr);
    }
    {
        MaybeScope t(timed, c, "v_tom_fixed", s);
        launch_tom_commit(s, P, lc, np * 2, 2, 4 * W.n);
    }
    return ZK_OK;
}
// The P-256 relation of a small chunk (<= V_SIDE_MAXP proofs), one term per lane, on the lane's auxiliary streams 2 and 3 -- which wait for stage 1 (stage2a), or are handed
// these kernels by the host when stage 1 is over (VerifyJob::host_release) -- or on stream 3 alone (one_stream: a chunk that takes the batched checks first).
static zk_status small_chunk_p256(zk_ctx* c, bool timed, uint32_t lane, uint32_t cnt, bool one_stream) {
    const DevParams& P = c->P;
    const Workspace& W = c->pl[lane].W;
    const VWork& V = c->vl[lane].V;
    auto& A = c->vl[lane];
    hipStream_t fixed_s = one_stream ? A.aux[3] : A.aux[2];   // one_stream: only aux 3 holds a wait for stage 1 (stage2a), the table walks run behind the sums
    {
        MaybeScope t(timed, c, "v_straus_p256", A.aux[3]);
        launch_v_p256_straus(A.aux[3], V, cnt, 1);
    }
    {   // SR * R + SH * h_NIST (two table walks) beside the sums of the A_j
        MaybeScope t(timed, c, "v_p256_total", fixed_s);
        launch_v_p256_total_fixed(fixed_s, P, W, V, cnt);
    }
    if (!one_stream) {
        HIPCHK(c, hipEventRecord(A.aux_done[2], A.aux[2]));
        HIPCHK(c, hipStreamWaitEvent(A.aux[3], A.aux_done[2], 0));
    }
    {
        MaybeScope t(timed, c, "v_p256_total", A.aux[3]);
        launch_v_p256_total_sum(A.aux[3], P, W, V, cnt);
    }
    HIPCHK(c, hipEventRecord(A.aux_done[3], A.aux[3]));
    return ZK_OK;
}
zk_status VerifyJob::stage2a(uint64_t chunk_no) {
    const DevParams& P = c->P;
    const uint64_t first = plan[chunk_no].first;
    const uint32_t cnt = plan[chunk_no].cnt;
    const uint32_t lane = lane_of(chunk_no);
    const uint32_t nq = (c->n + 1) / 2;
    Workspace& W = c->pl[lane].W;
    VWork& V = c->vl[lane].V;
    const MsmBuf& M = c->vl[lane].M;
    hipStream_t s = c->pl[lane].stream;
    // Tom-256 relations: one bucket-method sum per group of the chunk's proofs (8 or 64 groups, one pass); the per-proof
    // windowed sums only run for the groups whose total is not the identity -- some proof of theirs is bad -- to tell which
    const uint32_t G = c->vs_groups;   // 8 or 64 groups per chunk (zk_ctx_set_verify_groups)
    uint32_t gsz = cnt;
    const bool wide_chunk = side_streams(cnt);
    if (wide_chunk && timed) c->timing_forked = true;
    auto& A = c->vl[lane];
    A.msm_pending = A.pm_pending = A.stage2_forked = A.released_by_host = false;   // (a failed call may have left them set)
    if (wide_chunk) {   // the P-256 sums of a small chunk, one term per lane, beside everything below: streams 2 and 3 start from here ...
        HIPCHK(c, hipEventRecord(A.aux_fork, s));
        A.stage2_forked = true;   // one fork per chunk: an event between two kernels of the main stream costs a call of one proof 10-20 us each
        // ... and a call of a few proofs launches them BEHIND the Tom-256 sums' kernels (stage2b): those chains are the longer ones, and a launch costs 5 us of host time
        A.p256_launched = cnt > V_WIDE_MAXP || (c->verify_batch_min && cnt >= c->verify_batch_min && M.cap);   // (... unless the host is going to wait for the bucket pass first)
        A.released_by_host = host_release && !A.p256_launched;   // stage2b's host waits for the event; no queue holds a wait while stage 1's tail runs
        if (A.p256_launched) {
            // ONE queue waits (two or more waiting queues cost every other queue ~17 us per kernel boundary): the P-256 relation's kernels in a row on
            // aux 3, beside the bucket pass; a range that fails the pass forks the other streams itself (per_proof_range)
            HIPCHK(c, hipStreamWaitEvent(A.aux[3], A.aux_fork, 0));
            A.stage2_forked = false;
            if (zk_status zr = small_chunk_p256(c, timed, lane, cnt, true)) return zr;
        } else if (!A.released_by_host)
            for (int i = 0; i < 4; i++) HIPCHK(c, hipStreamWaitEvent(A.aux[i], A.aux_fork, 0));
    }
    // P-256 relation: one bucket-method sum per group as well (k_pmsm.hip), on an auxiliary stream beside the Tom-256 pass; its verdicts arrive with that
    // pass's (one host round trip).  A group that fails sends the chunk through the per-proof sums.
    const bool pm = p256_batched(cnt, lane);
    uint32_t* pm_flags = A.h_msm + 128;
    if (pm) {   // the bucket sums are arithmetic: they run beside the Tom-256 pass's grouping kernels, which are not
        if (timed) c->timing_forked = true;
        HIPCHK(c, hipEventRecord(A.aux_fork, s));
        HIPCHK(c, hipStreamWaitEvent(A.aux[3], A.aux_fork, 0));
        {
            MaybeScope t(timed, c, "v_msm_p256", A.aux[3]);
            pmsm_sums(A.aux[3], P, cnt, A.PM, G, pm_flags);
        }
        HIPCHK(c, hipEventRecord(A.aux_done[3], A.aux[3]));
    }
    if (c->verify_batch_min && cnt >= c->verify_batch_min && M.cap) {
        MaybeScope t(timed, c, "v_msm_tom", s);
        TimerRec sub{"+v_msm_bucket", nullptr, nullptr};   // a part of v_msm_tom ('+': not added to the total again)
        TimerRec sub2{"+v_msm_group", nullptr, nullptr};   // the grouping of the keys (hand-written counting passes, k_msm.hip)
        if (timed) sub.e0 = get_event(c), sub.e1 = get_event(c), sub2.e0 = get_event(c), sub2.e1 = get_event(c);
        hipError_t e = run_msm(s, P, W, V, cnt, nq, M, G, &gsz, sub.e0, sub.e1, sub2.e0, sub2.e1);
        if (timed && e == hipSuccess) c->trecs.push_back(sub), c->trecs.push_back(sub2);
        if (e != hipSuccess) {
            c->err = std::string("batched verification failed: ") + hipGetErrorString(e);
            return ZK_E_DEVICE;
        }
        A.msm_pending = true;
    }
    A.pm_pending = pm, A.msm_gsz = gsz;
    if (A.msm_pending) {   // the host waits for this one and reads the groups' verdicts
        if (!A.msm_done) HIPCHK(c, hipEventCreateWithFlags(&A.msm_done, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(A.msm_done, s));
    }
    return ZK_OK;
}
zk_status VerifyJob::stage2b(uint64_t chunk_no) {
    const DevParams& P = c->P;
    const uint64_t first = plan[chunk_no].first;
    const uint32_t cnt = plan[chunk_no].cnt;
    const uint32_t lane = lane_of(chunk_no);
    Workspace& W = c->pl[lane].W;
    VWork& V = c->vl[lane].V;
    const MsmBuf& M = c->vl[lane].M;
    hipStream_t s = c->pl[lane].stream;
    const uint32_t G = c->vs_groups;
    auto& A = c->vl[lane];
    const bool wide_chunk = side_streams(cnt), pm = A.pm_pending;
    uint32_t* pm_flags = A.h_msm + 128;
    uint32_t flags[MSM_G_MAX], gsz = A.msm_gsz;
    if (A.msm_pending) {
        hipError_t e = hipEventSynchronize(A.msm_done);
        A.msm_pending = false;
        if (e != hipSuccess) {
            c->err = std::string("batched verification failed: ") + hipGetErrorString(e);
            return ZK_E_DEVICE;
        }
        msm_read_flags(M, G, flags);
        c->dbg_msm_terms += M.host[0];   // live terms of this chunk's bucket pass (zk_test_counter 2)
    } else {
        for (auto& f : flags) f = 0;   // every proof goes through the per-proof sums
    }
    uint32_t ranges = 0;
    VGroupFlags gf;
    for (uint32_t g = 0; g < MSM_G_MAX; g++) gf.v[g] = 1;
    for (uint32_t g = 0; g < G && (uint64_t)g * gsz < cnt;) {   // maximal runs of groups that failed
        if (flags[g]) {
            g++;
            continue;
        }
        uint32_t g1 = g;
        while (g1 < G && (uint64_t)g1 * gsz < cnt && !flags[g1]) g1++;
        const uint32_t p0 = g * gsz, p1 = std::min<uint32_t>(cnt, g1 * gsz);
        // few slots: a slot's 36 terms over 4 lanes (the chain of one lane is ~12 ms long, the GPU is far from full)
        uint32_t tsplit = (uint64_t)(p1 - p0) * VK * V_SLOT_SPLIT <= 524288 ? V_SLOT_SPLIT : 1;   // up to two residencies of the GPU (4 waves per SIMD, 262 144 lanes)
        if (zk_status zr = per_proof_range(c, timed, s, lane, W, V, p0, p1, ranges++, tsplit, wide_chunk && A.released_by_host ? A.aux_fork : nullptr, cnt)) return zr;
        if (p1 - p0 <= V_WIDE_MAXP) tsplit = 1 | V_FOLDED;   // folded: one accumulator per proof
        c->dbg_recheck_proofs += p1 - p0;
        for (uint32_t k = g; k < g1; k++) gf.v[k] = V_RECHECK | tsplit;
        g = g1;
    }
    if (wide_chunk && !A.p256_launched) {
        if (A.released_by_host && !ranges)   // (no range went through per_proof_range, which waits: cannot happen below the batched check's size, kept for safety)
            if (zk_status zr = host_wait(c, A.aux_fork)) return zr;
        if (zk_status zr = small_chunk_p256(c, timed, lane, cnt)) return zr;
    }
    if (wide_chunk) HIPCHK(c, hipStreamWaitEvent(s, A.aux_done[3], 0));
    A.stage2_forked = false;
    if (pm) {
        HIPCHK(c, hipEventSynchronize(A.aux_done[3]));
        HIPCHK(c, hipStreamWaitEvent(s, A.aux_done[3], 0));
        // Every proof of a group whose total is the identity has its verdict (k_pm_all_ok); the per-proof sums run over the failing groups' proofs only, as
        // maximal runs like the Tom-256 side -- one forged proof costs its group (an eighth of the chunk) the pass's gain, not the whole chunk.
        launch_pm_all_ok(s, V, cnt);
        const uint32_t parts = VK / 5 + 1;
        for (uint32_t g = 0; g < G && (uint64_t)g * gsz < cnt;) {
            if (pm_flags[g] == 1) {
                c->dbg_p256_batched += std::min<uint32_t>(cnt, (g + 1) * gsz) - g * gsz;
                g++;
                continue;
            }
            uint32_t g1 = g;
            while (g1 < G && (uint64_t)g1 * gsz < cnt && pm_flags[g1] != 1) g1++;
            const uint32_t p0 = g * gsz, p1 = std::min<uint32_t>(cnt, g1 * gsz);
            VWork Vr = V;   // the unchanged kernels on views that start at proof p0 (strides stay those of the chunk)
            Workspace Wr = W;
            const size_t a = (size_t)p0 * VK;
            Vr.pa_dig += a, Vr.cl_dig += p0, Vr.pa_tab += a * 8 * RTAB_ENTRY_WORDS, Vr.cl_tab += (size_t)p0 * 8 * RTAB_ENTRY_WORDS;
            for (Soa* q : {&Vr.pa_sc, &Vr.pa_x, &Vr.pa_y}) q->p += a;
            for (Soa* q : {&Vr.pSL, &Vr.pSR, &Vr.pSH, &Vr.clx, &Vr.cly}) q->p += p0;
            for (Soa* q : {&Vr.pacc.x, &Vr.pacc.y, &Vr.pacc.z}) q->p += (size_t)p0 * parts;
            Vr.st += p0, Vr.okflags += p0, Vr.p256_ok += p0;
            Wr.rtab += (size_t)p0 * rtab_words(RTAB_VERIFY_BITS);
            {
                MaybeScope t(timed, c, "v_straus_p256", s);
                launch_v_p256_straus(s, Vr, p1 - p0, 5);
            }
            {
                MaybeScope t(timed, c, "v_p256_total", s);
                launch_v_p256_total(s, P, Wr, Vr, p1 - p0, 5);
            }
            g = g1;
        }
    }
    if (side_streams(cnt)) HIPCHK(c, hipStreamWaitEvent(s, c->vl[lane].aux_done[0], 0));   // k_v_sample_check (stage 1): the recomputed challenge against the header's bits
    {
        MaybeScope t(timed, c, "v_final", s);
        if (!wide_chunk && !pm) launch_v_p256_total(s, P, W, V, cnt, 5);
        launch_v_final(s, W, V, cnt, d_ok, d_status, first, gf, gsz);
    }
    return ZK_OK;
}

// verifier seeds when the caller supplies none: derived on the device from 32 bytes of OS randomness drawn for this call
// Enqueued on s, not waited for: the master travels as a kernel argument (no copy), and whoever reads the seeds orders itself behind s.
zk_status make_default_vseeds(zk_ctx* c, uint64_t B, uint8_t* d_seeds /* 32 * B bytes */, hipStream_t s) {
    VMaster master;
    if (!os_random((uint8_t*)master.w, sizeof master.w)) {
        c->err = "no OS randomness for the verifier (getrandom / /dev/urandom failed)";
        return ZK_E_DEVICE;
    }
    hipLaunchKernelGGL(k_default_vseeds, dim3((uint32_t)((B + 255) / 256)), dim3(256), 0, s, B, master, d_seeds);
    volatile uint32_t* w = master.w;
    for (int i = 0; i < 8; i++) w[i] = 0;
    HIPCHK(c, hipGetLastError());
    return ZK_OK;
}

static zk_status verify_device(zk_ctx* c, uint64_t B, const uint8_t* d_msg, const uint8_t* d_proofs, const uint64_t* d_off, const uint8_t* d_vseeds, uint8_t* d_ok,
                               int32_t* d_status, const uint8_t* host_src = nullptr, const uint64_t* host_off = nullptr, bool inputs_on_stream = false) {
    if (!c->params_set || !c->N) return ZK_E_BUFFER;
    if (c->P.sec < VK) return ZK_E_SECLEVEL;
    if (B == 0) return ZK_OK;
    if (c->stream_busy) {
        c->err = "streamed jobs are in flight on this context (zk_prove_wait / zk_verify_wait them first)";
        return ZK_E_ARG;
    }
    VerifyJob J;
    J.c = c, J.B = B, J.d_msg = d_msg, J.d_proofs = d_proofs, J.d_off = d_off, J.d_ok = d_ok, J.d_status = d_status, J.host_src = host_src, J.host_off = host_off;
    J.timed = zk_timed(c, B);
    J.C = (uint32_t)std::min<uint64_t>(c->chunk, B);
    J.plan = make_chunk_plan(B, J.C, 1, false);   // uniform: see ctx.h
    J.NL = (uint32_t)std::min<size_t>(c->lanes, J.plan.size());   // chunks rotate over NL streams / workspaces
    {
        static const bool no_release = getenv("ZKATTEST_NO_HOST_RELEASE") != nullptr;   // A/B switch (tools/ab_release.sh)
        J.host_release = J.plan.size() == 1 && B <= V_WIDE_MAXP && !no_release && !zk_one_lane_chains();
    }
    if (c->wire == ZK_WIRE_ZKA1P) {   // the proofs handed in are packed: every chunk is expanded into the context's staging first
        uint64_t total = 0;
        if (host_off) total = host_off[B];
        else HIPCHK(c, hipMemcpy(&total, d_off + B, 8, hipMemcpyDeviceToHost));
        const size_t need = unpack_stage_bytes(B, total, J.C), ents = unpack_off_entries(B, J.C);
        if (need > c->unp_bytes) {
            if (c->unp_buf) HIPCHK(c, hipFree(c->unp_buf));
            c->unp_buf = nullptr, c->unp_bytes = 0;
            HIPCHK(c, hipMalloc(&c->unp_buf, need));
            c->unp_bytes = need;
        }
        if (ents > c->unp_off_entries) {
            if (c->unp_off) HIPCHK(c, hipFree(c->unp_off));
            c->unp_off = nullptr, c->unp_off_entries = 0;
            HIPCHK(c, hipMalloc((void**)&c->unp_off, 8 * ents));
            c->unp_off_entries = ents;
        }
        J.d_packed = d_proofs, J.d_poff = d_off, J.d_proofs = (const uint8_t*)c->unp_buf, J.d_uoff = c->unp_off;
        zk_status zp = J.plan_unpack();
        if (zp) return zp;
    }
    zk_status zs = ensure_workspace(c, J.C, J.NL);
    if (zs) return zs;
    zs = ensure_vworkspace(c, J.C, J.NL);
    if (zs) return zs;
    timing_begin(c);
    if (!d_vseeds) {   // the verifier's own seeds: the context's grow-only buffer, derived on c->stream (no allocation, copy or host wait per call)
        if (c->seed_bytes < 32 * B) {
            if (c->seed_buf) HIPCHK(c, hipFree(c->seed_buf));
            c->seed_buf = nullptr, c->seed_bytes = 0;
            HIPCHK(c, hipMalloc(&c->seed_buf, 32 * B + 32 * B / 4 + 4096));
            c->seed_bytes = 32 * B + 32 * B / 4 + 4096;
        }
        zs = make_default_vseeds(c, B, (uint8_t*)c->seed_buf, c->stream);
        if (zs) return zs;
        d_vseeds = (const uint8_t*)c->seed_buf;
        inputs_on_stream = true;
    }
    J.d_vseeds = d_vseeds;
    if (inputs_on_stream) {   // the lanes order themselves behind what c->stream carries for this call (input copies, the seeds' kernel)
        zs = ensure_h_stage(c, 0);
        if (zs) return zs;
        HIPCHK(c, hipEventRecord(c->in_ready, c->stream));
        J.inputs_ready = c->in_ready;
    }
    auto drain = [&] {   // nothing of this call may still be running when it returns
        for (uint32_t l = 0; l < J.NL; l++) {
            hipStreamSynchronize(c->pl[l].stream);
            for (hipStream_t a : c->vl[l].aux)   // joined into the lane's stream by events on every regular path; an error may have cut a fork short
                if (a) hipStreamSynchronize(a);
        }
        if (host_src) hipStreamSynchronize(c->copy_stream);
    };
    zs = J.enqueue_h2d();
    if (zs) {
        drain();
        return zs;
    }
    const uint64_t nchunks = J.plan.size();
    // stage 1 of the next NL - 1 chunks is enqueued on the other lanes before the host blocks on this chunk's batched check
    // ... and so are the batched passes of those chunks (stage2a: no host round trip), so that the reductions that end one chunk's pass -- dependent chains, 3 ms with
    // the GPU nearly idle -- run beside the next chunk's bucket sums; the host blocks in stage2b of the oldest chunk only
    for (uint64_t k = 0; k < nchunks && !zs; k++) {
        while (!zs && J.next_s1 < nchunks && J.next_s1 < k + J.NL) zs = J.stage1(J.next_s1++);
        while (!zs && J.next_s2a < J.next_s1) zs = J.stage2a(J.next_s2a++);
        if (!zs) zs = J.stage2b(k);
    }
    hipError_t e1 = hipSuccess;
    for (uint32_t l = 0; l < J.NL; l++) {
        hipError_t e = hipStreamSynchronize(c->pl[l].stream);
        if (e1 == hipSuccess) e1 = e;
    }
    hipError_t e3 = host_src ? hipStreamSynchronize(c->copy_stream) : hipSuccess;
    if (zs || e1 != hipSuccess) drain();   // an error between a fork and its join may have left kernels on an auxiliary stream: they finish before the buffers go
    if (zs) return zs;
    HIPCHK(c, e1);
    HIPCHK(c, e3);
    HIPCHK(c, hipGetLastError());
    timing_end(c);
    return ZK_OK;
}

extern "C" zk_status zk_verify_batch_device(zk_ctx* c, uint64_t B, const void* d_msg, const void* d_proofs, const void* d_off, const void* d_vseeds, void* d_ok,
                                            void* d_status) {
    if (!c || (B && (!d_msg || !d_proofs || !d_off || !d_ok || !d_status))) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    return verify_device(c, B, (const uint8_t*)d_msg, (const uint8_t*)d_proofs, (const uint64_t*)d_off, (const uint8_t*)d_vseeds, (uint8_t*)d_ok, (int32_t*)d_status);
}
extern "C" zk_status zk_verify_batch(zk_ctx* c, uint64_t B, const uint8_t* msg, const uint8_t* proofs, const uint64_t* off, const uint8_t* vseeds, uint8_t* ok,
                                     int32_t* status) {
    if (!c || (B && (!msg || !proofs || !off || !ok || !status))) return ZK_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->params_set || !c->N) return ZK_E_BUFFER;
    if (B == 0) return ZK_OK;
    if (c->stream_busy) {
        c->err = "streamed jobs are in flight on this context (zk_prove_wait / zk_verify_wait them first)";
        return ZK_E_ARG;
    }
    if (off[0] != 0) return ZK_E_ARG;
    for (uint64_t b = 0; b < B; b++)
        if (off[b + 1] < off[b]) return ZK_E_ARG;  // every proof lies inside [0, off[B])
    uint64_t total = off[B];
    // the small per-proof arrays live in the context's grow-only input buffer (no hipMalloc / hipFree per call)
    Carver k0(nullptr);
    auto carve_in = [&](Carver& kk, uint8_t*& m, uint64_t*& o, uint8_t*& okp, int32_t*& st_, uint8_t*& sd) {
        m = (uint8_t*)kk.take(32 * B), o = (uint64_t*)kk.take(8 * (B + 1)), okp = (uint8_t*)kk.take(B), st_ = (int32_t*)kk.take(4 * B);
        sd = (uint8_t*)kk.take(vseeds ? 32 * B : 32);
    };
    uint8_t *d_msg, *d_ok, *d_seeds;
    uint64_t* d_off;
    int32_t* d_st;
    carve_in(k0, d_msg, d_off, d_ok, d_st, d_seeds);
    zk_status zs = ensure_in_buf(c, k0.off + 256);
    if (zs) return zs;
    Carver k1((uint8_t*)c->in_buf);
    carve_in(k1, d_msg, d_off, d_ok, d_st, d_seeds);
    zs = ensure_io_buf(c, total + 64);  // proof bytes: the context's grow-only staging buffer
    if (zs) return zs;
    uint8_t* d_proofs = (uint8_t*)c->io_buf;
    // page-locked `proofs` (zk_host_alloc): chunk-wise DMA under the kernels of the earlier chunks; pageable: one blocking copy
    const bool pinned = host_ptr_is_pinned(proofs);
    if (pinned) {
        zs = ensure_copy_stream(c);
        if (zs) return zs;
    } else {
        HIPCHK(c, hipMemcpy(d_proofs, proofs, total, hipMemcpyHostToDevice));
    }
    // A call of a few proofs: messages, offsets and seeds cross in ONE copy out of the context's page-locked mirror of in_buf, and the lanes wait for its event
    // instead of the host (api.hip: zk_prove_batch does the same); verdicts and statuses come back in one.
    const size_t in_end = (size_t)(d_ok - (uint8_t*)c->in_buf), res_end = (size_t)((uint8_t*)(d_st + B) - (uint8_t*)c->in_buf);
    const bool staged = k1.off <= ZK_STAGE_MAX;
    if (staged) {
        if ((zs = ensure_h_stage(c, k1.off))) return zs;
        auto at = [&](const void* d) { return c->h_stage + ((const uint8_t*)d - (const uint8_t*)c->in_buf); };
        memcpy(at(d_msg), msg, 32 * B), memcpy(at(d_off), off, 8 * (B + 1));
        HIPCHK(c, hipMemcpyAsync(c->in_buf, c->h_stage, in_end, hipMemcpyHostToDevice, c->stream));
        if (vseeds) {
            memcpy(at(d_seeds), vseeds, 32 * B);
            HIPCHK(c, hipMemcpyAsync(d_seeds, at(d_seeds), 32 * B, hipMemcpyHostToDevice, c->stream));
        }
    } else {
        HIPCHK(c, hipMemcpyAsync(d_msg, msg, 32 * B, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_off, off, 8 * (B + 1), hipMemcpyHostToDevice, c->stream));
        if (vseeds) HIPCHK(c, hipMemcpyAsync(d_seeds, vseeds, 32 * B, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));   // both lanes read these arrays
    }
    zs = verify_device(c, B, d_msg, d_proofs, d_off, vseeds ? d_seeds : nullptr, d_ok, d_st, pinned ? proofs : nullptr, pinned ? off : nullptr, staged);
    if (zs) {
        if (staged) (void)hipStreamSynchronize(c->stream);   // the mirror is reused by the next call
        return zs;
    }
    if (staged) {   // verdicts and statuses lie next to each other in in_buf
        HIPCHK(c, hipMemcpy(c->h_stage + in_end, d_ok, res_end - in_end, hipMemcpyDeviceToHost));
        memcpy(ok, c->h_stage + in_end, B);
        memcpy(status, c->h_stage + ((uint8_t*)d_st - (uint8_t*)c->in_buf), 4 * B);
    } else {
        HIPCHK(c, hipMemcpy(ok, d_ok, B, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(status, d_st, 4 * B, hipMemcpyDeviceToHost));
    }
    return ZK_OK;
}
