// retrieval.hip -- flat-L2 index retrieval (rvc/src/rvc.rs:159 is a TODO in the reference; definition: SURVEY.md Appendix A.4, BASELINE configs 3-5):
// index load and its device-side layouts, the search section of an infer plan, the index entry points of the C ABI.
#include "engine_int.h"

namespace rvc {

// The retrieval section of an infer plan: queries from the ContentVec output, one-pass approximate scan (or one implicit GEMM for many streams),
// exact re-rank + blend into `phone`, exhaustive fallback for streams whose candidate set overflowed.
std::atomic<int> g_knn_test_lose{0};
static void build_exhaustive(rvc_engine *e, Plan &pl, int B, int C, int nq, int nblk, int first_raw, uint32_t skip_head, uint32_t R, int T, const T1 &phone, const T1 &cvo,
                             bool fast, int *d_overflow, float *d_q = nullptr, float *cand_d = nullptr, int *cand_i = nullptr);

void build_retrieval(rvc_engine *e, Plan &pl, int B, int T, int C, uint32_t skip_head, uint32_t R, const T1 &phone)
{
        if (e->index_dim != (size_t)C) throw std::runtime_error("index dimension does not match the feature dimension");
        // unique raw frames behind the sliced frames (Q2): first_raw .. last_raw
        const int first_raw = std::min((int)skip_head / 2, T - 1), last_raw = std::min((int)(skip_head + R - 1) / 2, T - 1);
        const int nq = last_raw - first_raw + 1;
        const int nblk = (int)((e->index_n + 255) / 256);
        pl.d_knn_idx = (int *)pl.arena.alloc((size_t)B * R * KNN_K * sizeof(int));
        pl.d_knn_dist = pl.arena.floats((size_t)B * R * KNN_K);
        T1 cvo = pl.cv_out;
        // Stage A + B: approximate distances on the matrix cores in one pass over the index (HBM-bound), exact re-rank of a
        // provably sufficient candidate set.
        const bool fast = C % 16 == 0 && !test_opt("RVC_KNN_EXHAUSTIVE");
        // many streams: all queries against the index as ONE implicit GEMM (queries = weight operand in fragment order, transposed
        // index = activation operand, -|y|^2 / 2 as a per-column residual, scale -2): one pass over the index instead of one per 16
        // queries (64 streams x 11 queries: 44 passes, 3.5 ms -> one ~1 ms MFMA-bound launch).  Same approximate distances up to
        // fp32 summation order; the exact re-rank behind it is unchanged.
        const int Q = B * nq, Qpad = (Q + 127) / 128 * 128;
        // (the GEMM path addresses its operands with 32-bit byte / element offsets: the one-pass scan, whose strides are 64-bit, takes
        // indexes beyond that range)
        const bool gemm_fits = (size_t)C * e->index_n * sizeof(float) < ((size_t)1 << 31) && (size_t)Qpad * e->index_n < ((size_t)1 << 31);
        const bool gemm_scan = fast && Q >= 128 && gemm_fits && e->d_nhn && !test_opt("RVC_KNN_NO_GEMM");
        if (gemm_scan || !fast) ensure_index_transposed(e);
        if (fast && !gemm_scan) {
            // one stream / few streams: ONE launch per group of 16 queries (knn_scan_select_kernel: scan, select, exact re-rank, blend)
            // three workgroups per CU of the stream this runs on, all resident at once (the ContentVec branch is CU-masked at <= 4 streams)
            const unsigned knn_wgs = tune_env("RVC_KNN_WGS") ? (unsigned)atoi(tune_env("RVC_KNN_WGS")) : 3u * (unsigned)e->cv_cus;
            const unsigned G = std::min(std::min((unsigned)((e->index_n + 63) / 64), std::max(knn_wgs / (unsigned)B, 64u)), (unsigned)KNN_FUSED_MAXG);
            const int ngroups = (nq + 15) / 16;
            unsigned long long *lists = (unsigned long long *)pl.arena.alloc((size_t)ngroups * B * 16 * G * KNN_K * sizeof(unsigned long long));
            unsigned *ticket = (unsigned *)pl.arena.alloc((size_t)ngroups * B * 2 * sizeof(unsigned));
            HIPCHK(hipMemset(ticket, 0, (size_t)ngroups * B * 2 * sizeof(unsigned)));
            pl.knn_ticket = ticket; pl.knn_ticket_bytes = (size_t)ngroups * B * 2 * sizeof(unsigned);
            for (int gi = 0; gi < ngroups; gi++) {
                const size_t lds = knn_fused_lds_floats(C, std::min(16, nq - gi * 16), (int)G) * sizeof(float);
                if (lds > 128 * 1024) throw ShapeError("feature dimension too large for the retrieval kernel");
                KnnFusedP fp{}; fp.indexF = e->d_indexF; fp.index = e->d_index; fp.ynorm = e->d_ynorm; fp.n = (int)e->index_n; fp.dim = C;
                fp.cv = cvo.p; fp.cv_cs = cvo.ld; fp.cv_bs = cvo.bs; fp.first_raw = first_raw; fp.nq = nq; fp.q0 = gi * 16;
                fp.lists = lists + (size_t)gi * B * 16 * G * KNN_K; fp.ticket = ticket + (size_t)gi * B * 2;
                fp.skip_head = (int)skip_head; fp.T = T; fp.R = (int)R; fp.rate = e->index_rate;
                fp.phone = phone.p; fp.ph_cs = phone.ld; fp.ph_bs = phone.bs; fp.out_idx = pl.d_knn_idx; fp.out_dist = pl.d_knn_dist;
                fp.status = &e->d_state[0].status; fp.status_stride = (int)(sizeof(StreamState) / sizeof(int));
                fp.spin_limit = 1u << 22;
                dim3 grid(G, B);
                Plan *plp = &pl;
                const double scan_bytes = (double)e->index_n * C * sizeof(float) * B;     // algorithmic bytes: the index, read once per query group
                pl.ops.push_back([=](hipStream_t s) {
                    ProfEvent *pe = nullptr;
                    if (plp->profile) {
                        if (plp->prof_used == plp->prof.size()) { ProfEvent ev; HIPCHK(hipEventCreate(&ev.a)); HIPCHK(hipEventCreate(&ev.b)); ev.flops = 0; ev.bytes = 0; plp->prof.push_back(ev); }
                        pe = &plp->prof[plp->prof_used++]; pe->flops = 0; pe->bytes = scan_bytes;
                    }
                    if (g_knn_test_lose.load(std::memory_order_relaxed)) {       // test hook (rvc_debug_option RVC_KNN_LOSE_TICKET): a hand-off that cannot complete
                        KnnFusedP f2 = fp; f2.test_lose = 1; f2.spin_limit = 1u << 12;
                        hipLaunchKernelGGL(knn_scan_select_kernel, grid, dim3(256), lds, s, f2);
                    } else if (pe) hipExtLaunchKernelGGL(knn_scan_select_kernel, grid, dim3(256), (uint32_t)lds, s, pe->a, pe->b, 0, fp);
                    else hipLaunchKernelGGL(knn_scan_select_kernel, grid, dim3(256), lds, s, fp);
                });
            }
            // What the engine runs instead when a selector gave up (ST_KNN_TIMEOUT: a workgroup of the launch did not arrive in time, e.g. on a GPU
            // shared with another process): the exhaustive exact scan over the row-major index + merge + blend, built into a list of its own.
            // The chunk is then recomputed from `phone` on and the call returns RVC_OK (engine.hip recover_retrieval).
            if (!pl.bucket) {
                OpList main_ops = std::move(pl.ops);
                pl.ops = OpList();
                build_exhaustive(e, pl, B, C, nq, nblk, first_raw, skip_head, R, T, phone, cvo, false, nullptr);
                pl.knn_fallback = std::move(pl.ops.v);
                pl.ops = std::move(main_ops);
            }
            return;
        }
        // many streams (GEMM scan) or the exhaustive definition: explicit query rows, candidate lists per 256-vector block
        float *d_q = pl.arena.floats((size_t)B * nq * C);
        float *cand_d = pl.arena.floats((size_t)B * nq * nblk * KNN_K);
        int *cand_i = (int *)pl.arena.alloc((size_t)B * nq * nblk * KNN_K * sizeof(int));
        {
            dim3 grid((nq * C + 255) / 256, B);
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(knn_queries_kernel, grid, dim3(256), 0, s, cvo.p, cvo.ld, cvo.bs, C, first_raw, nq, d_q); });
        }
        // the exhaustive exact scan below runs for every stream (the definition), or only for streams whose candidate set overflowed
        int *d_overflow = (int *)pl.arena.alloc((size_t)B * sizeof(int));
        if (fast) {
            float *d_approx = pl.arena.floats((size_t)Qpad * e->index_n);
            float *d_qf = pl.arena.floats((size_t)Qpad * C);
            const int n_idx = (int)e->index_n;
            {
                dim3 grid(Qpad / 16, C / 16); int *ovf = d_overflow; const int nb = B;
                pl.ops.push_back([=](hipStream_t s) {
                    HIPCHK(hipMemsetAsync(ovf, 0, (size_t)nb * sizeof(int), s));
                    hipLaunchKernelGGL(knn_pack_queries_kernel, grid, dim3(64), 0, s, d_q, Q, C, d_qf);
                });
            }
            ConvW qw; qw.w = d_qf; qw.bias = nullptr; qw.M = Qpad; qw.K = C; qw.Kp = C; qw.Cin = C; qw.Cout = Qpad; qw.KW = 1; qw.groups = 1; qw.nphase = 1; qw.owns = false;
            T1 xi; xi.p = e->d_indexT; xi.B = 1; xi.C = C; xi.T = n_idx; xi.ld = n_idx; xi.halo = 0; xi.bs = (long long)C * n_idx;
            T1 ya; ya.p = d_approx; ya.B = 1; ya.C = Qpad; ya.T = n_idx; ya.ld = n_idx; ya.halo = 0; ya.bs = (long long)Qpad * n_idx;
            ConvOpts o; o.no_bias = true; o.res = e->d_nhn; o.res_cs = 0; o.res_bs = 0; o.scale = -2.0f;
            add_conv1d(pl, qw, xi, ya, 1, 0, 1, o);
            KnnSelP sp{}; sp.approx = d_approx; sp.approx_bs = (long long)nq * e->index_n; sp.n = (int)e->index_n; sp.dim = C; sp.nq = nq;
            sp.index = e->d_index; sp.q = d_q; sp.q_bs = (long long)nq * C; sp.skip_head = (int)skip_head; sp.T = T; sp.R = (int)R; sp.first_raw = first_raw;
            sp.rate = e->index_rate; sp.phone = phone.p; sp.ph_cs = phone.ld; sp.ph_bs = phone.bs; sp.out_idx = pl.d_knn_idx; sp.out_dist = pl.d_knn_dist;
            sp.overflow = d_overflow;
            dim3 sgrid(nq, B);
            const size_t slds = (size_t)33 * (C + 4) * sizeof(float);
            if (slds > 128 * 1024) throw ShapeError("feature dimension too large for the retrieval kernel");
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(knn_select_blend_kernel, sgrid, dim3(1024), slds, s, sp); });
        }
        build_exhaustive(e, pl, B, C, nq, nblk, first_raw, skip_head, R, T, phone, cvo, fast, d_overflow, d_q, cand_d, cand_i);
}

// The exhaustive exact scan (= the definition) + merge + blend.  d_q == nullptr: the section also gathers its own queries and owns its buffers.
static void build_exhaustive(rvc_engine *e, Plan &pl, int B, int C, int nq, int nblk, int first_raw, uint32_t skip_head, uint32_t R, int T, const T1 &phone, const T1 &cvo,
                             bool fast, int *d_overflow, float *d_q, float *cand_d, int *cand_i)
{
        if (!d_q) {
            d_q = pl.arena.floats((size_t)B * nq * C);
            cand_d = pl.arena.floats((size_t)B * nq * nblk * KNN_K);
            cand_i = (int *)pl.arena.alloc((size_t)B * nq * nblk * KNN_K * sizeof(int));
            dim3 grid((nq * C + 255) / 256, B);
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(knn_queries_kernel, grid, dim3(256), 0, s, cvo.p, cvo.ld, cvo.bs, C, first_raw, nq, d_q); });
        }
        for (int q0 = 0; q0 < nq; q0 += KNN_MAXQ) {
            const int qn = std::min(KNN_MAXQ, nq - q0);
            KnnP kp{}; kp.indexT = e->d_indexT; kp.index = e->d_index; kp.n = (int)e->index_n; kp.dim = C; kp.nblk = nblk;
            kp.v_stride = e->d_indexT ? 1 : C; kp.d_stride = e->d_indexT ? (long long)e->index_n : 1;
            // query sub-range: pointers offset so that [B][nq] strides stay those of the full arrays
            kp.q = d_q + (size_t)q0 * C; kp.nq = qn; kp.cand_d = cand_d + (size_t)q0 * nblk * KNN_K; kp.cand_i = cand_i + (size_t)q0 * nblk * KNN_K;
            kp.overflow = fast ? d_overflow : nullptr;
            const int nq_total = nq;
            dim3 grid(nblk, B);
            pl.ops.push_back([=](hipStream_t s) {
                KnnP k2 = kp; k2.q_bs = (long long)nq_total * C; k2.cand_bs = (long long)nq_total * nblk * KNN_K;
                hipLaunchKernelGGL(knn_scan_kernel, grid, dim3(256), 0, s, k2);
            });
        }
        KnnBlendP bp{}; bp.cand_d = cand_d; bp.cand_i = cand_i; bp.nblk = nblk; bp.nq = nq; bp.index = e->d_index; bp.dim = C; bp.q = d_q;
        bp.skip_head = (int)skip_head; bp.T = T; bp.R = (int)R; bp.first_raw = first_raw; bp.rate = e->index_rate;
        bp.phone = phone.p; bp.ph_cs = phone.ld; bp.ph_bs = phone.bs; bp.out_idx = pl.d_knn_idx; bp.out_dist = pl.d_knn_dist;
        bp.overflow = fast ? d_overflow : nullptr;
        dim3 grid(nq, B);
        pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(knn_merge_blend_kernel, grid, dim3(256), 0, s, bp); });
}

// // Everything the retrieval kernels need besides the row-major matrix, built ON THE DEVICE from the copy that is already in HBM
// (uploaded once, or delivered by the RCCL broadcast): the MFMA-fragment-order copy for the one-pass approximate scan and the vector
// norms.  No host round trip (round 2 copied the 307 MB matrix back to the host, repacked it in a single-threaded loop and uploaded two
// more copies: seconds per rank behind a 2 ms broadcast).  The transposed copy is NOT built here: see ensure_index_transposed.
void build_index_aux(rvc_engine *e)
{
    if (e->d_indexT) { (void)hipFree(e->d_indexT); e->d_indexT = nullptr; }
    if (e->d_indexF) { (void)hipFree(e->d_indexF); e->d_indexF = nullptr; }
    if (e->d_ynorm) (void)hipFree(e->d_ynorm);
    if (e->d_nhn) (void)hipFree(e->d_nhn);
    e->d_ynorm = e->d_nhn = nullptr;
    hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    HIPCHK(hipEventRecord(a, e->stream));
    if (e->index_dim % 16 == 0) {
        const long long nt = ((long long)e->index_n + 15) / 16, nc = (long long)e->index_dim / 16, total4 = nt * nc * 64;
        HIPCHK(hipMalloc(&e->d_indexF, (size_t)total4 * 4 * sizeof(float)));
        hipLaunchKernelGGL(knn_pack_index_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, e->stream, e->d_index, (long long)e->index_n, (int)e->index_dim, e->d_indexF, total4);
    }
    HIPCHK(hipMalloc(&e->d_ynorm, e->index_n * sizeof(float)));
    HIPCHK(hipMalloc(&e->d_nhn, e->index_n * sizeof(float)));
    hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)((e->index_n + 255) / 256)), dim3(256), 0, e->stream, e->d_index, (int)e->index_n, (int)e->index_dim, e->d_ynorm, e->d_nhn);
    HIPCHK(hipEventRecord(b, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, a, b));
    e->index_prep_ms = ms;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
}

// [dim][n] copy of the index, built by a device transpose the first time a plan needs it: the many-stream distance GEMM (the index is
// its activation operand) and the forced / non-MFMA exhaustive scan.  A single stream never builds it (HBM then holds the index twice:
// row-major for the exact re-rank and the blend, fragment order for the scan); its degenerate-data fallback walks the row-major copy.
void ensure_index_transposed(rvc_engine *e)
{
    if (e->d_indexT || !e->d_index) return;
    HIPCHK(hipMalloc(&e->d_indexT, e->index_n * e->index_dim * sizeof(float)));
    dim3 grid((unsigned)((e->index_n + 31) / 32), (unsigned)((e->index_dim + 31) / 32));
    hipLaunchKernelGGL(knn_transpose_kernel, grid, dim3(256), 0, e->stream, e->d_index, (long long)e->index_n, (int)e->index_dim, e->d_indexT);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
}

void retrieval_kernel_attrs()
{
    HIPCHK(hipFuncSetAttribute((const void *)knn_select_blend_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));   // + ~5 KB static
    HIPCHK(hipFuncSetAttribute((const void *)knn_scan_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
}

}  // namespace rvc

using namespace rvc;
extern "C" {

rvc_status rvc_load_index(rvc_engine *e, const float *vectors, size_t n, size_t dim)
{
    return guarded(e, [&]() {
        if (n < KNN_K || dim < 1) throw ShapeError("index needs at least 4 vectors");
        HIPCHK(hipDeviceSynchronize());
        if (e->d_index && e->index_owned) (void)hipFree(e->d_index);
        HIPCHK(hipMalloc(&e->d_index, n * dim * sizeof(float)));
        e->index_owned = true;
        HIPCHK(hipMemcpy(e->d_index, vectors, n * dim * sizeof(float), hipMemcpyHostToDevice));
        e->index_n = n; e->index_dim = dim;
        build_index_aux(e);
        e->plans.clear(); e->last_plan = nullptr;
        return RVC_OK;
    });
}

rvc_status rvc_load_index_device(rvc_engine *e, const void *d_vectors, size_t n, size_t dim)
{
    return guarded(e, [&]() {
        if (n < KNN_K || dim < 1) throw ShapeError("index needs at least 4 vectors");
        HIPCHK(hipDeviceSynchronize());
        if (e->d_index && e->index_owned) (void)hipFree(e->d_index);
        HIPCHK(hipMalloc(&e->d_index, n * dim * sizeof(float)));
        e->index_owned = true;
        HIPCHK(hipMemcpy(e->d_index, d_vectors, n * dim * sizeof(float), hipMemcpyDeviceToDevice));
        e->index_n = n; e->index_dim = dim;
        build_index_aux(e);
        e->plans.clear(); e->last_plan = nullptr;
        return RVC_OK;
    });
}

void *rvc_index_device_ptr(rvc_engine *e, size_t *bytes)
{
    if (!e || !e->d_index) { if (bytes) *bytes = 0; return nullptr; }
    if (bytes) *bytes = e->index_n * e->index_dim * sizeof(float);
    return e->d_index;
}

rvc_status rvc_get_knn(rvc_engine *e, int32_t *idx, float *dist, size_t cap_rows, size_t *rows)
{
    return guarded(e, [&]() {
        Plan *pl = e->last_plan;
        // (after rvc_infer_batch_g the last plan is one geometry bucket's, in bucket-local stream order: no rows are reported for such a call)
        if (!pl || !pl->with_index || !e->last_knn_rows) { if (rows) *rows = 0; return RVC_OK; }
        // stream 0's return_length rows always; the further streams' rows (stream-major) as far as the caller's capacity holds whole streams
        if (cap_rows < pl->R) { if (rows) *rows = pl->R; return RVC_SHAPE; }
        const size_t r = pl->R * std::min((size_t)pl->B, cap_rows / pl->R);
        if (rows) *rows = r;
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(idx, pl->d_knn_idx, r * KNN_K * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(dist, pl->d_knn_dist, r * KNN_K * sizeof(float), hipMemcpyDeviceToHost));
        return RVC_OK;
    });
}

// chunks whose retrieval was recomputed through the exhaustive launches after a hand-off time-out of the one-launch form (they returned RVC_OK)
long long rvc_retrieval_recoveries(rvc_engine *e) { return e ? e->knn_recoveries : 0; }

}  // extern "C"
