// knn_probe.hip -- TEST INFRASTRUCTURE: the one-launch retrieval kernel (knn_scan_select_kernel of kernels.hip.h) alone, on BASELINE's index
// (100 k x 768 fp32, N(0, 0.35^2), 11 queries): launch time with the index cold (1 GiB of other traffic between launches), wall-clock stamps
// of the phases (scan end, ticket, all arrived, lists read, exact re-rank, end), and the hits against the exhaustive definition
// (knn_scan_kernel + knn_merge_blend_kernel of the same header).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRVC_KNN_STAMPS -I obs_rvc_amd/csrc -I include tests/tools/knn_probe.hip -o knn_probe && ./knn_probe [G] [n]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#include <random>
#include "kernels.hip.h"
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace rvc;

int main(int argc, char **argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 768;
    const long long n = argc > 2 ? atoll(argv[2]) : 100000;
    const int dim = 768, T = 111, R = 21, skip_head = 200, first_raw = 100, nq = 11, ld = 128;
    std::mt19937 rng(7); std::normal_distribution<float> nd(0.f, 0.35f);
    std::vector<float> hidx((size_t)n * dim), hcv((size_t)dim * ld);
    for (auto &v : hidx) v = nd(rng);
    for (auto &v : hcv) v = nd(rng);
    float *d_index, *d_indexF, *d_ynorm, *d_nhn, *d_cv, *d_phone, *d_phone2, *d_dist, *d_dist2, *d_q, *cand_d; int *d_idx, *d_idx2, *cand_i, *d_status;
    const long long nt = (n + 15) / 16, ncq = dim / 16, total4 = nt * ncq * 64;
    CHK(hipMalloc(&d_index, (size_t)n * dim * 4)); CHK(hipMalloc(&d_indexF, (size_t)total4 * 16)); CHK(hipMalloc(&d_ynorm, n * 4)); CHK(hipMalloc(&d_nhn, n * 4));
    CHK(hipMalloc(&d_cv, hcv.size() * 4)); CHK(hipMalloc(&d_phone, (size_t)dim * 32 * 4)); CHK(hipMalloc(&d_phone2, (size_t)dim * 32 * 4));
    CHK(hipMalloc(&d_dist, R * 4 * 4)); CHK(hipMalloc(&d_dist2, R * 4 * 4)); CHK(hipMalloc(&d_idx, R * 4 * 4)); CHK(hipMalloc(&d_idx2, R * 4 * 4)); CHK(hipMalloc(&d_status, 64));
    CHK(hipMemcpy(d_index, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_cv, hcv.data(), hcv.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemset(d_phone, 0, (size_t)dim * 32 * 4)); CHK(hipMemset(d_phone2, 0, (size_t)dim * 32 * 4)); CHK(hipMemset(d_status, 0, 64));
    hipLaunchKernelGGL(knn_pack_index_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, 0, d_index, n, dim, d_indexF, total4);
    hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_index, (int)n, dim, d_ynorm, d_nhn);
    CHK(hipDeviceSynchronize());
    // the definition: exhaustive exact scan
    const int nblk = (int)((n + 255) / 256);
    CHK(hipMalloc(&d_q, (size_t)nq * dim * 4)); CHK(hipMalloc(&cand_d, (size_t)nq * nblk * 4 * 4)); CHK(hipMalloc(&cand_i, (size_t)nq * nblk * 4 * 4));
    hipLaunchKernelGGL(knn_queries_kernel, dim3((nq * dim + 255) / 256, 1), dim3(256), 0, 0, d_cv, ld, (long long)dim * ld, dim, first_raw, nq, d_q);
    KnnP kp{}; kp.indexT = nullptr; kp.index = d_index; kp.n = (int)n; kp.dim = dim; kp.nblk = nblk; kp.v_stride = dim; kp.d_stride = 1; kp.q = d_q; kp.nq = nq;
    kp.cand_d = cand_d; kp.cand_i = cand_i; kp.overflow = nullptr; kp.q_bs = (long long)nq * dim; kp.cand_bs = (long long)nq * nblk * 4;
    hipLaunchKernelGGL(knn_scan_kernel, dim3(nblk, 1), dim3(256), 0, 0, kp);
    KnnBlendP bp{}; bp.cand_d = cand_d; bp.cand_i = cand_i; bp.nblk = nblk; bp.nq = nq; bp.index = d_index; bp.dim = dim; bp.q = d_q; bp.skip_head = skip_head; bp.T = T; bp.R = R;
    bp.first_raw = first_raw; bp.rate = 0.75f; bp.phone = d_phone2; bp.ph_cs = 32; bp.ph_bs = (long long)dim * 32; bp.out_idx = d_idx2; bp.out_dist = d_dist2; bp.overflow = nullptr;
    hipLaunchKernelGGL(knn_merge_blend_kernel, dim3(nq, 1), dim3(256), 0, 0, bp);
    CHK(hipDeviceSynchronize());
    // the one-launch kernel
    KnnFusedP fp{}; fp.indexF = d_indexF; fp.index = d_index; fp.ynorm = d_ynorm; fp.n = (int)n; fp.dim = dim; fp.cv = d_cv; fp.cv_cs = ld; fp.cv_bs = (long long)dim * ld;
    fp.first_raw = first_raw; fp.nq = nq; fp.q0 = 0; fp.skip_head = skip_head; fp.T = T; fp.R = R; fp.rate = 0.75f; fp.phone = d_phone; fp.ph_cs = 32; fp.ph_bs = (long long)dim * 32;
    fp.out_idx = d_idx; fp.out_dist = d_dist; fp.status = d_status; fp.status_stride = 1;
    const unsigned Gx = (unsigned)std::min<long long>((n + 15) / 16, G);
    CHK(hipMalloc(&fp.lists, (size_t)16 * Gx * 4 * 8)); CHK(hipMalloc(&fp.ticket, 8)); CHK(hipMemset(fp.ticket, 0, 8));
    CHK(hipMalloc(&fp.stamps, (size_t)Gx * 16 * 8)); CHK(hipMemset(fp.stamps, 0, (size_t)Gx * 16 * 8));
    const size_t lds = knn_fused_lds_floats(dim, nq, (int)Gx) * 4;
    CHK(hipFuncSetAttribute((const void *)knn_scan_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    float *d_flush; const size_t fl = (size_t)1 << 30; CHK(hipMalloc(&d_flush, fl));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const int reps = 20; double sum = 0, mn = 1e9;
    for (int r = 0; r < reps + 2; r++) {
        CHK(hipMemsetAsync(d_flush, r, fl, 0));
        hipExtLaunchKernelGGL(knn_scan_select_kernel, dim3(Gx, 1), dim3(256), (uint32_t)lds, 0, a, b, 0, fp);
        CHK(hipDeviceSynchronize());
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        if (r >= 2) { sum += ms; mn = std::min(mn, (double)ms); }
    }
    printf("knn_scan_select_kernel: G %u, n %lld: mean %.1f us, min %.1f us (%.2f TB/s of index at the mean)\n", Gx, n, sum / reps * 1e3, mn * 1e3, (double)n * dim * 4 / (sum / reps * 1e-3) * 1e-12);
    std::vector<long long> st((size_t)Gx * 16); CHK(hipMemcpy(st.data(), fp.stamps, st.size() * 8, hipMemcpyDeviceToHost));
    long long t0 = st[0]; for (unsigned g = 0; g < Gx; g++) t0 = std::min(t0, st[g * 16]);
    auto us = [&](long long t) { return (t - t0) * 0.01; };       // wall_clock64: 100 MHz
    double s1 = 0, s2 = 0, s3 = 0, l0 = 0, l2 = 0, l3 = 0;
    for (unsigned g = 0; g < Gx; g++) { s1 += us(st[g * 16 + 1]) - us(st[g * 16]); s2 += us(st[g * 16 + 2]) - us(st[g * 16 + 1]); s3 += us(st[g * 16 + 3]) - us(st[g * 16 + 2]);
                                        l0 = std::max(l0, us(st[g * 16])); l2 = std::max(l2, us(st[g * 16 + 2])); l3 = std::max(l3, us(st[g * 16 + 3])); }
    { int h[16] = {0}; for (unsigned g = 0; g < Gx; g++) { int k = (int)(us(st[g * 16 + 2]) / 10.0); h[k < 15 ? k : 15]++; }
      printf("scan end, workgroups per 10 us bin:"); for (int k = 0; k < 16; k++) printf(" %d", h[k]); printf("\n"); }
    printf("workgroups: start <= %.1f us; gather %.2f us, scan %.2f us, lists+ticket %.2f us on average; last scan end %.1f us, last ticket %.1f us\n", l0, s1 / Gx, s2 / Gx, s3 / Gx, l2, l3);
    for (unsigned g = 0; g < Gx; g++)
        if (st[g * 16 + 7] > st[g * 16 + 3] && st[g * 16 + 4] > st[g * 16 + 3])
            printf("  selector wg %4u: ticket %.1f  all-arrived %.1f  lists %.1f  wave-top4 %.1f  a4 %.1f  candidates %.1f  staged %.1f  exact-done %.1f  final-four %.1f  blended %.1f  end %.1f us\n", g, us(st[g * 16 + 3]), us(st[g * 16 + 4]), us(st[g * 16 + 12]), us(st[g * 16 + 13]),
                   us(st[g * 16 + 5]), us(st[g * 16 + 10]), us(st[g * 16 + 11]), us(st[g * 16 + 6]), us(st[g * 16 + 8]), us(st[g * 16 + 9]), us(st[g * 16 + 7])), printf("     chain: %lld shader cycles\n", st[g * 16 + 1]);
    std::vector<int> i1(R * 4), i2(R * 4); std::vector<float> x1(R * 4), x2(R * 4), p1((size_t)dim * 32), p2((size_t)dim * 32);
    CHK(hipMemcpy(i1.data(), d_idx, R * 16, hipMemcpyDeviceToHost)); CHK(hipMemcpy(i2.data(), d_idx2, R * 16, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(x1.data(), d_dist, R * 16, hipMemcpyDeviceToHost)); CHK(hipMemcpy(x2.data(), d_dist2, R * 16, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(p1.data(), d_phone, p1.size() * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(p2.data(), d_phone2, p2.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < R * 4; i++) bad += (i1[i] != i2[i]) || (x1[i] != x2[i]);
    size_t badp = 0; for (size_t i = 0; i < p1.size(); i++) badp += p1[i] != p2[i];
    printf("hits / distances differing from the exhaustive definition: %d of %d; blended values differing: %zu of %zu\n", bad, R * 4, badp, p1.size());
    return bad || badp;
}
