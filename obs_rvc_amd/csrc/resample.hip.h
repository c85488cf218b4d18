// resample.hip.h -- the plugin's two sample-rate converters, included by engine.hip.
//
// Reference: obs-rvc/src/lib.rs:236-242 builds `rubato::FftFixedInOut::<f32>::new(rate_in, rate_out, chunk, 1)` twice
// (host rate -> 16 kHz in front of RvcInfer::infer, model rate -> host rate behind it) and calls `process` /
// `process_into_buffer` once per chunk (lib.rs:675, 747-749).  rubato 0.15.0 (Cargo.lock:1223) is not part of /root/reference;
// oracle/resample_oracle.py restates its synchronous-FFT algorithm (parity unpinned, see there).
//
// MI355X formulation.  One `process` call is a fixed linear map of the chunk: zero-pad to Lin = 2*fft_in, real FFT, multiply by
// the filter spectrum F, keep new_len bins, inverse real FFT of length Lout = 2*fft_out.  Written in the time domain,
//     y[m] = sum_{n < fft_in} x[n] * c(m*fft_in/fft_out - n),   c(u) = sum_{k < new_len} w_k Re(F[k] e^{2 pi i k u / Lin}),
// w_0 = 1, w_k = 2 (the output Nyquist bin is never kept).  With fft_in/fft_out = P/Q in lowest terms, output m only touches
// c at u = (mP mod Q)/Q + integer: a POLYPHASE filter with Q rows of Lin taps, row[r][i] = c((r + i*Q)/Q), and
//     y[m] = sum_n x[n] * row[mP mod Q][(floor(mP/Q) - n) mod Lin].
// The table is built once per converter ON THE GPU in fp64 (two brute-force DFTs with exact integer phase reduction: no FFT
// library, ~1e8..3e10 terms); per chunk one kernel runs the polyphase sum with the row (up to 150 KiB; longer chunks -- the plugin allows
// 1.5 s -- read it from L2) and the chunk resident in the 160 KiB LDS of each CU: no general-length (2^a 3^b 5^c 7^d) FFT on the per-chunk path, one launch, LDS-bandwidth bound
// (fft_in * Lout MACs: 50 M for 48k->16k at 160 ms).  Same result as the FFT form up to fp32 rounding (tests: 2e-5 abs).
#pragma once

namespace rvc {

// F[k] = sum_n h[n] e^{-2 pi i k n / Lin}   (h = filter taps, already scaled by 1/Lin), k < new_len
__global__ void resample_filter_spectrum_kernel(const float *h, int fft_in, int Lin, int new_len, double *Fr, double *Fi)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= new_len) return;
    double sr = 0.0, si = 0.0;
    const double w = -2.0 * 3.14159265358979323846 / (double)Lin;
    for (int n = 0; n < fft_in; n++) {
        const long long ph = ((long long)k * n) % Lin;
        double s, c;
        sincos(w * (double)ph, &s, &c);
        sr += (double)h[n] * c; si += (double)h[n] * s;
    }
    Fr[k] = sr; Fi[k] = si;
}

// row[r][i] = c((r + i*Q)/Q) = sum_k w_k Re(F[k] e^{2 pi i k j / (Lin*Q)}),  j = r + i*Q
__global__ void resample_table_kernel(const double *Fr, const double *Fi, int new_len, int Lin, int Q, float *table)
{
    const long long tot = (long long)Lin * Q;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;    // idx = r*Lin + i
    if (idx >= tot) return;
    const int r = (int)(idx / Lin), i = (int)(idx - (long long)r * Lin);
    const long long j = (long long)r + (long long)i * Q;
    const double w = 2.0 * 3.14159265358979323846 / (double)tot;
    double ds, dc;
    sincos(w * (double)j, &ds, &dc);            // rotation per k step
    double acc = 0.0;
    for (int k0 = 0; k0 < new_len; k0 += 32) {
        // exact restart of the phasor every 32 bins: phase index (k0 * j) mod (Lin*Q) in integers
        const long long ph = (long long)(((unsigned long long)k0 * (unsigned long long)j) % (unsigned long long)tot);
        double s, c;
        sincos(w * (double)ph, &s, &c);
        const int k1 = k0 + 32 < new_len ? k0 + 32 : new_len;
        for (int k = k0; k < k1; k++) {
            const double term = Fr[k] * c - Fi[k] * s;
            acc += k == 0 ? term : 2.0 * term;
            const double cn = c * dc - s * ds, sn = s * dc + c * ds;
            c = cn; s = sn;
        }
    }
    table[idx] = (float)acc;
}

struct ResampleP {
    const float *x;          // [fft_in] chunk
    const float *table;      // [Q][Lin]
    const float *ov_old;     // [fft_out]
    float *ov_new;           // [fft_out]
    float *out;              // [fft_out]
    int fft_in, fft_out, Lin, P, Q, splits, x_in_lds, row_in_lds;
    long long x_bs, out_bs;     // stream strides (blockIdx.z = stream); the overlap buffers are [streams][fft_out]
};

// grid = (splits, Q, streams): workgroup (s, a) owns the outputs m = a + t*Q, t in its share of [0, ceil((Lout - a)/Q)); they all use
// polyphase row (a*P) mod Q.  Row and chunk live in LDS; one wave per group of 4 outputs (they share the x reads), lanes along n.
__global__ __launch_bounds__(1024) void resample_polyphase_kernel(ResampleP p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *row = smem, *xs = smem + p.Lin;
    const int a = blockIdx.y, Lout = 2 * p.fft_out;
    const int r = (int)(((long long)a * p.P) % p.Q);
    const float *rsrc = p.table + (long long)r * p.Lin;
    const float *xg = p.x + blockIdx.z * p.x_bs, *ovo = p.ov_old + (long long)blockIdx.z * p.fft_out;
    float *ovn = p.ov_new + (long long)blockIdx.z * p.fft_out, *og = p.out + blockIdx.z * p.out_bs;
    if (p.row_in_lds) {
        // staging in batches of 4 independent 16-byte loads per thread (latency: one round trip per batch, not per element)
        const int n4 = p.Lin / 4, step = (int)blockDim.x;
        for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * step) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) if (i0 + u * step < n4) v[u] = reinterpret_cast<const f32x4 *>(rsrc)[i0 + u * step];
#pragma unroll
            for (int u = 0; u < 4; u++) if (i0 + u * step < n4) reinterpret_cast<f32x4 *>(row)[i0 + u * step] = v[u];
        }
        for (int i = n4 * 4 + threadIdx.x; i < p.Lin; i += step) row[i] = rsrc[i];
    }
    if (p.x_in_lds) { float *xd = p.row_in_lds ? xs : smem; for (int i = threadIdx.x; i < p.fft_in; i += (int)blockDim.x) xd[i] = xg[i]; }
    __syncthreads();
    // long chunks (plugin sample_length up to 1.5 s): the polyphase row stays in global memory / L2, only the chunk sits in LDS
    const float *xv = p.x_in_lds ? (p.row_in_lds ? xs : smem) : xg;
    const float *rowp = p.row_in_lds ? row : rsrc;
    const int n_t = a < Lout ? (Lout - a + p.Q - 1) / p.Q : 0;               // outputs of this residue class
    const int per = (n_t + p.splits - 1) / p.splits;
    const int t0 = blockIdx.x * per, t1 = (t0 + per < n_t) ? t0 + per : n_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int tb = t0 + wave * 4; tb < t1; tb += nw * 4) {
        int s[4]; float acc[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const long long m = (long long)a + (long long)(tb + q < t1 ? tb + q : t1 - 1) * p.Q;
            s[q] = (int)((m * p.P) / p.Q);        // < Lin
            acc[q] = 0.f;
        }
        // row index (s[q] - n) mod Lin: no wrap while n <= s[0] (s ascends with q), always wrapped once n > s[3]; only the few
        // taps in between need the generic form -- the two long segments run with plain descending addresses
        const int nA = s[0] + 1 < p.fft_in ? s[0] + 1 : p.fft_in, nB = s[3] + 1 < p.fft_in ? s[3] + 1 : p.fft_in;
        const float *r0 = rowp + s[0], *r1 = rowp + s[1], *r2 = rowp + s[2], *r3 = rowp + s[3];
        int n = lane;
#pragma unroll 4
        for (; n < nA; n += 64) {
            const float xvn = xv[n];
            acc[0] += xvn * r0[-n]; acc[1] += xvn * r1[-n]; acc[2] += xvn * r2[-n]; acc[3] += xvn * r3[-n];
        }
        for (; n < nB; n += 64) {
            const float xvn = xv[n];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int i = s[q] - n; i += i < 0 ? p.Lin : 0;
                acc[q] += xvn * rowp[i];
            }
        }
        r0 += p.Lin; r1 += p.Lin; r2 += p.Lin; r3 += p.Lin;
#pragma unroll 4
        for (; n < p.fft_in; n += 64) {
            const float xvn = xv[n];
            acc[0] += xvn * r0[-n]; acc[1] += xvn * r1[-n]; acc[2] += xvn * r2[-n]; acc[3] += xvn * r3[-n];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float v = wave_sum(acc[q]);
            const int t = tb + q;
            if (lane == 0 && t < t1) {
                const int m = a + t * p.Q;
                if (m < p.fft_out) og[m] = v + ovo[m];
                else ovn[m - p.fft_out] = v;
            }
        }
    }
}

}  // namespace rvc

using namespace rvc;

struct rvc_resampler {
    rvc_engine *e = nullptr;
    int rate_in = 0, rate_out = 0, fft_in = 0, fft_out = 0, P = 1, Q = 1, Lin = 0, nb = 1;   // nb = streams converted per call
    float *d_table = nullptr, *d_ov[2] = {nullptr, nullptr}, *d_x = nullptr, *d_out = nullptr;
    int parity = 0;
};

static long long gcd_ll(long long a, long long b) { while (b) { long long t = a % b; a = b; b = t; } return a; }

extern "C" {

// FftFixedInOut::<f32>::new(sample_rate_input, sample_rate_output, chunk_size_in, 1)   (obs-rvc/src/lib.rs:236-242)
static rvc_status resampler_create_n(rvc_engine *e, size_t rate_in, size_t rate_out, size_t chunk_size_in, int nb, rvc_resampler **out)
{
    if (out) *out = nullptr;
    return guarded(e, [&]() {
        if (!out || rate_in == 0 || rate_out == 0 || chunk_size_in == 0 || rate_in > (1u << 22) || rate_out > (1u << 22) || chunk_size_in > (1u << 20))
            throw ShapeError("resampler: bad rates / chunk size");
        std::unique_ptr<rvc_resampler> r(new rvc_resampler());
        r->e = e; r->rate_in = (int)rate_in; r->rate_out = (int)rate_out; r->nb = nb;
        const long long g = gcd_ll((long long)rate_in, (long long)rate_out);
        const int min_in = (int)(rate_in / g);
        // fft_chunks = ceil(chunk_size_in as f32 / min_chunk_in as f32)
        const int chunks = (int)ceilf((float)chunk_size_in / (float)min_in);
        r->fft_in = (int)((long long)chunks * (long long)rate_in / g); r->fft_out = (int)((long long)chunks * (long long)rate_out / g);
        r->P = min_in; r->Q = (int)(rate_out / g); r->Lin = 2 * r->fft_in;
        const int fi = r->fft_in, fo = r->fft_out, Lin = r->Lin;
        if (fi > (1 << 18)) throw ShapeError("resampler: chunk too long");
        if ((long long)Lin * r->Q > (1ll << 28)) throw ShapeError("resampler: polyphase table too large (rates with a tiny common divisor)");
        // filter taps in f32 as the crate computes them: window^2 * sinc, normalised to unit sum, then / (2 fft_in)
        float cutoff = powf(0.4f, 16.0f / (float)fi);
        if (fi > fo) cutoff = cutoff * (float)fo / (float)fi;
        std::vector<float> h(fi);
        {
            const float pi = 3.14159265358979323846f, np = (float)fi;
            float sum = 0.f;
            for (int x = 0; x < fi; x++) {
                const float xf = (float)x;
                float w = 0.35875f - 0.48829f * cosf(2.f * pi * xf / np) + 0.14128f * cosf(4.f * pi * xf / np) - 0.01168f * cosf(6.f * pi * xf / np);
                w = w * w;
                const float t = (xf - (float)(fi / 2)) * cutoff;
                const float sv = t == 0.f ? 1.f : sinf(t * pi) / (t * pi);
                h[x] = w * sv; sum += h[x];
            }
            for (int x = 0; x < fi; x++) h[x] = h[x] / sum / (float)(2 * fi);
        }
        const int new_len = fi < fo ? fi + 1 : fo;
        float *d_h = upload_f(h);
        double *d_Fr, *d_Fi;
        HIPCHK(hipMalloc(&d_Fr, (size_t)new_len * 8)); HIPCHK(hipMalloc(&d_Fi, (size_t)new_len * 8));
        HIPCHK(hipMalloc(&r->d_table, (size_t)Lin * r->Q * sizeof(float)));
        hipLaunchKernelGGL(resample_filter_spectrum_kernel, dim3((new_len + 63) / 64), dim3(64), 0, e->stream, d_h, fi, Lin, new_len, d_Fr, d_Fi);
        const long long tot = (long long)Lin * r->Q;
        hipLaunchKernelGGL(resample_table_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, e->stream, d_Fr, d_Fi, new_len, Lin, r->Q, r->d_table);
        for (int i = 0; i < 2; i++) { HIPCHK(hipMalloc(&r->d_ov[i], (size_t)nb * fo * 4)); HIPCHK(hipMemsetAsync(r->d_ov[i], 0, (size_t)nb * fo * 4, e->stream)); }
        HIPCHK(hipMalloc(&r->d_x, (size_t)nb * fi * 4)); HIPCHK(hipMalloc(&r->d_out, (size_t)nb * fo * 4));
        HIPCHK(hipStreamSynchronize(e->stream));
        HIPCHK(hipGetLastError());
        wfree(d_h); (void)hipFree(d_Fr); (void)hipFree(d_Fi);
        HIPCHK(hipFuncSetAttribute((const void *)resample_polyphase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        *out = r.release();
        return RVC_OK;
    });
}

rvc_status rvc_resampler_create(rvc_engine *e, size_t rate_in, size_t rate_out, size_t chunk_size_in, rvc_resampler **out)
{
    return resampler_create_n(e, rate_in, rate_out, chunk_size_in, 1, out);
}

void rvc_resampler_destroy(rvc_resampler *r)
{
    if (!r) return;
    (void)hipSetDevice(r->e->device);
    (void)hipFree(r->d_table); (void)hipFree(r->d_ov[0]); (void)hipFree(r->d_ov[1]); (void)hipFree(r->d_x); (void)hipFree(r->d_out);
    delete r;
}

size_t rvc_resampler_input_frames_next(rvc_resampler *r) { return r ? (size_t)r->fft_in : 0; }    // Resampler::input_frames_next
size_t rvc_resampler_output_frames_max(rvc_resampler *r) { return r ? (size_t)r->fft_out : 0; }    // Resampler::output_frames_max (lib.rs:244)

// Resampler::reset: forget the saved overlap
void rvc_resampler_reset(rvc_resampler *r)
{
    if (!r) return;
    (void)guarded(r->e, [&]() {
        for (int i = 0; i < 2; i++) HIPCHK(hipMemsetAsync(r->d_ov[i], 0, (size_t)r->nb * r->fft_out * 4, r->e->stream));
        HIPCHK(hipStreamSynchronize(r->e->stream));
        return RVC_OK;
    });
}

// queue one chunk (of every stream) on the engine's stream; d_in / d_out are device pointers, stream b at d_in + b*in_bs / d_out + b*out_bs
static void resampler_launch(rvc_resampler *r, const float *d_in, float *d_out, long long in_bs = 0, long long out_bs = 0)
{
    ResampleP p{};
    p.x = d_in; p.table = r->d_table; p.ov_old = r->d_ov[r->parity]; p.ov_new = r->d_ov[r->parity ^ 1]; p.out = d_out;
    p.fft_in = r->fft_in; p.fft_out = r->fft_out; p.Lin = r->Lin; p.P = r->P; p.Q = r->Q;
    p.x_bs = in_bs; p.out_bs = out_bs;
    p.row_in_lds = ((size_t)r->Lin * sizeof(float) <= 150 * 1024) ? 1 : 0;
    p.x_in_lds = ((size_t)((p.row_in_lds ? r->Lin : 0) + r->fft_in) * sizeof(float) <= 152 * 1024) ? 1 : 0;
    const int Lout = 2 * r->fft_out, per_class = (Lout + r->Q - 1) / r->Q;
    // 32..64 outputs (8..16 waves x 4) per workgroup: the 100+ KiB of LDS staging is amortised and the grid still covers the chip
    const int per_target = r->Q == 1 ? 32 : 64;      // measured: 14 us (48k->16k), 40 us (48k->48k), 13 us (44.1k->16k) per call
    int splits = std::max(1, (per_class + per_target - 1) / per_target);
    while (splits > 1 && (long long)splits * r->Q > 4096) splits = (splits + 1) / 2;
    p.splits = splits;
    const int per_wg = (per_class + splits - 1) / splits;
    const int threads = std::min(1024, std::max(64, ((per_wg + 3) / 4) * 64));
    const size_t lds = (size_t)((p.row_in_lds ? r->Lin : 0) + (p.x_in_lds ? r->fft_in : 0)) * sizeof(float);
    hipLaunchKernelGGL(resample_polyphase_kernel, dim3(splits, r->Q, r->nb), dim3(threads), lds, r->e->stream, p);
    r->parity ^= 1;
}

// Resampler::process / process_into_buffer for one channel (lib.rs:675, 747-749): exactly input_frames_next() frames in,
// output_frames_max() frames out.  A wrong input length is rubato's ResampleError::WrongNumberOfInputFrames (the plugin
// panics on it): RVC_SHAPE here.
rvc_status rvc_resampler_process(rvc_resampler *r, const float *in, size_t n_in, float *out, size_t cap, size_t *n_out)
{
    if (!r) return RVC_BACKEND;
    return guarded(r->e, [&]() {
        if (n_out) *n_out = (size_t)r->fft_out;
        if (n_in != (size_t)r->fft_in) throw ShapeError("resampler: wrong number of input frames");
        if (!in || !out || cap < (size_t)r->fft_out) throw ShapeError("resampler: output buffer too small");
        hipStream_t s = r->e->stream;
        HIPCHK(hipMemcpyAsync(r->d_x, in, n_in * 4, hipMemcpyHostToDevice, s));
        resampler_launch(r, r->d_x, r->d_out);
        HIPCHK(hipMemcpyAsync(out, r->d_out, (size_t)r->fft_out * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipGetLastError());
        return RVC_OK;
    });
}

// device-resident variant: d_in [input_frames_next] and d_out [output_frames_max] are HIP device pointers; no sync unless asked
rvc_status rvc_resampler_process_device(rvc_resampler *r, const void *d_in, void *d_out, int sync)
{
    if (!r) return RVC_BACKEND;
    return guarded(r->e, [&]() {
        if (!d_in || !d_out) throw ShapeError("resampler: null device pointer");
        resampler_launch(r, static_cast<const float *>(d_in), static_cast<float *>(d_out));
        if (sync) HIPCHK(hipStreamSynchronize(r->e->stream));
        HIPCHK(hipGetLastError());
        return RVC_OK;
    });
}

}  // extern "C"
