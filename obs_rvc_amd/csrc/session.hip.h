// session.hip.h -- the plugin's per-chunk state machine `process_one_frame` (obs-rvc/src/lib.rs:659-795, geometry from
// lib.rs:200-227) as ONE native call with every buffer resident in HBM (SURVEY.md section 8 rows f1-f3 chained), included by engine.hip.
//
// The reference moves each chunk host -> rubato -> pipe -> ONNX Runtime -> pipe -> rubato -> envelope -> SOLA -> host.  Here the
// host-rate ring, the 16 kHz ring, the model output, both resampler states and the SOLA tail live on the device and the chunk
// costs one H2D copy (the new samples), one D2H copy (the finished frame) and one stream synchronisation; everything else is
// queued on the engine's stream:
//   shift+append (ping-pong ring) -> polyphase downsampler -> 16 kHz ring update -> infer plan -> polyphase upsampler
//   -> RMS envelope mixing -> SOLA search / crossfade / tail save.
#pragma once

namespace rvc {

// out[i] = i < n - f ? in[i + f] : chunk[i - (n - f)]        (lib.rs:661-665 "move and append the last n samples")
// (blockIdx.y = stream; rings are [streams][n], chunks [streams][f])
__global__ void ring_shift_append_kernel(const float *in, float *out, int n, int f, const float *chunk)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    in += (long long)blockIdx.y * n; out += (long long)blockIdx.y * n; chunk += (long long)blockIdx.y * f;
    if (i < n) out[i] = i < n - f ? in[i + f] : chunk[i - (n - f)];
}

// 16 kHz ring (lib.rs:669-679): shift by f, then overwrite [copy_begin, n) with res[skip ..] (copy_begin = n - f - skip: the
// converter's output re-writes the `skip` samples before the new chunk as well)
__global__ void ring16_update_kernel(const float *in, float *out, int n, int f, const float *res, int skip, int copy_begin, long long res_bs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    in += (long long)blockIdx.y * n; out += (long long)blockIdx.y * n; res += blockIdx.y * res_bs;
    if (i < n) out[i] = i >= copy_begin ? res[skip + i - copy_begin] : in[i + f];
}

}  // namespace rvc

using namespace rvc;

struct rvc_session {
    rvc_engine *e = nullptr;
    // lib.rs:200-227
    int sample_rate = 0, zc = 0, sample_frame_size = 0, sample_frame_16k = 0, crossfade_frame_size = 0, sola_buffer_frame_size = 0,
        sola_search_frame_size = 0, extra_frame_size = 0, input_buffer_size = 0, input_buffer_16k_size = 0, model_return_length = 0,
        model_return_size = 0, skip_head = 0, model_rate = 0, up_out = 0;
    bool skip_inference = false;
    // per-stream settings (every stream is a caller of its own, obs-rvc/src/lib.rs:701-707, 174-185): pitch shift and RMS mix rate
    std::vector<int32_t> pitch_shift; std::vector<double> rms_mix_rate; float *d_mixpow = nullptr; bool mix_dirty = true;
    rvc_resampler *down = nullptr, *up = nullptr;
    float *d_in[2] = {nullptr, nullptr}, *d_in16[2] = {nullptr, nullptr};
    int par = 0, par16 = 0;
    float *d_chunk = nullptr, *d_down = nullptr, *d_model = nullptr, *d_up = nullptr, *d_rms = nullptr, *d_sola = nullptr, *d_frame = nullptr, *d_cor = nullptr;
    int *d_off = nullptr; int n_rms = 0;
    int B = 1;                     // streams (= the engine's stream count at creation); every buffer below has a leading [B] axis
    std::vector<int> h_off;
};

extern "C" {

void rvc_session_destroy(rvc_session *s)
{
    if (!s) return;
    (void)hipSetDevice(s->e->device);
    (void)hipStreamSynchronize(s->e->stream);
    rvc_resampler_destroy(s->down); rvc_resampler_destroy(s->up);
    for (float *p : {s->d_in[0], s->d_in[1], s->d_in16[0], s->d_in16[1], s->d_chunk, s->d_down, s->d_model, s->d_up, s->d_rms, s->d_sola, s->d_frame, s->d_cor}) (void)hipFree(p);
    (void)hipFree(s->d_off); (void)hipFree(s->d_mixpow);
    delete s;
}

// `create` / `update` of the filter (lib.rs:181-260): host sample rate, the three length settings (seconds), the synthesizer's output
// rate; skip_inference != 0 is the plugin's pass-through mode (lib.rs:224-227, 697-699).  The session covers every stream of the
// engine (rvc_set_streams before creating it): all rings, converter states and SOLA tails get a leading [streams] axis.
rvc_status rvc_session_create(rvc_engine *e, size_t sample_rate, double sample_length, double crossfade_length, double extra_inference_time,
                              size_t model_output_sample_rate, int32_t pitch_shift, double rms_mix_rate, int skip_inference, rvc_session **out)
{
    if (out) *out = nullptr;
    return guarded(e, [&]() {
        if (!out || sample_rate < 1000 || sample_rate % 100 != 0 || sample_rate > 384000 || model_output_sample_rate % 100 != 0 || model_output_sample_rate == 0)
            throw ShapeError("session: unsupported sample rate");
        // destroyed through rvc_session_destroy on every failure path (a throwing hipMalloc part-way through included): both converters
        // and every device buffer allocated so far are released
        std::unique_ptr<rvc_session, void (*)(rvc_session *)> sp(new rvc_session(), rvc_session_destroy);
        rvc_session *s = sp.get();
        s->e = e; s->B = e->n_streams; s->h_off.resize(s->B); s->sample_rate = (int)sample_rate; s->pitch_shift.assign(s->B, pitch_shift); s->rms_mix_rate.assign(s->B, rms_mix_rate); s->skip_inference = skip_inference != 0;
        const int zc = s->zc = (int)sample_rate / 100;                                                             // lib.rs:200
        const int sft = (int)llround(sample_length * (double)sample_rate / zc);                                     // lib.rs:202
        s->sample_frame_size = sft * zc; s->sample_frame_16k = sft * 160;                                           // lib.rs:203-205
        s->crossfade_frame_size = (int)llround(crossfade_length * (double)sample_rate / zc) * zc;                   // lib.rs:206-207
        s->sola_buffer_frame_size = std::min(s->crossfade_frame_size, 4 * zc);                                      // lib.rs:208
        s->sola_search_frame_size = zc;                                                                             // lib.rs:209
        s->extra_frame_size = (int)llround(extra_inference_time * (double)sample_rate / zc) * zc;                   // lib.rs:210-211
        s->input_buffer_size = s->extra_frame_size + s->crossfade_frame_size + s->sola_search_frame_size + s->sample_frame_size;   // lib.rs:213-214
        s->input_buffer_16k_size = 160 * s->input_buffer_size / zc;                                                 // lib.rs:217
        s->model_return_length = (s->sample_frame_size + s->sola_buffer_frame_size + s->sola_search_frame_size) / zc;   // lib.rs:220-221
        s->model_rate = s->skip_inference ? 16000 : (int)model_output_sample_rate;                                  // lib.rs:224-226
        s->model_return_size = s->model_return_length * (s->model_rate / 100);                                      // lib.rs:222,226
        s->skip_head = s->extra_frame_size / zc;                                                                    // lib.rs:694
        if (sft < 1 || s->sola_buffer_frame_size < 1 || s->sola_search_frame_size + 1 > 1024) throw ShapeError("session: unsupported length settings");
        // lib.rs:236-242
        rvc_status rc = resampler_create_n(e, sample_rate, 16000, (size_t)s->sample_frame_size + 2 * zc, s->B, &s->down);
        if (rc != RVC_OK) return rc;
        rc = resampler_create_n(e, (size_t)s->model_rate, sample_rate, (size_t)s->model_return_size, s->B, &s->up);
        if (rc != RVC_OK) return rc;
        s->up_out = s->up->fft_out;
        const bool ok = s->down->fft_in == s->sample_frame_size + 2 * zc && s->down->fft_out == s->sample_frame_16k + 320 && s->up->fft_in == s->model_return_size &&
                        s->up_out >= s->sola_buffer_frame_size + s->sola_search_frame_size + s->sample_frame_size;
        if (!ok) {   // rubato would return WrongNumberOfInputFrames on the first chunk and the plugin would panic (lib.rs:680-682)
            throw ShapeError("session: chunk sizes are not multiples of the resampling ratios");
        }
        const size_t NB = (size_t)s->B;
        auto dev = [&](float **p, size_t n) { n = std::max<size_t>(n, 4) * NB; HIPCHK(hipMalloc(p, n * 4)); HIPCHK(hipMemsetAsync(*p, 0, n * 4, e->stream)); };
        for (int i = 0; i < 2; i++) { dev(&s->d_in[i], s->input_buffer_size); dev(&s->d_in16[i], s->input_buffer_16k_size); }
        dev(&s->d_chunk, s->sample_frame_size); dev(&s->d_down, s->down->fft_out); dev(&s->d_model, s->model_return_size); dev(&s->d_up, s->up_out);
        const int frame = 4 * zc, hop = zc;
        s->n_rms = (s->up_out + 2 * (frame / 2) - frame) / hop + 1;
        dev(&s->d_rms, (size_t)2 * s->n_rms); dev(&s->d_sola, s->sola_buffer_frame_size); dev(&s->d_frame, s->sample_frame_size);
        dev(&s->d_cor, (size_t)s->sola_search_frame_size + 1);
        HIPCHK(hipMalloc(&s->d_off, 4 * NB));
        HIPCHK(hipMalloc(&s->d_mixpow, 4 * NB));
        HIPCHK(hipStreamSynchronize(e->stream));
        *out = sp.release();
        return RVC_OK;
    });
}

size_t rvc_session_frame_size(rvc_session *s) { return s ? (size_t)s->sample_frame_size : 0; }
void rvc_session_set_params(rvc_session *s, int32_t pitch_shift, double rms_mix_rate)
{
    if (!s) return;
    s->pitch_shift.assign(s->B, pitch_shift); s->rms_mix_rate.assign(s->B, rms_mix_rate); s->mix_dirty = true;
}
// one stream's settings (the other streams keep theirs)
rvc_status rvc_session_set_params_stream(rvc_session *s, int stream, int32_t pitch_shift, double rms_mix_rate)
{
    if (!s) return RVC_BACKEND;
    if (stream < 0 || stream >= s->B) { s->e->err = "session: stream out of range"; return RVC_SHAPE; }
    s->pitch_shift[stream] = pitch_shift; s->rms_mix_rate[stream] = rms_mix_rate; s->mix_dirty = true;
    return RVC_OK;
}

// geometry as the plugin derives it (tests): 0 sample_frame_size, 1 sample_frame_16k, 2 input_buffer_size, 3 input_buffer_16k_size,
// 4 model_return_length, 5 model_return_size, 6 skip_head, 7 sola_buffer_frame_size, 8 sola_search_frame_size, 9 extra_frame_size
void rvc_session_geometry(rvc_session *s, int32_t out[10])
{
    if (!s) return;
    const int v[10] = {s->sample_frame_size, s->sample_frame_16k, s->input_buffer_size, s->input_buffer_16k_size, s->model_return_length,
                       s->model_return_size, s->skip_head, s->sola_buffer_frame_size, s->sola_search_frame_size, s->extra_frame_size};
    for (int i = 0; i < 10; i++) out[i] = v[i];
}

// process_one_frame (lib.rs:659-795) for every stream of the engine: input_sample [B][n] with n = sample_frame_size samples at the host
// rate, output [B][cap_per_stream] (sample_frame_size samples each), sola_offset [B] (may be NULL).  B = 1 is the plugin's case.
rvc_status rvc_session_process(rvc_session *s, const float *input_sample, size_t n, float *output, size_t cap, size_t *sola_offset)
{
    if (!s) return RVC_BACKEND;
    rvc_engine *e = s->e;
    return guarded(e, [&]() {
        if (n != (size_t)s->sample_frame_size || cap < (size_t)s->sample_frame_size || !input_sample || !output) throw ShapeError("session: wrong chunk size");
        if (e->n_streams != s->B) throw ShapeError("session: the engine's stream count changed since the session was created");
        hipStream_t st = e->stream;
        const int T = 256, B = s->B;
        HIPCHK(hipMemcpyAsync(s->d_chunk, input_sample, (size_t)B * n * 4, hipMemcpyHostToDevice, st));
        // lib.rs:661-665
        hipLaunchKernelGGL(ring_shift_append_kernel, dim3((s->input_buffer_size + T - 1) / T, B), dim3(T), 0, st, s->d_in[s->par], s->d_in[s->par ^ 1],
                           s->input_buffer_size, s->sample_frame_size, s->d_chunk);
        s->par ^= 1;
        const float *ring = s->d_in[s->par];
        // lib.rs:669-683: the converter sees the new chunk plus the 2*zc samples before it; its first 160 outputs are dropped
        const int down_start = s->input_buffer_size - s->sample_frame_size - 2 * s->sample_rate / 100;
        resampler_launch(s->down, ring + down_start, s->d_down, s->input_buffer_size, s->down->fft_out);
        const int copy_begin = s->input_buffer_16k_size - (s->sample_frame_size / (s->sample_rate / 100) + 1) * 160;
        hipLaunchKernelGGL(ring16_update_kernel, dim3((s->input_buffer_16k_size + T - 1) / T, B), dim3(T), 0, st, s->d_in16[s->par16], s->d_in16[s->par16 ^ 1],
                           s->input_buffer_16k_size, s->sample_frame_16k, s->d_down, 160, copy_begin, (long long)s->down->fft_out);
        s->par16 ^= 1;
        const float *ring16 = s->d_in16[s->par16];
        // lib.rs:694-707
        if (s->skip_inference) {
            HIPCHK(hipMemcpy2DAsync(s->d_model, (size_t)s->model_return_size * 4, ring16 + (s->input_buffer_16k_size - s->model_return_size),
                                    (size_t)s->input_buffer_16k_size * 4, (size_t)s->model_return_size * 4, B, hipMemcpyDeviceToDevice, st));
        } else {
            size_t got = 0;
            // with retrieval on, the chunk is synchronised before the post-processing chain: a hand-off time-out of the one-launch retrieval is
            // recovered inside infer_common (the SOLA / envelope state behind it must only ever see the recovered chunk)
            // (only plans that CAN be recovered pay for that -- the one-launch retrieval of up to 11 streams, whose plan carries the fallback launches;
            //  many-stream plans search through the distance GEMM and hand nothing over inside a launch: their chain stays asynchronous)
            bool sync_infer = false;
            if (e->d_index && e->index_rate > 0.f) {
                Plan *peek = get_plan(e, 0, (size_t)s->input_buffer_16k_size, (size_t)s->sample_frame_16k, (uint32_t)s->skip_head, (uint32_t)s->model_return_length, 0);
                sync_infer = !peek->knn_fallback.empty();
            }
            rvc_status rc = infer_common(e, ring16, true, (size_t)s->input_buffer_16k_size, (size_t)s->sample_frame_16k, 0, (uint32_t)s->skip_head,
                                         (uint32_t)s->model_return_length, s->d_model, true, (size_t)s->model_return_size, &got, sync_infer, s->pitch_shift.data());
            if (rc != RVC_OK) return rc;
            if (got != (size_t)s->model_return_size) throw ShapeError("session: the loaded synthesizer's output rate does not match model_output_sample_rate");
        }
        // lib.rs:742-756
        resampler_launch(s->up, s->d_model, s->d_up, s->model_return_size, s->up_out);
        // lib.rs:758-765
        const long long up_bs = s->up_out;
        bool any_mix = false;
        for (int b = 0; b < B; b++) any_mix = any_mix || s->rms_mix_rate[b] < 1.0;
        if (any_mix) {
            if (s->mix_dirty) {      // exponent 1 - rate per stream; a stream at rate >= 1 gets 0: powf(x, 0) = 1 leaves it untouched (lib.rs:758)
                std::vector<float> mp(B);
                for (int b = 0; b < B; b++) mp[b] = s->rms_mix_rate[b] < 1.0 ? (float)(1.0 - s->rms_mix_rate[b]) : 0.f;
                HIPCHK(hipMemcpy(s->d_mixpow, mp.data(), 4 * (size_t)B, hipMemcpyHostToDevice));
                s->mix_dirty = false;
            }
            const int nn = s->up_out, frame = 4 * s->zc, hop = s->zc, nf = s->n_rms;
            hipLaunchKernelGGL(post_rms_kernel, dim3(nf, B), dim3(256), 0, st, ring + s->extra_frame_size, nn, frame, hop, s->d_rms, (long long)s->input_buffer_size, 2LL * nf);
            hipLaunchKernelGGL(post_rms_kernel, dim3(nf, B), dim3(256), 0, st, s->d_up, nn, frame, hop, s->d_rms + nf, up_bs, 2LL * nf);
            hipLaunchKernelGGL(post_mix_kernel, dim3((nn + 255) / 256, B), dim3(256), 0, st, s->d_up, nn, s->d_rms, nf, s->d_rms + nf, nf, 0.f, up_bs, 2LL * nf, s->d_mixpow);
        }
        // lib.rs:768-794
        const long long cor_bs = s->sola_search_frame_size + 1;
        hipLaunchKernelGGL(post_sola_corr_kernel, dim3((unsigned)(s->sola_search_frame_size + 4) / 4, B), dim3(256), 0, st, s->d_up, s->d_sola,
                           s->sola_buffer_frame_size, s->sola_search_frame_size, s->d_cor, up_bs, (long long)s->sola_buffer_frame_size, cor_bs);
        hipLaunchKernelGGL(post_sola_kernel, dim3(B), dim3(1024), 0, st, s->d_up, s->d_sola, s->sola_buffer_frame_size, s->sola_search_frame_size,
                           s->sample_frame_size, s->d_frame, s->d_off, s->d_cor, up_bs, (long long)s->sola_buffer_frame_size, (long long)s->sample_frame_size, cor_bs);
        HIPCHK(hipMemcpy2DAsync(output, cap * 4, s->d_frame, (size_t)s->sample_frame_size * 4, (size_t)s->sample_frame_size * 4, B, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(s->h_off.data(), s->d_off, 4 * (size_t)B, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGetLastError());
        if (sola_offset) for (int b = 0; b < B; b++) sola_offset[b] = (size_t)s->h_off[b];
        return s->skip_inference ? RVC_OK : final_status(e);
    });
}

}  // extern "C"
