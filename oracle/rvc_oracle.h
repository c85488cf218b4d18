/*
 * rvc_oracle.h -- CPU restatement of the RVC per-chunk hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (obs_rvc_amd/csrc) never links, imports or executes it.
 *
 * The in-tree Rust code of the reference is restated function by function (each function
 * in rvc_oracle.c cites the file:line it follows).  The three neural networks are opaque
 * ONNX files in the reference (rvc/src/rvc.rs:92,195; rvc/src/f0/rmvpe.rs:235) that are not
 * part of /root/reference; they are restated from the public upstream architectures
 *  and PARITY FOR THOSE THREE GRAPHS IS UNPINNED against the
 * reference itself (no ONNX Runtime, no model files, no Rust toolchain in this image).
 * What is pinned: stft / pad_reflect against the reference's own inline KATs
 * (rmvpe.rs:278-308), the 2T+1 feature duplication against rvc/src/tests/feats.npy, and the
 * dense layers against independent torch-CPU / HF-HuBERT implementations (tests/).
 */
#ifndef RVC_ORACLE_H
#define RVC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* status codes shared with include/rvc_mi355x.h (rvc-common/src/errors.rs:2-8) */
enum { ORA_OK = 0, ORA_MODEL_NOT_LOADED = 1, ORA_CONTENTVEC_NOT_LOADED = 2, ORA_F0_NOT_LOADED = 3,
       ORA_BACKEND = 4, ORA_SHAPE = 5, ORA_PANIC = 6 };

typedef struct ora_engine ora_engine;

/* --- RvcInfer mirror (rvc/src/rvc.rs:30-220) --- */
ora_engine *ora_new(const char *data_path);
void ora_free(ora_engine *e);
int ora_load_contentvec(ora_engine *e, int version);        /* 1 = v1 (256, layer 9), 2 = v2 (768, layer 12) */
int ora_load_model(ora_engine *e, const char *model_path);
int ora_load_f0(ora_engine *e, int algorithm);              /* 1 = rmvpe */
void ora_unload_model(ora_engine *e);
int ora_hubert(ora_engine *e, const float *in, size_t n, float *out, size_t cap, size_t dims[3]);           /* (1,C,T) */
int ora_extract_feature(ora_engine *e, const float *in, size_t n, float *out, size_t cap, size_t dims[3]);  /* (1,2T+1,C) */
int ora_pitch(ora_engine *e, const float *in, size_t n, int pitch_shift, size_t sample_frame_16k,
              float *out, size_t cap, size_t *out_len);
int ora_infer(ora_engine *e, const float *in, size_t n, size_t sample_frame_16k, int has_pitch_shift,
              int pitch_shift, uint32_t skip_head, uint32_t return_length, float *out, size_t cap, size_t *out_len);
const char *ora_last_error(ora_engine *e);

/* --- extensions that the reference leaves as TODO / bakes into its ONNX export --- */
int ora_load_index(ora_engine *e, const float *vecs, size_t n, size_t dim);  /* flat-L2 index (rvc.rs:159 TODO) */
void ora_set_index_rate(ora_engine *e, float rate);
void ora_set_noise_seed(ora_engine *e, uint32_t seed, uint32_t stream_id);
void ora_reset_state(ora_engine *e);                         /* zero pitch cache + chunk counter */
void ora_get_pitch_cache(ora_engine *e, float *out1024);
/* kNN hits of the last infer: idx[return_length][4] (int32) and squared distances */
int ora_get_knn(ora_engine *e, int32_t *idx, float *dist, size_t cap_rows, size_t *rows);
/* named intermediate of the last call, contiguous row-major; returns ORA_OK or ORA_SHAPE if unknown */
int ora_get_tap(ora_engine *e, const char *name, const float **data, size_t *n);
void ora_enable_taps(ora_engine *e, int on);
void ora_set_threads(int n);

/* --- stage-level functions (rvc/src/f0/rmvpe.rs, rvc/src/f0/mod.rs) --- */
void ora_hann_periodic(size_t n, float *out);                                           /* rmvpe.rs:33-37 */
void ora_pad_reflect(const float *in, size_t n, size_t pad, float *out);                /* rmvpe.rs:47-68 */
/* rmvpe.rs:80-116: magnitude (fft_size/2+1, T) row-major, T = 1 + n/hop; returns T */
size_t ora_stft(const float *sig, size_t n, size_t fft_size, size_t hop, const float *window, int center, float *out);
void ora_mel_filterbank(double sr, size_t n_fft, size_t n_mels, double fmin, double fmax, float *out); /* rmvpe.rs:146-148 */
size_t ora_mel_extract(const float *sig, size_t n, float *out /* (128, T) */);           /* rmvpe.rs:159-205 */
/* rmvpe.rs:118-133 + 243-248: salience (T,360) -> f0 Hz; returns ORA_PANIC if the reference would index out of bounds */
int ora_decode(const float *salience, size_t T, float threshold, float *f0);
void ora_get_f0_post(const float *f0, size_t n, int32_t *coarse);                        /* f0/mod.rs:7-12 */
float ora_uppower(int pitch_shift);                                                       /* rvc.rs:121 */
size_t ora_f0_extractor_frame(size_t sample_frame_16k);                                   /* rmvpe.rs:256 */
/* flat-L2 top-k with sequential-fmaf distances, ascending (distance, index) */
void ora_knn_search(const float *index, size_t n, size_t dim, const float *q, size_t nq, int k, int32_t *idx, float *dist);
/* Philox4x32-10 + Box-Muller normal stream shared with the GPU path */
void ora_philox_normal(uint32_t seed, uint32_t stream, uint32_t chunk, uint32_t purpose, size_t n, float *out);

/* --- caller-side post-processing (obs-rvc/src/rt_utils.rs, obs-rvc/src/lib.rs:758-794) --- */
size_t ora_rms(const float *y, size_t n, size_t frame_length, size_t hop_length, float *out);         /* rt_utils.rs:94-103; returns #frames */
void ora_lerp_align_corners(const float *in, size_t n_in, size_t size, float *out);                    /* rt_utils.rs:105-117 */
void ora_envelop_mixing(const float *input, float *output, size_t output_len, size_t sample_rate, double mix_rate);  /* rt_utils.rs:119-132 */
size_t ora_sola_offset(const float *input_buffer, const float *sola_buffer, size_t buffer_frame_size, size_t search_frame_size); /* rt_utils.rs:60-90 */
/* lib.rs:768-794: offset search, sin^2 crossfade with the previous tail, save the new tail, return the frame */
size_t ora_sola_step(float *output, size_t output_len, float *sola_buffer, size_t sola_len, size_t search, size_t frame, float *frame_out);

#ifdef __cplusplus
}
#endif
#endif
