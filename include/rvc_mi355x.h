/*
 * rvc_mi355x.h -- C ABI of the MI355X-native RVC streaming inference engine.
 *
 * Drop-in boundary for the `rvc` crate of RVC-Project/obs-rvc: every entry point below
 * replaces one method of `rvc::RvcInfer` (reference: rvc/src/rvc.rs:30-220) or one
 * variant of `rvc_common::errors::RvcInferError` (rvc-common/src/errors.rs:2-8).  Plain
 * pointers and sizes only; one handle = one engine bound to one GPU; a handle is NOT
 * re-entrant (the reference methods take `&mut self`, rvc.rs:133-134).
 *
 * Reference method                                  -> C entry point
 *   RvcInfer::new(data_path)            rvc.rs:30-44   rvc_create
 *   (Drop)                                             rvc_destroy
 *   load_contentvec(RvcModelVersion)    rvc.rs:46-54   rvc_load_contentvec
 *   load_model(model_path)              rvc.rs:56-60   rvc_load_model
 *   load_f0(PitchAlgorithm)             rvc.rs:62-75   rvc_load_f0
 *   unload_model()                      rvc.rs:77-79   rvc_unload_model
 *   hubert(input) -> (1,C,T)            rvc.rs:81-97   rvc_hubert
 *   extract_feature(input) -> (1,2T+1,C) rvc.rs:99-109 rvc_extract_feature
 *   pitch(input, shift, frame) -> f0    rvc.rs:111-131 rvc_pitch
 *   infer(input, frame, shift, skip_head, return_length) rvc.rs:133-220  rvc_infer
 *
 * File naming follows rvc/src/models.rs:58-61,72 with the native extension:
 *   <data>/contentvec/vec-{256,768}-layer-{9,12}.rvcw, <data>/f0/rmvpe.rvcw, <model>.rvcw
 * (a path ending in ".onnx" is mapped to its ".rvcw" sibling).
 */
#ifndef RVC_MI355X_H
#define RVC_MI355X_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* RvcInferError (rvc-common/src/errors.rs:2-8) + RVC_PANIC for inputs on which the reference
 * panics instead of returning an error (e.g. rmvpe.rs:124 out-of-bounds gather). */
typedef enum {
    RVC_OK = 0,
    RVC_MODEL_NOT_LOADED = 1,
    RVC_CONTENTVEC_NOT_LOADED = 2,
    RVC_F0_NOT_LOADED = 3,
    RVC_BACKEND = 4,          /* Ort(..) in the reference: load / device / kernel failure */
    RVC_SHAPE = 5,            /* NdarrayShapeError(..): bad sizes, output buffer too small */
    RVC_PANIC = 6
} rvc_status;

/* RvcModelVersion / PitchAlgorithm as the i64 conversions of rvc-common/src/enums.rs:32-48,96-110 */
#define RVC_VERSION_V1 1
#define RVC_VERSION_V2 2
#define RVC_PITCH_RMVPE 1

typedef struct rvc_engine rvc_engine;

/* Process environment: when the library is loaded it plants GPU_MAX_HW_QUEUES=16 unless the variable is already set (the HIP runtime
 * maps a process's streams onto 4 hardware queues by default; the engine runs 4 streams per chunk, and a second engine or an RCCL
 * communicator in the same process would share queues with them: +5-20 % per-chunk latency).  The runtime reads
 * the variable when IT initialises, so load this library before the first HIP call or set the variable yourself; set
 * RVC_NO_RUNTIME_DEFAULTS=1 to have the library leave the environment alone. */

/* RvcInfer::new.  device = HIP device ordinal (-1: current / LOCAL_RANK default 0). */
rvc_status rvc_create(const char *data_path, int device, rvc_engine **out);
void rvc_destroy(rvc_engine *e);
rvc_status rvc_load_contentvec(rvc_engine *e, int model_version);
rvc_status rvc_load_model(rvc_engine *e, const char *model_path);
rvc_status rvc_load_f0(rvc_engine *e, int pitch_algorithm);
void rvc_unload_model(rvc_engine *e);

/* Caller-owned buffers.  On RVC_SHAPE the required element count is still written to dims / out_len. */
rvc_status rvc_hubert(rvc_engine *e, const float *input, size_t n, float *out, size_t cap, size_t dims[3]);
rvc_status rvc_extract_feature(rvc_engine *e, const float *input, size_t n, float *out, size_t cap, size_t dims[3]);
rvc_status rvc_pitch(rvc_engine *e, const float *input, size_t n, int32_t pitch_shift, size_t sample_frame_16k_size,
                     float *out, size_t cap, size_t *out_len);
/* has_pitch_shift = 0 mirrors Option::None (rvc.rs:163) */
rvc_status rvc_infer(rvc_engine *e, const float *input, size_t n, size_t sample_frame_16k_size, int has_pitch_shift,
                     int32_t pitch_shift, uint32_t skip_head, uint32_t return_length, float *out, size_t cap, size_t *out_len);
const char *rvc_last_error_message(rvc_engine *e);

/* ---- capabilities the reference plumbs through but never implements / bakes into its export ---- */
/* flat-L2 retrieval index (index_path / index_rate settings, obs-rvc/src/lib.rs:78,81; rvc.rs:159 TODO) */
rvc_status rvc_load_index(rvc_engine *e, const float *vectors, size_t n, size_t dim);
rvc_status rvc_load_index_device(rvc_engine *e, const void *d_vectors, size_t n, size_t dim);  /* already in HBM (RCCL-broadcast) */
void rvc_set_index_rate(rvc_engine *e, float rate);
/* kNN hits of the last infer: idx[rows][4], squared distances; rows = return_length for stream 0, followed (stream-major) by the rows of as many
   further streams of a batched call as cap_rows holds whole.  After rvc_infer_batch_g (streams bucketed by geometry) no rows are reported: *rows = 0. */
rvc_status rvc_get_knn(rvc_engine *e, int32_t *idx, float *dist, size_t cap_rows, size_t *rows);
/* The one-launch retrieval (up to 11 streams) hands its partial lists from workgroup to workgroup inside the launch; if a workgroup does not arrive
   in time (a GPU shared with another process), the engine recomputes that chunk's retrieval through the exhaustive scan and the rest of the chunk,
   and the call still returns RVC_OK with the same hits.  This counts such chunks (0 on a GPU of one's own).  Unsynchronised calls (sync = 0) cannot
   be recomputed in order: there the time-out is reported by rvc_synchronize as RVC_BACKEND. */
long long rvc_retrieval_recoveries(rvc_engine *e);
/* the synthesizer's two noise inputs are explicit counter-based (Philox4x32-10) streams */
void rvc_set_noise_seed(rvc_engine *e, uint32_t seed, uint32_t stream_id);
void rvc_reset_state(rvc_engine *e);     /* zero the 1024-entry pitch cache and the chunk counter */

/* ---- multi-GPU (BASELINE configs[4]; no counterpart in the reference: one RvcInfer per process, rvc.rs:133-134) ---- */
/* Streams shard across GPUs with NO per-chunk collective: one process + one engine per GPU, stream s on rank s mod world.  The one
 * exchange step is at load: the shared retrieval index travels from rank 0 into every rank's HBM with ONE ncclBroadcast over
 * RCCL / xGMI.  Rank 0 calls rvc_rccl_unique_id and hands the 128 bytes to the other ranks by any host-side means (pipe, file,
 * TCP store); then EVERY rank calls rvc_index_broadcast with the same id.  Rank 0 passes the (n, dim) fp32 matrix (or NULL to
 * send the index its engine already holds); the other ranks pass vectors = NULL and n = dim = 0 (or the shape they expect, which
 * is then checked).  librccl is loaded lazily (dlopen; RVC_RCCL_LIB overrides the name): single-GPU use never touches it. */
#define RVC_RCCL_UNIQUE_ID_BYTES 128
rvc_status rvc_rccl_unique_id(void *id128);
rvc_status rvc_index_broadcast(rvc_engine *e, const void *unique_id128, int rank, int world, const float *vectors, size_t n, size_t dim);
/* RVC_OK when librccl can be loaded in this process (creates no communicator).  Hosts agree on it across ranks BEFORE calling
 * rvc_index_broadcast, so that a rank without the library cannot leave the others waiting in the communicator set-up.  Inside
 * rvc_index_broadcast nothing a single rank finds wrong with its own arguments makes it leave alone: rank 0 sends an empty header when
 * its index is unusable, every rank checks the header against what it expects, and ONE all-reduce of the verdicts precedes the payload
 * broadcast -- the ranks return the error together (tests/test_gpu_multi.py runs this with two ranks). */
rvc_status rvc_rccl_available(void);
/* the engine's last rvc_index_broadcast: ms[0] communicator set-up, ms[1] header + agreement + payload broadcast, ms[2] device-side
 * repack of the index (MFMA-fragment order + norms; the matrix never returns to the host); *ranks = ncclCommCount */
rvc_status rvc_index_broadcast_info(rvc_engine *e, double ms[3], int *ranks);

/* ---- many concurrent streams on one GPU (BASELINE configs 4-5) ---- */
/* The engine then holds n_streams independent stream states (pitch cache, noise counters) that share weights. */
rvc_status rvc_set_streams(rvc_engine *e, int n_streams);
/* input [n_streams][n], out [n_streams][cap_per_stream]; all streams use the same geometry */
rvc_status rvc_infer_batch(rvc_engine *e, const float *input, size_t n, size_t sample_frame_16k_size, int32_t pitch_shift,
                           uint32_t skip_head, uint32_t return_length, float *out, size_t cap_per_stream, size_t *out_len);
/* device-resident variant (input/out are HIP device pointers on the engine's device; no host copy, no sync
 * unless sync != 0).  Used by the throughput bench so that the timed region starts with inputs in HBM. */
rvc_status rvc_infer_device(rvc_engine *e, const void *d_input, size_t n, size_t sample_frame_16k_size, int32_t pitch_shift,
                            uint32_t skip_head, uint32_t return_length, void *d_out, size_t cap_per_stream, size_t *out_len, int sync);
/* the same with one pitch shift PER STREAM (pitch_shift[n_streams]): every stream of a batch is a caller of its own with its own
 * settings, as every stream is its own process in the reference (obs-rvc/src/lib.rs:701-707) */
rvc_status rvc_infer_batch_v(rvc_engine *e, const float *input, size_t n, size_t sample_frame_16k_size, const int32_t *pitch_shift,
                             uint32_t skip_head, uint32_t return_length, float *out, size_t cap_per_stream, size_t *out_len);
rvc_status rvc_infer_device_v(rvc_engine *e, const void *d_input, size_t n, size_t sample_frame_16k_size, const int32_t *pitch_shift,
                              uint32_t skip_head, uint32_t return_length, void *d_out, size_t cap_per_stream, size_t *out_len, int sync);
/* many streams, every stream with ITS OWN geometry (in the reference every stream is a process with its own chunk length, crossfade and
 * extra context: obs-rvc/src/lib.rs:200-227, 694).  All arrays have n_streams entries (inputs / outs: host pointers per stream;
 * pitch_shift may be NULL = 0 for every stream).  Streams with equal (n, sample_frame_16k_size, skip_head, return_length) run as one
 * batch; a server can mix 160 ms and 300 ms callers in one call.  At most as many different geometries per call as the plan cache holds
 * (8 unless rvc_set_plan_cache raised it). */
rvc_status rvc_infer_batch_g(rvc_engine *e, const float *const *inputs, const size_t *n, const size_t *sample_frame_16k_size, const int32_t *pitch_shift,
                             const uint32_t *skip_head, const uint32_t *return_length, float *const *outs, const size_t *caps, size_t *out_lens);
rvc_status rvc_synchronize(rvc_engine *e);
void rvc_set_use_graph(rvc_engine *e, int on);    /* replay the per-chunk launch sequence from a hipGraph */
/* Plan cache.  The engine keeps one "plan" per call geometry (n, sample_frame_16k_size, skip_head, return_length, stream count, retrieval on/off):
 * its activation arena, plan-time weight copies and launch list.  Every plugin instance has its own geometry (obs-rvc/src/lib.rs:200-227), so a
 * server sees as many plans as it has distinct caller settings.  The cache holds n_plans of them (default 8, 2..256), least recently used evicted
 * first.  A MISS costs a plan build: arena allocation (tens of MB at one stream, GBs at 64), composed weights and a device synchronisation --
 * tens of milliseconds; a server that rotates through more geometries than the cache holds pays that on every call.  rvc_plan_cache_info reports
 * capacity, plans currently cached and plans built since rvc_create (-> 1 on a valid engine). */
rvc_status rvc_set_plan_cache(rvc_engine *e, int n_plans);
/* Plan-time selection by measurement.  The planner's kernel / tile rules are thresholds measured on one box at a handful of stream counts and one geometry;
 * every plugin instance has its own geometry (obs-rvc/src/lib.rs:200-227) and GPUs of one pool differ.  With on = 1 (the default) a plan of MORE THAN 4 STREAMS
 * times, while it is built and on the engine's own device, the eligible kernels / tiles of every layer that has more than one (the rule-based choice and its
 * neighbours across the nearest thresholds: a warm-up and 1-3 timed launches each) and keeps the fastest; results are cached per process by (device, layer
 * signature, stream count).  All candidates are parity-tested kernels: the choice affects fp32 summation order only.  A first plan build at 64 streams takes
 * ~0.1-0.3 s longer, later builds of the same layers nothing.  on = 0: the rules only (results then do not depend on timing).  rvc_plan_autotune_info reports
 * the engine's LAST plan build: layers tuned by trials, layers whose choice differs from the rules, layers served from the cache, ms in trials, ms in all. */
rvc_status rvc_set_plan_autotune(rvc_engine *e, int on);
rvc_status rvc_plan_autotune_info(rvc_engine *e, int *tuned, int *changed, int *cache_hits, double *tune_ms, double *build_ms);
/* EXPLORATORY (no counterpart in the reference, off by default, never used for the headline figure): mode 1 = the 1-D layers with >= 128 output rows (ContentVec's projections and stem, the decoder's wide stages) run every
 * fp32 product as three bf16 matrix-core products (a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulate) in launches of >= 250 workgroups of 128 x 128 (many
 * streams); ~2^-16 relative per product instead of fp32 rounding.  mode 0 = fp32 everywhere, the reference's arithmetic. */
rvc_status rvc_set_gemm_precision(rvc_engine *e, int mode);
int rvc_plan_cache_info(rvc_engine *e, int *capacity, int *cached, long long *builds);
/* Offline throughput mode (no counterpart in the reference, whose protocol is one request at a time): with on != 0, consecutive
 * rvc_infer_device(..., sync = 0) calls overlap chunk i+1's ContentVec / f0 branches with chunk i's synthesizer (two plan slots).
 * Results are identical to the serial order; every call needs its own output buffer until rvc_synchronize. */
void rvc_set_pipeline(rvc_engine *e, int on);

/* ---- caller-side post-processing of the plugin, same host-buffer convention ---- */
/* envelop_mixing (obs-rvc/src/rt_utils.rs:119-132): output[i] *= (rms(input)/max(rms(output),1e-3))^(1-mix_rate) */
rvc_status rvc_envelop_mixing(rvc_engine *e, const float *input, float *output, size_t output_len, size_t sample_rate, double mix_rate);
/* get_sola_offset (rt_utils.rs:60-90) + crossfade / tail save / frame extraction (obs-rvc/src/lib.rs:768-794).
 * output needs sola_len + search + frame valid samples; sola_buffer (sola_len) is updated in place. */
rvc_status rvc_sola_step(rvc_engine *e, float *output, size_t output_len, float *sola_buffer, size_t sola_len, size_t search,
                         size_t frame, float *frame_out, size_t *sola_offset);

/* ---- the plugin's two sample-rate converters ---- */
/* rubato::FftFixedInOut::<f32>::new(rate_in, rate_out, chunk_size_in, 1) at obs-rvc/src/lib.rs:236-242 (host rate -> 16 kHz in front
 * of infer, model rate -> host rate behind it); one channel.  The converter runs on the engine's device and stream and must be
 * destroyed before the engine. */
typedef struct rvc_resampler rvc_resampler;
rvc_status rvc_resampler_create(rvc_engine *e, size_t rate_in, size_t rate_out, size_t chunk_size_in, rvc_resampler **out);
void rvc_resampler_destroy(rvc_resampler *r);
size_t rvc_resampler_input_frames_next(rvc_resampler *r);    /* Resampler::input_frames_next: frames one process call consumes */
size_t rvc_resampler_output_frames_max(rvc_resampler *r);    /* Resampler::output_frames_max (lib.rs:244): frames it produces */
void rvc_resampler_reset(rvc_resampler *r);                  /* Resampler::reset */
/* Resampler::process (lib.rs:675) / process_into_buffer (lib.rs:747-749).  n_in must equal input_frames_next(), otherwise
 * RVC_SHAPE (rubato: ResampleError::WrongNumberOfInputFrames, on which the plugin panics); *n_out = output_frames_max(). */
rvc_status rvc_resampler_process(rvc_resampler *r, const float *in, size_t n_in, float *out, size_t cap, size_t *n_out);
/* device-resident variant (HIP device pointers, engine's stream; no sync unless sync != 0) */
rvc_status rvc_resampler_process_device(rvc_resampler *r, const void *d_in, void *d_out, int sync);

/* ---- the plugin's per-chunk state machine as one call (SURVEY.md section 8 rows f1-f3 chained, all buffers resident in HBM) ---- */
/* `create`/`update` + `process_one_frame` of the filter (obs-rvc/src/lib.rs:181-260, 659-795): host-rate ring, 16 kHz ring, both
 * resamplers, RvcInfer::infer, RMS envelope mixing and SOLA.  One H2D copy (the new chunk), one D2H copy (the finished frame) and
 * one synchronisation per chunk.  Lengths in seconds as in the plugin's settings; skip_inference != 0 = pass-through mode
 * (lib.rs:224-227).  The session covers every stream of the engine (rvc_set_streams before rvc_session_create): process then takes
 * input [streams][n] and writes output [streams][cap], sola_offset [streams].  Destroy the session before the engine. */
typedef struct rvc_session rvc_session;
rvc_status rvc_session_create(rvc_engine *e, size_t sample_rate, double sample_length, double crossfade_length, double extra_inference_time,
                              size_t model_output_sample_rate, int32_t pitch_shift, double rms_mix_rate, int skip_inference, rvc_session **out);
void rvc_session_destroy(rvc_session *s);
size_t rvc_session_frame_size(rvc_session *s);                 /* sample_frame_size: samples per process call, in and out */
void rvc_session_set_params(rvc_session *s, int32_t pitch_shift, double rms_mix_rate);      /* every stream */
/* one stream's pitch shift and RMS mix rate (the plugin's per-instance settings, obs-rvc/src/lib.rs:174-185); the others keep theirs */
rvc_status rvc_session_set_params_stream(rvc_session *s, int stream, int32_t pitch_shift, double rms_mix_rate);
void rvc_session_geometry(rvc_session *s, int32_t out[10]);   /* the derived sizes of lib.rs:200-227 (see session.hip.h) */
rvc_status rvc_session_process(rvc_session *s, const float *input_sample, size_t n, float *output, size_t cap, size_t *sola_offset);

/* ---- what THIS GPU sustains, measured in-run (bench.py; boxes of one pool differ by ~10 % on matrix-core-bound work) ---- */
/* rvc_calibrate: ~50 ms of device time.  A bare v_mfma_f32_32x32x2_f32 stream on every SIMD (four waves per SIMD, non-zero operands) -> fp32 matrix-core
 * TFLOP/s actually reached and the shader clock it ran at (s_memtime cycles per s_memrealtime tick); a float4 read stream over 1 GiB -> HBM TB/s. */
typedef struct {
    double mfma_f32_tflops, mfma_sclk_mhz, mfma_ms;
    double hbm_read_tbs, hbm_sclk_mhz;
    double ms_total;
    int compute_units;
} rvc_calibration;
rvc_status rvc_calibrate(int device, rvc_calibration *out);
/* Effective shader clock WHILE other work runs: start leaves eight sleeping one-wave workgroups (one per XCD) on a stream of their own that count shader
 * cycles against the 100 MHz real-time counter; stop ends them and reports the mean / minimum over the XCDs and the seconds observed.  The waves leave
 * by themselves after 20 s (do not run it under a profiler that serialises dispatches).  One monitor per device. */
rvc_status rvc_clock_monitor_start(int device);
rvc_status rvc_clock_monitor_stop(int device, double *sclk_mhz_mean, double *sclk_mhz_min, double *seconds);

/* ---- measurement / debugging ---- */
/* total milliseconds of the last infer measured with HIP events on the engine's stream */
float rvc_last_gpu_ms(rvc_engine *e);
/* HIP-event timing of the dominant (implicit-GEMM) kernel class over the last infer call:
 * number of launches, summed milliseconds, summed algorithmic FLOPs (2*M*N*K per launch) */
rvc_status rvc_profile_last(rvc_engine *e, int *launches, double *kernel_ms, double *flops);
/* same for the HBM-bound retrieval launch (knn_scan_select_kernel: scan, select, exact re-rank and blend in one launch): launches, summed ms, summed algorithmic bytes (index size per pass) */
rvc_status rvc_profile_last_knn(rvc_engine *e, int *launches, double *kernel_ms, double *bytes);
void rvc_set_profile(rvc_engine *e, int on);
/* named intermediate tensor of stream 0 of the last call, contiguous row-major (tests).  on = 1: taps on the EXPLICIT plan (every
 * LayerNorm its own launch, WaveNets layer by layer: each tap has the oracle's meaning); on = 2: taps on the PRODUCTION plan (folded
 * LayerNorms, composed WaveNets: tensors that are not yet normalised there carry a ".raw" suffix); 0 = off */
void rvc_enable_taps(rvc_engine *e, int on);
rvc_status rvc_get_tap(rvc_engine *e, const char *name, float *out, size_t cap, size_t *n);
void rvc_get_pitch_cache(rvc_engine *e, int stream, float *out1024);
/* raw device pointer of the engine's weights/index for RCCL broadcast at load (config 5) */
void *rvc_index_device_ptr(rvc_engine *e, size_t *bytes);
int rvc_device(rvc_engine *e);
const char *rvc_version(void);

#ifdef __cplusplus
}
#endif
#endif
