//! `extern "C"` declarations for `include/rvc_mi355x.h` (hand-written; `bindgen include/rvc_mi355x.h` gives the same items).
//! Names, argument order and integer widths follow the header one to one: `size_t` = `usize`, `int` = `c_int`,
//! `rvc_status` = `c_int` (0 ok, 1 ModelNotLoaded, 2 ContentvecNotLoaded, 3 F0NotLoaded, 4 Backend, 5 Shape, 6 Panic).
#![allow(dead_code)]
use std::os::raw::{c_char, c_double, c_float, c_int, c_longlong, c_void};

#[repr(C)]
pub struct RvcEngine {
    _private: [u8; 0],
}
#[repr(C)]
pub struct RvcResampler {
    _private: [u8; 0],
}
#[repr(C)]
pub struct RvcSession {
    _private: [u8; 0],
}
/// rvc_calibration (include/rvc_mi355x.h)
#[repr(C)]
pub struct RvcCalibration {
    pub mfma_f32_tflops: c_double,
    pub mfma_sclk_mhz: c_double,
    pub mfma_ms: c_double,
    pub hbm_read_tbs: c_double,
    pub hbm_sclk_mhz: c_double,
    pub ms_total: c_double,
    pub compute_units: c_int,
}

pub const RVC_OK: c_int = 0;
pub const RVC_MODEL_NOT_LOADED: c_int = 1;
pub const RVC_CONTENTVEC_NOT_LOADED: c_int = 2;
pub const RVC_F0_NOT_LOADED: c_int = 3;
pub const RVC_BACKEND: c_int = 4;
pub const RVC_SHAPE: c_int = 5;
pub const RVC_PANIC: c_int = 6;
pub const RVC_RCCL_UNIQUE_ID_BYTES: usize = 128;

extern "C" {
    // ---- RvcInfer (rvc/src/rvc.rs:30-220)
    pub fn rvc_create(data_path: *const c_char, device: c_int, out: *mut *mut RvcEngine) -> c_int;
    pub fn rvc_destroy(e: *mut RvcEngine);
    pub fn rvc_load_contentvec(e: *mut RvcEngine, model_version: c_int) -> c_int;
    pub fn rvc_load_model(e: *mut RvcEngine, model_path: *const c_char) -> c_int;
    pub fn rvc_load_f0(e: *mut RvcEngine, pitch_algorithm: c_int) -> c_int;
    pub fn rvc_unload_model(e: *mut RvcEngine);
    pub fn rvc_hubert(e: *mut RvcEngine, input: *const c_float, n: usize, out: *mut c_float, cap: usize, dims: *mut usize) -> c_int;
    pub fn rvc_extract_feature(e: *mut RvcEngine, input: *const c_float, n: usize, out: *mut c_float, cap: usize, dims: *mut usize) -> c_int;
    pub fn rvc_pitch(e: *mut RvcEngine, input: *const c_float, n: usize, pitch_shift: i32, sample_frame_16k_size: usize,
                     out: *mut c_float, cap: usize, out_len: *mut usize) -> c_int;
    pub fn rvc_infer(e: *mut RvcEngine, input: *const c_float, n: usize, sample_frame_16k_size: usize, has_pitch_shift: c_int,
                     pitch_shift: i32, skip_head: u32, return_length: u32, out: *mut c_float, cap: usize, out_len: *mut usize) -> c_int;
    pub fn rvc_last_error_message(e: *mut RvcEngine) -> *const c_char;

    // ---- retrieval index, noise seed, state
    pub fn rvc_load_index(e: *mut RvcEngine, vectors: *const c_float, n: usize, dim: usize) -> c_int;
    pub fn rvc_load_index_device(e: *mut RvcEngine, d_vectors: *const c_void, n: usize, dim: usize) -> c_int;
    pub fn rvc_set_index_rate(e: *mut RvcEngine, rate: c_float);
    pub fn rvc_get_knn(e: *mut RvcEngine, idx: *mut i32, dist: *mut c_float, cap_rows: usize, rows: *mut usize) -> c_int;
    pub fn rvc_set_noise_seed(e: *mut RvcEngine, seed: u32, stream_id: u32);
    pub fn rvc_reset_state(e: *mut RvcEngine);

    // ---- multi-GPU: the one collective (index broadcast at load, RCCL over xGMI)
    pub fn rvc_rccl_unique_id(id128: *mut c_void) -> c_int;
    pub fn rvc_index_broadcast(e: *mut RvcEngine, unique_id128: *const c_void, rank: c_int, world: c_int,
                               vectors: *const c_float, n: usize, dim: usize) -> c_int;
    pub fn rvc_rccl_available() -> c_int;
    pub fn rvc_index_broadcast_info(e: *mut RvcEngine, ms: *mut f64, ranks: *mut c_int) -> c_int;

    // ---- many streams per GPU
    pub fn rvc_set_streams(e: *mut RvcEngine, n_streams: c_int) -> c_int;
    pub fn rvc_infer_batch(e: *mut RvcEngine, input: *const c_float, n: usize, sample_frame_16k_size: usize, pitch_shift: i32,
                           skip_head: u32, return_length: u32, out: *mut c_float, cap_per_stream: usize, out_len: *mut usize) -> c_int;
    pub fn rvc_infer_batch_v(e: *mut RvcEngine, input: *const c_float, n: usize, sample_frame_16k_size: usize, pitch_shift: *const i32,
                             skip_head: u32, return_length: u32, out: *mut c_float, cap_per_stream: usize, out_len: *mut usize) -> c_int;
    pub fn rvc_infer_device_v(e: *mut RvcEngine, d_input: *const c_void, n: usize, sample_frame_16k_size: usize, pitch_shift: *const i32,
                              skip_head: u32, return_length: u32, d_out: *mut c_void, cap_per_stream: usize, out_len: *mut usize, sync: c_int) -> c_int;
    pub fn rvc_infer_batch_g(e: *mut RvcEngine, inputs: *const *const c_float, n: *const usize, sample_frame_16k_size: *const usize, pitch_shift: *const i32,
                             skip_head: *const u32, return_length: *const u32, outs: *const *mut c_float, caps: *const usize, out_lens: *mut usize) -> c_int;
    pub fn rvc_infer_device(e: *mut RvcEngine, d_input: *const c_void, n: usize, sample_frame_16k_size: usize, pitch_shift: i32,
                            skip_head: u32, return_length: u32, d_out: *mut c_void, cap_per_stream: usize, out_len: *mut usize,
                            sync: c_int) -> c_int;
    pub fn rvc_synchronize(e: *mut RvcEngine) -> c_int;
    pub fn rvc_set_use_graph(e: *mut RvcEngine, on: c_int);
    pub fn rvc_set_pipeline(e: *mut RvcEngine, on: c_int);
    // plan cache (one plan per call geometry; LRU) and the retrieval's recovered hand-off time-outs
    pub fn rvc_set_plan_cache(e: *mut RvcEngine, n_plans: c_int) -> c_int;
    pub fn rvc_plan_cache_info(e: *mut RvcEngine, capacity: *mut c_int, cached: *mut c_int, builds: *mut c_longlong) -> c_int;
    pub fn rvc_retrieval_recoveries(e: *mut RvcEngine) -> c_longlong;
    pub fn rvc_set_gemm_precision(e: *mut RvcEngine, mode: c_int) -> c_int;
    // plan-time selection among eligible kernels / tiles by measurement (default on above 4 streams)
    pub fn rvc_set_plan_autotune(e: *mut RvcEngine, on: c_int) -> c_int;
    pub fn rvc_plan_autotune_info(e: *mut RvcEngine, tuned: *mut c_int, changed: *mut c_int, cache_hits: *mut c_int, tune_ms: *mut c_double,
                                  build_ms: *mut c_double) -> c_int;
    // what this GPU sustains, measured in-run (bare fp32-MFMA stream, HBM read stream), and the effective shader clock while other work runs
    pub fn rvc_calibrate(device: c_int, out: *mut RvcCalibration) -> c_int;
    pub fn rvc_clock_monitor_start(device: c_int) -> c_int;
    pub fn rvc_clock_monitor_stop(device: c_int, sclk_mhz_mean: *mut c_double, sclk_mhz_min: *mut c_double, seconds: *mut c_double) -> c_int;

    // ---- caller-side steps of the plugin (obs-rvc/src/rt_utils.rs, obs-rvc/src/lib.rs:236-260,659-795)
    pub fn rvc_envelop_mixing(e: *mut RvcEngine, input: *const c_float, output: *mut c_float, output_len: usize, sample_rate: usize,
                              mix_rate: c_double) -> c_int;
    pub fn rvc_sola_step(e: *mut RvcEngine, output: *mut c_float, output_len: usize, sola_buffer: *mut c_float, sola_len: usize,
                         search: usize, frame: usize, frame_out: *mut c_float, sola_offset: *mut usize) -> c_int;
    pub fn rvc_resampler_create(e: *mut RvcEngine, rate_in: usize, rate_out: usize, chunk_size_in: usize, out: *mut *mut RvcResampler) -> c_int;
    pub fn rvc_resampler_destroy(r: *mut RvcResampler);
    pub fn rvc_resampler_input_frames_next(r: *mut RvcResampler) -> usize;
    pub fn rvc_resampler_output_frames_max(r: *mut RvcResampler) -> usize;
    pub fn rvc_resampler_reset(r: *mut RvcResampler);
    pub fn rvc_resampler_process(r: *mut RvcResampler, input: *const c_float, n_in: usize, out: *mut c_float, cap: usize, n_out: *mut usize) -> c_int;
    pub fn rvc_resampler_process_device(r: *mut RvcResampler, d_in: *const c_void, d_out: *mut c_void, sync: c_int) -> c_int;
    pub fn rvc_session_create(e: *mut RvcEngine, sample_rate: usize, sample_length: c_double, crossfade_length: c_double,
                              extra_inference_time: c_double, model_output_sample_rate: usize, pitch_shift: i32, rms_mix_rate: c_double,
                              skip_inference: c_int, out: *mut *mut RvcSession) -> c_int;
    pub fn rvc_session_destroy(s: *mut RvcSession);
    pub fn rvc_session_frame_size(s: *mut RvcSession) -> usize;
    pub fn rvc_session_set_params(s: *mut RvcSession, pitch_shift: i32, rms_mix_rate: c_double);
    pub fn rvc_session_set_params_stream(s: *mut RvcSession, stream: c_int, pitch_shift: i32, rms_mix_rate: f64) -> c_int;
    pub fn rvc_session_geometry(s: *mut RvcSession, out: *mut i32);
    pub fn rvc_session_process(s: *mut RvcSession, input_sample: *const c_float, n: usize, output: *mut c_float, cap: usize,
                               sola_offset: *mut usize) -> c_int;

    // ---- measurement / debugging
    pub fn rvc_last_gpu_ms(e: *mut RvcEngine) -> c_float;
    pub fn rvc_profile_last(e: *mut RvcEngine, launches: *mut c_int, kernel_ms: *mut c_double, flops: *mut c_double) -> c_int;
    pub fn rvc_profile_last_knn(e: *mut RvcEngine, launches: *mut c_int, kernel_ms: *mut c_double, bytes: *mut c_double) -> c_int;
    pub fn rvc_set_profile(e: *mut RvcEngine, on: c_int);
    pub fn rvc_enable_taps(e: *mut RvcEngine, on: c_int);
    pub fn rvc_get_tap(e: *mut RvcEngine, name: *const c_char, out: *mut c_float, cap: usize, n: *mut usize) -> c_int;
    pub fn rvc_get_pitch_cache(e: *mut RvcEngine, stream: c_int, out1024: *mut c_float);
    pub fn rvc_index_device_ptr(e: *mut RvcEngine, bytes: *mut usize) -> *mut c_void;
    pub fn rvc_device(e: *mut RvcEngine) -> c_int;
    pub fn rvc_version() -> *const c_char;
}
