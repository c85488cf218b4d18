//! `RvcInfer` of the reference (`rvc/src/rvc.rs:18-220`) over `librvc_mi355x.so`: same method names, argument meaning, return
//! shapes and error behaviour; the three ONNX Runtime sessions, the mel / STFT front end, the pitch cache and `get_f0_post` now
//! live on the GPU behind one opaque handle.  One handle = one stream = one thread at a time, as `&mut self` already demanded.
use std::ffi::{CStr, CString};
use std::path::PathBuf;

use ndarray::{Array1, Array3, ArrayView1};
use rvc_common::{
    enums::{PitchAlgorithm, RvcModelVersion},
    errors::{BackendError, RvcInferError},
};

use crate::ffi;

pub struct RvcInfer {
    handle: *mut ffi::RvcEngine,
}

// The reference's RvcInfer is Send (ort::Session is); the engine handle may move between threads but is not re-entrant.
unsafe impl Send for RvcInfer {}

fn c_path(p: &PathBuf) -> CString {
    CString::new(p.to_string_lossy().as_bytes()).expect("path contains a NUL byte")
}

impl RvcInfer {
    fn message(&self) -> String {
        unsafe {
            let m = ffi::rvc_last_error_message(self.handle);
            if m.is_null() { String::new() } else { CStr::from_ptr(m).to_string_lossy().into_owned() }
        }
    }

    /// rvc_status -> the reference's error type (`rvc-common/src/errors.rs:2-8`).  `RVC_PANIC` marks inputs on which the
    /// reference itself panics (`rmvpe.rs:124` out-of-range gather, `rvc.rs:155` slice out of range): panic here too.
    fn check(&self, rc: i32) -> Result<(), RvcInferError> {
        match rc {
            ffi::RVC_OK => Ok(()),
            ffi::RVC_MODEL_NOT_LOADED => Err(RvcInferError::ModelNotLoaded),
            ffi::RVC_CONTENTVEC_NOT_LOADED => Err(RvcInferError::ContentvecNotLoaded),
            ffi::RVC_F0_NOT_LOADED => Err(RvcInferError::F0NotLoaded),
            ffi::RVC_PANIC => panic!("rvc engine: {}", self.message()),
            code => Err(RvcInferError::Backend(BackendError { code, message: self.message() })),
        }
    }

    fn check_load(&self, rc: i32) -> Result<(), BackendError> {
        if rc == ffi::RVC_OK { Ok(()) } else { Err(BackendError { code: rc, message: self.message() }) }
    }

    /// `RvcInfer::new` (rvc.rs:30-44).  `data_path` holds `contentvec/`, `f0/` as before, with `.rvcw` blobs
    /// (`python -m obs_rvc_amd.importers` converts the `.onnx` / `.pth` files).  Panics without a usable MI355X: the reference has
    /// no fallible constructor either, and there is no CPU fallback.
    pub fn new(data_path: PathBuf) -> Self {
        let p = c_path(&data_path);
        let mut handle: *mut ffi::RvcEngine = std::ptr::null_mut();
        let rc = unsafe { ffi::rvc_create(p.as_ptr(), -1, &mut handle) };
        assert!(rc == ffi::RVC_OK && !handle.is_null(), "rvc_create failed (status {}): no HIP device?", rc);
        RvcInfer { handle }
    }

    /// rvc.rs:46-54: `<data>/contentvec/vec-{768-layer-12,256-layer-9}.rvcw` (models.rs:58-61)
    pub fn load_contentvec(&mut self, model_version: RvcModelVersion) -> Result<(), BackendError> {
        let v: i64 = model_version.into();
        self.check_load(unsafe { ffi::rvc_load_contentvec(self.handle, v as i32) })
    }

    /// rvc.rs:56-60: the user's synthesizer; a path ending in `.onnx` is mapped to its `.rvcw` sibling
    pub fn load_model(&mut self, model_path: PathBuf) -> Result<(), BackendError> {
        let p = c_path(&model_path);
        self.check_load(unsafe { ffi::rvc_load_model(self.handle, p.as_ptr()) })
    }

    /// rvc.rs:62-75: `<data>/f0/rmvpe.rvcw` (models.rs:72)
    pub fn load_f0(&mut self, pitch_algorithm: PitchAlgorithm) -> Result<(), BackendError> {
        let a: i64 = pitch_algorithm.into();
        self.check_load(unsafe { ffi::rvc_load_f0(self.handle, a as i32) })
    }

    /// rvc.rs:77-79
    pub fn unload_model(&mut self) {
        unsafe { ffi::rvc_unload_model(self.handle) }
    }

    /// rvc.rs:81-97: ContentVec features, shape (1, C, T)
    pub fn hubert(&self, input: ArrayView1<f32>) -> Result<Array3<f32>, RvcInferError> {
        let x = input.as_standard_layout();
        let cap = 1024 * (x.len() / 320 + 8);
        let mut out = vec![0f32; cap];
        let mut dims = [0usize; 3];
        self.check(unsafe { ffi::rvc_hubert(self.handle, x.as_ptr(), x.len(), out.as_mut_ptr(), cap, dims.as_mut_ptr()) })?;
        out.truncate(dims[0] * dims[1] * dims[2]);
        Ok(Array3::from_shape_vec((dims[0], dims[1], dims[2]), out)?)
    }

    /// rvc.rs:99-109: frames duplicated to 2T+1, shape (1, 2T+1, C)
    pub fn extract_feature(&self, input: ArrayView1<f32>) -> Result<Array3<f32>, RvcInferError> {
        let x = input.as_standard_layout();
        let cap = 1024 * (2 * (x.len() / 320) + 16);
        let mut out = vec![0f32; cap];
        let mut dims = [0usize; 3];
        self.check(unsafe { ffi::rvc_extract_feature(self.handle, x.as_ptr(), x.len(), out.as_mut_ptr(), cap, dims.as_mut_ptr()) })?;
        out.truncate(dims[0] * dims[1] * dims[2]);
        Ok(Array3::from_shape_vec((dims[0], dims[1], dims[2]), out)?)
    }

    /// rvc.rs:111-131: f0 in Hz, one value per RMVPE frame, already multiplied by 2^(pitch_shift / 12) (integer division, as there)
    pub fn pitch(&mut self, input: ArrayView1<f32>, pitch_shift: i32, sample_frame_16k_size: usize) -> Result<Array1<f32>, RvcInferError> {
        let x = input.as_standard_layout();
        let mut out = vec![0f32; 4096];
        let mut n = 0usize;
        self.check(unsafe {
            ffi::rvc_pitch(self.handle, x.as_ptr(), x.len(), pitch_shift, sample_frame_16k_size, out.as_mut_ptr(), out.len(), &mut n)
        })?;
        out.truncate(n);
        Ok(Array1::from_vec(out))
    }

    /// rvc.rs:133-220: one chunk through ContentVec -> (retrieval) -> RMVPE -> pitch cache -> synthesizer; float PCM at the model rate
    pub fn infer(
        &mut self,
        input: ArrayView1<f32>,
        sample_frame_16k_size: usize,
        pitch_shift: Option<i32>,
        skip_head: u32,
        return_length: u32,
    ) -> Result<Array1<f32>, RvcInferError> {
        let x = input.as_standard_layout();
        let mut out = vec![0f32; return_length as usize * 1024 + 16];
        let mut n = 0usize;
        self.check(unsafe {
            ffi::rvc_infer(self.handle, x.as_ptr(), x.len(), sample_frame_16k_size, pitch_shift.is_some() as i32, pitch_shift.unwrap_or(0),
                           skip_head, return_length, out.as_mut_ptr(), out.len(), &mut n)
        })?;
        out.truncate(n);
        Ok(Array1::from_vec(out))
    }

    // ---- what the reference plumbs through its settings but never implements (rvc.rs:159 `// TODO: index search`) ----

    /// flat-L2 retrieval index, row-major (n, dim) fp32; `set_index_rate(0.0)` switches retrieval off again
    pub fn load_index(&mut self, vectors: ndarray::ArrayView2<f32>) -> Result<(), RvcInferError> {
        let v = vectors.as_standard_layout();
        self.check(unsafe { ffi::rvc_load_index(self.handle, v.as_ptr(), v.nrows(), v.ncols()) })
    }

    pub fn set_index_rate(&mut self, rate: f32) {
        unsafe { ffi::rvc_set_index_rate(self.handle, rate) }
    }

    /// rank 0 of a multi-GPU job: the 128-byte id to hand to the other ranks (pipe, file, TCP)
    pub fn rccl_unique_id() -> Result<[u8; ffi::RVC_RCCL_UNIQUE_ID_BYTES], BackendError> {
        let mut id = [0u8; ffi::RVC_RCCL_UNIQUE_ID_BYTES];
        let rc = unsafe { ffi::rvc_rccl_unique_id(id.as_mut_ptr() as *mut _) };
        if rc == ffi::RVC_OK { Ok(id) } else { Err(BackendError { code: rc, message: "librccl not available".into() }) }
    }

    /// every rank: ONE ncclBroadcast of the shared index from rank 0 into this GPU's HBM (RCCL over xGMI); `vectors` only on rank 0
    pub fn index_broadcast(&mut self, unique_id: &[u8; ffi::RVC_RCCL_UNIQUE_ID_BYTES], rank: i32, world: i32,
                           vectors: Option<ndarray::ArrayView2<f32>>) -> Result<(), RvcInferError> {
        let rc = match vectors {
            Some(v) => {
                let v = v.as_standard_layout();
                unsafe { ffi::rvc_index_broadcast(self.handle, unique_id.as_ptr() as *const _, rank, world, v.as_ptr(), v.nrows(), v.ncols()) }
            }
            None => unsafe { ffi::rvc_index_broadcast(self.handle, unique_id.as_ptr() as *const _, rank, world, std::ptr::null(), 0, 0) },
        };
        self.check(rc)
    }

    pub fn set_noise_seed(&mut self, seed: u32, stream_id: u32) {
        unsafe { ffi::rvc_set_noise_seed(self.handle, seed, stream_id) }
    }
}

impl Drop for RvcInfer {
    fn drop(&mut self) {
        unsafe { ffi::rvc_destroy(self.handle) }
    }
}
