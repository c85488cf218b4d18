mod ffi;
mod rvc;
pub use rvc::*;
