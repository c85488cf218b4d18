// Link the `rvc` crate against librvc_mi355x.so (built in the engine's repository by
// `python -c "import __graft_entry__ as g; g.build()"`, found in obs_rvc_amd/csrc/).
//
// RVC_MI355X_LIB_DIR must name the directory that holds the library; there is no default (the crate is copied into the reference's
// workspace, where no relative path to the engine's repository means anything).  A library crate cannot put an rpath on the binaries
// that depend on it (`cargo:rustc-link-arg` does not propagate), so the directory is handed on as metadata -- `links = "rvc_mi355x"`
// in Cargo.toml makes it DEP_RVC_MI355X_LIBDIR in the build script of every dependent crate; `rvc-rpc/build.rs` turns it into the
// binary's rpath.
use std::{env, path::PathBuf};

fn main() {
    println!("cargo:rerun-if-env-changed=RVC_MI355X_LIB_DIR");
    let dir = match env::var("RVC_MI355X_LIB_DIR") {
        Ok(d) if !d.is_empty() => PathBuf::from(d),
        _ => panic!("set RVC_MI355X_LIB_DIR to the directory that holds librvc_mi355x.so (obs_rvc_amd/csrc of the engine's repository)"),
    };
    assert!(dir.join("librvc_mi355x.so").exists(), "{} does not hold librvc_mi355x.so", dir.display());
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rvc_mi355x");
    println!("cargo:libdir={}", dir.display());
}
