// Link against librvc_mi355x.so (built by `python -c "import __graft_entry__ as g; g.build()"` in the engine's repository).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("RVC_MI355X_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../obs_rvc_amd/csrc")
    });
    println!("cargo:rerun-if-env-changed=RVC_MI355X_LIB_DIR");
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rvc_mi355x");
    // let the binaries of dependent crates (rvc-rpc) find the library next to where it was built
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
