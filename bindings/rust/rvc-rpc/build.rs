// rvc-rpc/build.rs (new file; add `build = "build.rs"` to rvc-rpc/Cargo.toml, see main.rs.patch): the `rvc` crate links
// librvc_mi355x.so and publishes its directory as DEP_RVC_MI355X_LIBDIR (`links = "rvc_mi355x"`); the rpath has to be set where the
// binary is linked, i.e. here.  With it `rvc-rpc` starts without LD_LIBRARY_PATH, as the plugin spawns it (rvcadapter.rs:37-48).
fn main() {
    let dir = std::env::var("DEP_RVC_MI355X_LIBDIR").expect("the rvc crate's build script did not run (DEP_RVC_MI355X_LIBDIR)");
    println!("cargo:rustc-link-arg-bins=-Wl,-rpath,{}", dir);
}
