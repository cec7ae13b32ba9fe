// Points rustc at libphysics_hip.so (built in-tree by `make -C physics_amd/csrc`).
fn main() {
    let dir = std::env::var("PHYSICS_HIP_LIB_DIR").unwrap_or_else(|_| "../../physics_amd/csrc".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=physics_hip");
    println!("cargo:rerun-if-env-changed=PHYSICS_HIP_LIB_DIR");
}
