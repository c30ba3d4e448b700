//! `forma/build.rs` for the `hip` feature: link `libforma_hip.so` (built by `make -C forma_amd/csrc` in the
//! forma_hip repository: hipcc, `--offload-arch=gfx950`).
//!
//! `FORMA_HIP_LIB_DIR` names the directory holding `libforma_hip.so`; `ROCM_PATH` (default `/opt/rocm`) the ROCm
//! install whose `libamdhip64.so` the library itself depends on.

use std::{env, path::PathBuf};

fn main() {
    println!("cargo:rerun-if-env-changed=FORMA_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");

    if env::var_os("CARGO_FEATURE_HIP").is_none() {
        return;
    }

    let lib_dir = PathBuf::from(
        env::var_os("FORMA_HIP_LIB_DIR").expect("set FORMA_HIP_LIB_DIR to the directory that holds libforma_hip.so"),
    );
    assert!(
        lib_dir.join("libforma_hip.so").exists(),
        "{} does not hold libforma_hip.so",
        lib_dir.display()
    );

    let rocm = PathBuf::from(env::var_os("ROCM_PATH").unwrap_or_else(|| "/opt/rocm".into()));

    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-lib=dylib=forma_hip");
    // The library is found at run time next to where it was built, like the in-tree ctypes binding does.
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", rocm.join("lib").display());
}
