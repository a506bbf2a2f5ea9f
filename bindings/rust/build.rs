// Links libwafer_hip.so (built by `python -m wafer_amd.build`).  WAFER_HIP_LIB_DIR points at the
// directory holding it (default: ../../wafer_amd relative to this crate).
fn main() {
    let dir = std::env::var("WAFER_HIP_LIB_DIR").unwrap_or_else(|_| "../../wafer_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=wafer_hip");
    println!("cargo:rerun-if-env-changed=WAFER_HIP_LIB_DIR");
}
