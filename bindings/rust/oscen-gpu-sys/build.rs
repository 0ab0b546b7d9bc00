// Link against <repo>/oscen_amd/liboscen_gpu.so (python -m oscen_amd.build); OSCEN_GPU_LIB_DIR overrides.
fn main() {
    let dir = std::env::var("OSCEN_GPU_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../../oscen_amd").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=oscen_gpu");
    println!("cargo:rerun-if-env-changed=OSCEN_GPU_LIB_DIR");
}
