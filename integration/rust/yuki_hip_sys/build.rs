// Links libyuki_hip.so.  YUKI_HIP_LIB_DIR = directory holding the library built by
// `make -C yuki_amd/csrc` (it pulls in the HIP runtime itself; no ROCm link flags here).
fn main() {
    println!("cargo:rerun-if-env-changed=YUKI_HIP_LIB_DIR");
    if let Ok(dir) = std::env::var("YUKI_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=yuki_hip");
}
