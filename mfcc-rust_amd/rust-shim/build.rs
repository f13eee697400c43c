// Links libspeechsauce_amd.so (built by `make -C mfcc-rust_amd/csrc`).  Set SPEECHSAUCE_AMD_LIB_DIR to its directory.
fn main() {
    let dir = std::env::var("SPEECHSAUCE_AMD_LIB_DIR").unwrap_or_else(|_| "../lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=speechsauce_amd");
    println!("cargo:rerun-if-env-changed=SPEECHSAUCE_AMD_LIB_DIR");
}
