// Links the prebuilt engine (python -m fhe_string_amd.build puts libfhe_string_amd.so into <repo>/fhe_string_amd).
// FHE_STRING_AMD_LIB_DIR overrides the search directory.
fn main() {
    let dir = std::env::var("FHE_STRING_AMD_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{here}/../../fhe_string_amd")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=fhe_string_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=FHE_STRING_AMD_LIB_DIR");
}
