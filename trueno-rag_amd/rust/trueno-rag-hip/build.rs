// Link against libtrag.so built by `make -C trueno-rag_amd/csrc`.
fn main() {
    let dir = std::env::var("TRAG_LIB_DIR").unwrap_or_else(|_| "../../lib".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=trag");
    println!("cargo:rerun-if-env-changed=TRAG_LIB_DIR");
}
