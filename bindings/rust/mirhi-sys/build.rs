// points the linker at the in-tree libmirhi.so (renderer-rs_amd/); MIRHI_LIB_DIR overrides
fn main() {
    let dir = std::env::var("MIRHI_LIB_DIR").unwrap_or_else(|_| format!("{}/../../../renderer-rs_amd", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=MIRHI_LIB_DIR");
}
