// Links the prebuilt HIP library; KZG_MI355X_LIB_DIR must point at rust-kzg_amd/csrc.
fn main() {
    let dir = std::env::var("KZG_MI355X_LIB_DIR")
        .expect("set KZG_MI355X_LIB_DIR to the directory holding libkzg_mi355x.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=kzg_mi355x");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rerun-if-env-changed=KZG_MI355X_LIB_DIR");
}
