fn main() {
    // libzinc_groth16_hip.so is built by `make -C zinc_amd/csrc`; point ZINC_GROTH16_LIB_DIR at zinc_amd/lib
    if let Ok(dir) = std::env::var("ZINC_GROTH16_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=zinc_groth16_hip");
}
