// build.rs additions (the reference's build.rs:1-18 only exports the git hash / branch).
// THZGPU_LIB_DIR = directory holding libthzgpu.so and libthzio.so (thz_image_explorer_amd/ of the engine repo).
fn main() {
    if let Ok(dir) = std::env::var("THZGPU_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=thzgpu");
    println!("cargo:rustc-link-lib=dylib=thzio");
    println!("cargo:rerun-if-env-changed=THZGPU_LIB_DIR");
}
