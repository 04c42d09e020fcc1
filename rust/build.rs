// build.rs of the Rucene crate: link librucene_gpu.so (hipcc --offload-arch=gfx950; `python -c "import __graft_entry__ as g; g.build()"`
// in this repository leaves it at rucene_amd/librucene_gpu.so). RUCENE_GPU_LIB_DIR = the directory that holds it.
fn main() {
    println!("cargo:rustc-link-search=native={}", std::env::var("RUCENE_GPU_LIB_DIR").unwrap());
    println!("cargo:rustc-link-lib=dylib=rucene_gpu");
}
