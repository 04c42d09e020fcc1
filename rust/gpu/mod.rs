// src/core/search/gpu/mod.rs — add `pub mod gpu;` to src/core/search/mod.rs and the link line of rust/build.rs to build.rs.
// ffi.rs is generated from include/rucene_gpu.h (scripts/gen_rust_ffi.py); searcher.rs is the IndexSearcher seam.
pub mod ffi;
pub mod searcher;

pub use self::searcher::GpuIndexSearcher;
