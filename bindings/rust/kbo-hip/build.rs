// KBO_HIP_LIB_DIR = the directory that holds libkbo_hip.so (kbo_amd/ in this repository after
// `python -c "import __graft_entry__ as g; g.build()"`).
fn main() {
    let dir = std::env::var("KBO_HIP_LIB_DIR").expect("set KBO_HIP_LIB_DIR to the directory of libkbo_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=kbo_hip");
    println!("cargo:rerun-if-env-changed=KBO_HIP_LIB_DIR");
}
