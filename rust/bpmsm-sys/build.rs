// Links libbpmsm.so.  BPMSM_LIB_DIR = the directory that holds it (bulletproofs_amd/ of the bpmsm repository after
// `make -C bulletproofs_amd/csrc`); the HIP runtime comes in through the library's own DT_NEEDED entries.
fn main() {
    let dir = std::env::var("BPMSM_LIB_DIR").expect("set BPMSM_LIB_DIR to the directory holding libbpmsm.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=bpmsm");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=BPMSM_LIB_DIR");
}
