// libzkw.so is built by `python -c "import __graft_entry__ as g; g.build()"` into era-zk_evm_amd/
fn main() {
    let root = std::path::Path::new(env!("CARGO_MANIFEST_DIR")).join("../../era-zk_evm_amd");
    println!("cargo:rustc-link-search=native={}", root.display());
    println!("cargo:rustc-link-lib=dylib=zkw");
    println!("cargo:rerun-if-changed=../../include/zkw.h");
}
