// Links libaudioflow_dsp.so.  AUDIOFLOW_DSP_LIB_DIR = the directory holding it
// (this repository: audio-flow-rs_amd/).  Not compiled in this repository's image (no Rust toolchain).
fn main() {
    if let Ok(dir) = std::env::var("AUDIOFLOW_DSP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=audioflow_dsp");
    println!("cargo:rerun-if-env-changed=AUDIOFLOW_DSP_LIB_DIR");
}
