// Builds libplonk_hip.so from the HIP sources of this repository with hipcc for gfx950 and tells
// cargo to link it.  With the `prebuilt` feature (or PLONK_HIP_LIB_DIR set) an existing library is
// linked instead.  There is one code path: gfx950 code objects, no CUDA shims, no fallback.
use std::env;
use std::path::PathBuf;
use std::process::Command;

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let repo = manifest.join("../..").canonicalize().expect("repository root");
    let csrc = repo.join("plonk_amd/csrc");
    println!("cargo:rerun-if-env-changed=PLONK_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=HIPCC");
    println!("cargo:rerun-if-changed={}", repo.join("include/plonk_hip.h").display());

    if let Ok(dir) = env::var("PLONK_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-lib=dylib=plonk_hip");
        return;
    }
    if cfg!(feature = "prebuilt") {
        println!("cargo:rustc-link-search=native={}", repo.join("plonk_amd").display());
        println!("cargo:rustc-link-lib=dylib=plonk_hip");
        return;
    }

    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let mut objects = Vec::new();
    for entry in std::fs::read_dir(&csrc).expect("plonk_amd/csrc") {
        let path = entry.unwrap().path();
        if path.extension().map(|e| e == "hip").unwrap_or(false) {
            println!("cargo:rerun-if-changed={}", path.display());
            let obj = out.join(path.file_stem().unwrap()).with_extension("o");
            let status = Command::new(&hipcc)
                .args(["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c"])
                .arg(&path)
                .arg("-o")
                .arg(&obj)
                .status()
                .expect("hipcc not found (set HIPCC)");
            assert!(status.success(), "hipcc failed on {}", path.display());
            objects.push(obj);
        }
    }
    let lib = out.join("libplonk_hip.so");
    let status = Command::new(&hipcc)
        .args(["-shared", "-fPIC", "--offload-arch=gfx950"])
        .args(&objects)
        .arg("-o")
        .arg(&lib)
        .status()
        .expect("hipcc not found (set HIPCC)");
    assert!(status.success(), "linking libplonk_hip.so failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=plonk_hip");
}
