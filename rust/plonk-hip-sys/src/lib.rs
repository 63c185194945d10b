//! FFI to `libplonk_hip.so` (the C ABI of `include/plonk_hip.h`) and safe wrappers shaped like the
//! functions of the reference's `backend_new` crate they stand in for:
//! `fft` / `ifft` / `coset_fft` / `coset_ifft` (src/fft.rs:23,42,121,140), `Srs` + `commit`
//! (src/kzg.rs:26-35,231-245), `ProvingKey` (plonk/preprocess.rs:104), `Prover::new` / `prove`
//! (plonk/prover.rs:39,56) and `Verifier::verify` (plonk/verifier.rs:36).
//!
//! Field elements and points cross the boundary as raw Montgomery limbs -- the in-memory image of
//! `ark_bls12_381::Fr` (`.0.0: [u64; 4]`) and of `G1Affine { x, y, infinity }` -- so this crate has no
//! dependency on arkworks; a caller on arkworks 0.4 copies `fr.0 .0` into `PlonkFr { l }` and back.
//!
//! Source only: the environment this repository is built in has no Rust toolchain.  The `extern "C"`
//! block below is checked against the header symbol by symbol by `tests/test_abi.py`.
#![allow(clippy::too_many_arguments)]

use std::ffi::{c_char, c_void, CStr};
use std::os::raw::c_int;
use std::ptr;

pub const PLONK_OK: i32 = 0;
pub const PLONK_E_INVALID: i32 = -1;
pub const PLONK_E_HIP: i32 = -2;
pub const PLONK_E_NOMEM: i32 = -3;
pub const PLONK_E_INTERNAL: i32 = -4;
pub const PLONK_NTT_INVERSE: u32 = 1;
pub const PLONK_NTT_COSET: u32 = 2;
pub const PLONK_PROOF_WIRE_BYTES: usize = 624;
pub const PLONK_COMM_ID_BYTES: usize = 128;

#[repr(C)]
pub struct PlonkCtx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct PlonkSrs {
    _private: [u8; 0],
}
#[repr(C)]
pub struct PlonkPk {
    _private: [u8; 0],
}
#[repr(C)]
pub struct PlonkComm {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct PlonkFr {
    pub l: [u64; 4],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct PlonkG1 {
    pub x: [u64; 6],
    pub y: [u64; 6],
    pub infinity: u8,
    pub pad: [u8; 7],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct PlonkG2 {
    pub x_c0: [u64; 6],
    pub x_c1: [u64; 6],
    pub y_c0: [u64; 6],
    pub y_c1: [u64; 6],
    pub infinity: u8,
    pub pad: [u8; 7],
}

/// `Proof` of plonk/types.rs:74-114, same field order.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct PlonkProof {
    pub a_comm: PlonkG1,
    pub b_comm: PlonkG1,
    pub c_comm: PlonkG1,
    pub z_comm: PlonkG1,
    pub t_lo_comm: PlonkG1,
    pub t_mid_comm: PlonkG1,
    pub t_hi_comm: PlonkG1,
    pub a_eval: PlonkFr,
    pub b_eval: PlonkFr,
    pub c_eval: PlonkFr,
    pub s_sigma1_eval: PlonkFr,
    pub s_sigma2_eval: PlonkFr,
    pub z_omega_eval: PlonkFr,
    pub w_zeta_comm: PlonkG1,
    pub w_zeta_omega_comm: PlonkG1,
}

/// `VerificationKey` of plonk/types.rs:42-69 (public_input_positions travel beside it).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct PlonkVk {
    pub n: u64,
    pub q_m_comm: PlonkG1,
    pub q_l_comm: PlonkG1,
    pub q_r_comm: PlonkG1,
    pub q_o_comm: PlonkG1,
    pub q_c_comm: PlonkG1,
    pub s_sigma1_comm: PlonkG1,
    pub s_sigma2_comm: PlonkG1,
    pub s_sigma3_comm: PlonkG1,
    pub omega: PlonkFr,
    pub k1: PlonkFr,
    pub k2: PlonkFr,
}

/// `plonk_shard_slices`: a rank's SRS slices for rounds committing one, two and three polynomials
/// (`[1]`, `[2]` optional: null).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PlonkShardSlices {
    pub srs: [*const PlonkSrs; 3],
    pub lo: [u64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PlonkMsmTiming {
    pub sort_ms: f32,
    pub big_bucket_ms: f32,
    pub bucket_ms: f32,
    pub reduce_ms: f32,
    pub entries: u64,
    pub window_bits: u32,
    pub bucket_sets: u32,
}

pub type PlonkAllgatherFn =
    Option<unsafe extern "C" fn(user: *mut c_void, send: *const c_void, bytes: usize, recv: *mut c_void) -> i32>;

extern "C" {
    pub fn plonk_ctx_create(device: i32, out: *mut *mut PlonkCtx) -> i32;
    pub fn plonk_ctx_destroy(ctx: *mut PlonkCtx);
    pub fn plonk_last_error(ctx: *const PlonkCtx) -> *const c_char;
    pub fn plonk_sync(ctx: *mut PlonkCtx) -> i32;
    pub fn plonk_ctx_stream(ctx: *mut PlonkCtx) -> *mut c_void;
    pub fn plonk_dev_alloc(ctx: *mut PlonkCtx, bytes: usize, out_dev: *mut *mut c_void) -> i32;
    pub fn plonk_dev_free(ctx: *mut PlonkCtx, dev: *mut c_void) -> i32;
    pub fn plonk_dev_upload(ctx: *mut PlonkCtx, dst_dev: *mut c_void, src_host: *const c_void, bytes: usize) -> i32;
    pub fn plonk_dev_download(ctx: *mut PlonkCtx, dst_host: *mut c_void, src_dev: *const c_void, bytes: usize) -> i32;
    pub fn plonk_ntt_dev(ctx: *mut PlonkCtx, data_dev: *mut PlonkFr, log_n: u32, flags: u32, shift: *const PlonkFr,
                         batch: u32, stride: usize) -> i32;
    pub fn plonk_ntt_sweeps(log_n: u32) -> u32;
    pub fn plonk_ntt(ctx: *mut PlonkCtx, data_host: *mut PlonkFr, log_n: u32, flags: u32, shift: *const PlonkFr,
                     batch: u32, stride: usize) -> i32;
    pub fn plonk_ntt_dev_noalloc(ctx: *mut PlonkCtx, data_dev: *mut PlonkFr, log_n: u32, flags: u32,
                                 shift: *const PlonkFr) -> i32;
    pub fn plonk_ntt_noalloc_max_log_n() -> u32;
    pub fn plonk_srs_upload(ctx: *mut PlonkCtx, g1_powers: *const PlonkG1, n: usize, out: *mut *mut PlonkSrs) -> i32;
    pub fn plonk_srs_from_tau(ctx: *mut PlonkCtx, tau: *const PlonkFr, n: usize, out: *mut *mut PlonkSrs) -> i32;
    pub fn plonk_srs_from_tau_range(ctx: *mut PlonkCtx, tau: *const PlonkFr, first: usize, count: usize,
                                    out: *mut *mut PlonkSrs) -> i32;
    pub fn plonk_srs_len(srs: *const PlonkSrs) -> usize;
    pub fn plonk_srs_download(ctx: *mut PlonkCtx, srs: *const PlonkSrs, first: usize, count: usize, out: *mut PlonkG1) -> i32;
    pub fn plonk_srs_free(srs: *mut PlonkSrs);
    pub fn plonk_commit(ctx: *mut PlonkCtx, srs: *const PlonkSrs, coeffs_host: *const PlonkFr, len: usize,
                        out: *mut PlonkG1) -> i32;
    pub fn plonk_commit_dev(ctx: *mut PlonkCtx, srs: *const PlonkSrs, coeffs_dev: *const PlonkFr, len: usize,
                            out: *mut PlonkG1) -> i32;
    pub fn plonk_msm(ctx: *mut PlonkCtx, points_host: *const PlonkG1, scalars_host: *const PlonkFr, n: usize,
                     out: *mut PlonkG1) -> i32;
    pub fn plonk_poly_eval_dev(ctx: *mut PlonkCtx, coeffs_dev: *const PlonkFr, len: usize, z: *const PlonkFr,
                               y_out: *mut PlonkFr) -> i32;
    pub fn plonk_div_by_linear_dev(ctx: *mut PlonkCtx, coeffs_dev: *const PlonkFr, len: usize, z: *const PlonkFr,
                                   q_out_dev: *mut PlonkFr) -> i32;
    pub fn plonk_open_dev(ctx: *mut PlonkCtx, srs: *const PlonkSrs, coeffs_dev: *const PlonkFr, len: usize, z: *const PlonkFr,
                          y_out: *mut PlonkFr, proof_out: *mut PlonkG1) -> i32;
    pub fn plonk_open_batch_dev(ctx: *mut PlonkCtx, srs: *const PlonkSrs, polys_dev: *const *const PlonkFr, lens: *const usize,
                                count: usize, z: *const PlonkFr, gamma: *const PlonkFr, evals_out: *mut PlonkFr,
                                proof_out: *mut PlonkG1) -> i32;
    pub fn plonk_pk_create(ctx: *mut PlonkCtx, log_n: u32, q_m: *const PlonkFr, q_l: *const PlonkFr, q_r: *const PlonkFr,
                           q_o: *const PlonkFr, q_c: *const PlonkFr, s_sigma1: *const PlonkFr, s_sigma2: *const PlonkFr,
                           s_sigma3: *const PlonkFr, k1: *const PlonkFr, k2: *const PlonkFr, pi_positions: *const u32,
                           n_pi_positions: usize, out: *mut *mut PlonkPk) -> i32;
    pub fn plonk_pk_free(pk: *mut PlonkPk);
    pub fn plonk_pk_commitments(ctx: *mut PlonkCtx, pk: *const PlonkPk, srs: *const PlonkSrs, out: *mut PlonkG1) -> i32;
    pub fn plonk_pk_poly(ctx: *mut PlonkCtx, pk: *const PlonkPk, which: u32, out_host: *mut PlonkFr, len: *mut usize) -> i32;
    pub fn plonk_quotient_3wire_dev(ctx: *mut PlonkCtx, pk: *const PlonkPk, a_dev: *mut PlonkFr, b_dev: *mut PlonkFr,
                                    c_dev: *mut PlonkFr, z_dev: *mut PlonkFr, pi_dev: *mut PlonkFr, alpha: *const PlonkFr,
                                    beta: *const PlonkFr, gamma: *const PlonkFr, t_out_dev: *mut PlonkFr) -> i32;
    pub fn plonk_prove_3wire(ctx: *mut PlonkCtx, pk: *const PlonkPk, srs: *const PlonkSrs, a_evals: *const PlonkFr,
                             b_evals: *const PlonkFr, c_evals: *const PlonkFr, public_inputs: *const PlonkFr,
                             n_public_inputs: usize, blinding: *const PlonkFr, out: *mut PlonkProof) -> i32;
    pub fn plonk_prove_3wire_dev(ctx: *mut PlonkCtx, pk: *const PlonkPk, srs: *const PlonkSrs, a_evals_dev: *const PlonkFr,
                                 b_evals_dev: *const PlonkFr, c_evals_dev: *const PlonkFr, public_inputs: *const PlonkFr,
                                 n_public_inputs: usize, blinding: *const PlonkFr, out: *mut PlonkProof) -> i32;
    pub fn plonk_comm_unique_id(out: *mut u8) -> i32;
    pub fn plonk_comm_create_rccl(ctx: *mut PlonkCtx, id: *const u8, rank: u32, world: u32, out: *mut *mut PlonkComm) -> i32;
    pub fn plonk_comm_create_callback(ctx: *mut PlonkCtx, rank: u32, world: u32, allgather: PlonkAllgatherFn,
                                      user: *mut c_void, out: *mut *mut PlonkComm) -> i32;
    pub fn plonk_comm_create_loopback(ctx: *mut PlonkCtx, rank: u32, world: u32, total_points: usize,
                                      out: *mut *mut PlonkComm) -> i32;
    pub fn plonk_comm_destroy(comm: *mut PlonkComm);
    pub fn plonk_comm_stats(comm: *const PlonkComm, calls: *mut u64, bytes: *mut u64) -> i32;
    pub fn plonk_comm_timing(comm: *mut PlonkComm, enable: i32, elapsed_ms: *mut f64) -> i32;
    pub fn plonk_prove_3wire_sharded(ctx: *mut PlonkCtx, pk: *const PlonkPk, srs_slice: *const PlonkSrs, shard_lo: usize,
                                     comm: *mut PlonkComm, a_evals_dev: *const PlonkFr, b_evals_dev: *const PlonkFr,
                                     c_evals_dev: *const PlonkFr, public_inputs: *const PlonkFr, n_public_inputs: usize,
                                     blinding: *const PlonkFr, out: *mut PlonkProof) -> i32;
    pub fn plonk_prove_3wire_grouped(ctx: *mut PlonkCtx, pk: *const PlonkPk, slices: *const PlonkShardSlices, comm: *mut PlonkComm,
                                     a_evals_dev: *const PlonkFr, b_evals_dev: *const PlonkFr, c_evals_dev: *const PlonkFr,
                                     public_inputs: *const PlonkFr, n_public_inputs: usize, blinding: *const PlonkFr,
                                     out: *mut PlonkProof) -> i32;
    pub fn plonk_g1_sum(points: *const PlonkG1, n: usize, out: *mut PlonkG1) -> i32;
    pub fn plonk_msm_window_bits(n: usize) -> u32;
    pub fn plonk_msm_set_timing(ctx: *mut PlonkCtx, enable: i32) -> i32;
    pub fn plonk_msm_last_timing(ctx: *mut PlonkCtx, out: *mut PlonkMsmTiming) -> i32;
    pub fn plonk_quotient_last_timing(ctx: *mut PlonkCtx, elapsed_ms: *mut f64) -> i32;
    pub fn plonk_g1_compress(p: *const PlonkG1, out: *mut u8) -> i32;
    pub fn plonk_g1_decompress(input: *const u8, out: *mut PlonkG1) -> i32;
    pub fn plonk_fr_to_bytes(a: *const PlonkFr, out: *mut u8) -> i32;
    pub fn plonk_fr_from_bytes(input: *const u8, out: *mut PlonkFr) -> i32;
    pub fn plonk_proof_serialize(proof: *const PlonkProof, out: *mut u8) -> i32;
    pub fn plonk_proof_deserialize(input: *const u8, proof: *mut PlonkProof) -> i32;
    pub fn plonk_proof_challenges(proof: *const PlonkProof, public_inputs: *const PlonkFr, n_public_inputs: usize,
                                  out: *mut PlonkFr) -> i32;
    pub fn plonk_g2_generator(out: *mut PlonkG2) -> i32;
    pub fn plonk_g2_mul(p: *const PlonkG2, k: *const PlonkFr, out: *mut PlonkG2) -> i32;
    pub fn plonk_pairing_check(p1: *const PlonkG1, q1: *const PlonkG2, p2: *const PlonkG1, q2: *const PlonkG2,
                               equal: *mut i32) -> i32;
    pub fn plonk_open_verify(commitment: *const PlonkG1, z: *const PlonkFr, y: *const PlonkFr, proof: *const PlonkG1,
                             g2_tau: *const PlonkG2, accepted: *mut i32) -> i32;
    pub fn plonk_vk_from_pk(ctx: *mut PlonkCtx, pk: *const PlonkPk, srs: *const PlonkSrs, out: *mut PlonkVk) -> i32;
    pub fn plonk_verify_3wire(vk: *const PlonkVk, public_input_positions: *const u64, n_positions: usize,
                              g2_tau: *const PlonkG2, proof: *const PlonkProof, public_inputs: *const PlonkFr,
                              n_public_inputs: usize, accepted: *mut i32) -> i32;
    pub fn plonk_proof_to_json(proof: *const PlonkProof, out: *mut c_char, cap: usize, len: *mut usize) -> i32;
    pub fn plonk_proof_from_json(json: *const c_char, len: usize, proof: *mut PlonkProof) -> i32;
    pub fn plonk_synthetic_circuit(log_n: u32, a: *mut PlonkFr, b: *mut PlonkFr, c: *mut PlonkFr, q_m: *mut PlonkFr,
                                   q_l: *mut PlonkFr, q_r: *mut PlonkFr, q_o: *mut PlonkFr, q_c: *mut PlonkFr,
                                   s_sigma1: *mut PlonkFr, s_sigma2: *mut PlonkFr, s_sigma3: *mut PlonkFr,
                                   blinding: *mut PlonkFr, tau: *mut PlonkFr, k1: *mut PlonkFr, k2: *mut PlonkFr) -> i32;
    pub fn plonk_bench_field_mul(ctx: *mut PlonkCtx, which: u32, muls_per_second: *mut f64) -> i32;
    pub fn plonk_timer_start(ctx: *mut PlonkCtx) -> i32;
    pub fn plonk_timer_stop(ctx: *mut PlonkCtx, elapsed_ms: *mut f64) -> i32;
}

// ------------------------------------------------------------------------------------------------
// Safe wrappers

/// A failed call: the status code and the library's message.
#[derive(Debug, Clone)]
pub struct Error {
    pub code: i32,
    pub message: String,
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "plonk_hip error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}

pub type Result<T> = std::result::Result<T, Error>;

/// One HIP device + stream.  Not `Sync`: a context is used from one thread at a time.
pub struct Context {
    raw: *mut PlonkCtx,
}
unsafe impl Send for Context {}

impl Context {
    pub fn new(device: i32) -> Result<Self> {
        let mut raw = ptr::null_mut();
        let rc = unsafe { plonk_ctx_create(device, &mut raw) };
        if rc != PLONK_OK {
            return Err(Error { code: rc, message: format!("plonk_ctx_create(device = {device}) failed") });
        }
        Ok(Context { raw })
    }

    fn check(&self, rc: i32) -> Result<()> {
        if rc == PLONK_OK {
            return Ok(());
        }
        let message = unsafe { CStr::from_ptr(plonk_last_error(self.raw)) }.to_string_lossy().into_owned();
        Err(Error { code: rc, message })
    }

    pub fn as_ptr(&self) -> *mut PlonkCtx {
        self.raw
    }

    fn ntt(&self, data: &mut [PlonkFr], flags: u32, shift: Option<&PlonkFr>) -> Result<()> {
        assert!(data.len().is_power_of_two(), "FFT size must be a power of 2"); // fft.rs:24,44
        let shift = shift.map_or(ptr::null(), |s| s as *const PlonkFr);
        self.check(unsafe {
            plonk_ntt(self.raw, data.as_mut_ptr(), data.len().trailing_zeros(), flags, shift, 1, data.len())
        })
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { plonk_ctx_destroy(self.raw) }
    }
}

/// fft.rs:23 -- evaluations of `poly` (coefficients) on the domain of size `n`, natural order.
/// `omega` is implied by `n` (the library uses Fr::get_root_of_unity(n), field.rs:167-182).
pub fn fft(ctx: &Context, poly: &[PlonkFr], n: usize) -> Result<Vec<PlonkFr>> {
    assert!(n.is_power_of_two(), "FFT size must be a power of 2");
    assert!(poly.len() <= n, "polynomial longer than the domain");
    let mut v = poly.to_vec();
    v.resize(n, PlonkFr::default());
    ctx.ntt(&mut v, 0, None)?;
    Ok(v)
}

/// fft.rs:42 -- coefficients from evaluations, scaled by 1/n, trailing zeros stripped.
pub fn ifft(ctx: &Context, evals: &[PlonkFr]) -> Result<Vec<PlonkFr>> {
    let mut v = evals.to_vec();
    ctx.ntt(&mut v, PLONK_NTT_INVERSE, None)?;
    while v.last() == Some(&PlonkFr::default()) {
        v.pop();
    }
    Ok(v)
}

/// fft.rs:121 -- evaluations on the coset `k * H`.
pub fn coset_fft(ctx: &Context, poly: &[PlonkFr], k: &PlonkFr, n: usize) -> Result<Vec<PlonkFr>> {
    assert!(n.is_power_of_two(), "FFT size must be a power of 2");
    assert!(poly.len() <= n, "polynomial longer than the domain");
    let mut v = poly.to_vec();
    v.resize(n, PlonkFr::default());
    ctx.ntt(&mut v, PLONK_NTT_COSET, Some(k))?;
    Ok(v)
}

/// fft.rs:140 -- n coefficients (zero padded) from evaluations on the coset `k * H`.
pub fn coset_ifft(ctx: &Context, evals: &[PlonkFr], k: &PlonkFr) -> Result<Vec<PlonkFr>> {
    let mut v = evals.to_vec();
    ctx.ntt(&mut v, PLONK_NTT_COSET | PLONK_NTT_INVERSE, Some(k))?;
    Ok(v)
}

/// kzg.rs:26-35 -- `g1_powers` resident on the device.
pub struct Srs {
    raw: *mut PlonkSrs,
}
unsafe impl Send for Srs {}
unsafe impl Sync for Srs {} // immutable after creation

impl Srs {
    pub fn upload(ctx: &Context, g1_powers: &[PlonkG1]) -> Result<Self> {
        let mut raw = ptr::null_mut();
        ctx.check(unsafe { plonk_srs_upload(ctx.raw, g1_powers.as_ptr(), g1_powers.len(), &mut raw) })?;
        Ok(Srs { raw })
    }

    /// kzg.rs:71-94 for a known trapdoor (tests and benchmarks).
    pub fn from_tau(ctx: &Context, tau: &PlonkFr, max_degree: usize) -> Result<Self> {
        let mut raw = ptr::null_mut();
        ctx.check(unsafe { plonk_srs_from_tau(ctx.raw, tau, max_degree + 1, &mut raw) })?;
        Ok(Srs { raw })
    }

    pub fn max_degree(&self) -> usize {
        unsafe { plonk_srs_len(self.raw) }.saturating_sub(1)
    }

    pub fn as_ptr(&self) -> *const PlonkSrs {
        self.raw
    }
}

impl Drop for Srs {
    fn drop(&mut self) {
        unsafe { plonk_srs_free(self.raw) }
    }
}

/// kzg.rs:231-245 -- `Commitment::commit(poly, srs)`; panics like the reference when the degree exceeds the SRS.
pub fn commit(ctx: &Context, poly: &[PlonkFr], srs: &Srs) -> PlonkG1 {
    let mut out = PlonkG1::default();
    let rc = unsafe { plonk_commit(ctx.raw, srs.raw, poly.as_ptr(), poly.len(), &mut out) };
    if rc == PLONK_E_INVALID {
        panic!("Polynomial degree {} exceeds SRS max degree {}", poly.len().saturating_sub(1), srs.max_degree());
    }
    ctx.check(rc).expect("plonk_commit");
    out
}

/// curve.rs:169-173 -- `G1Projective::msm(points, scalars)`, result affine.
pub fn msm(ctx: &Context, points: &[PlonkG1], scalars: &[PlonkFr]) -> Result<PlonkG1> {
    let n = points.len().min(scalars.len());
    let mut out = PlonkG1::default();
    ctx.check(unsafe { plonk_msm(ctx.raw, points.as_ptr(), scalars.as_ptr(), n, &mut out) })?;
    Ok(out)
}

/// plonk/types.rs:12 `ProvingKey`, built on the device by what preprocess.rs:104-186 computes.
pub struct ProvingKey {
    raw: *mut PlonkPk,
    pub n: usize,
}
unsafe impl Send for ProvingKey {}
unsafe impl Sync for ProvingKey {}

impl ProvingKey {
    /// `selectors` = q_M, q_L, q_R, q_O, q_C (constraint.rs:385-393); `sigmas` = S_sigma1..3 evaluations
    /// (permutation.rs:113-126); every vector holds n = 2^k elements.
    pub fn new(ctx: &Context, selectors: [&[PlonkFr]; 5], sigmas: [&[PlonkFr]; 3], k1: &PlonkFr, k2: &PlonkFr,
               public_input_positions: &[u32]) -> Result<Self> {
        let n = selectors[0].len();
        assert!(n.is_power_of_two(), "domain size must be a power of 2");
        assert!(selectors.iter().chain(sigmas.iter()).all(|v| v.len() == n), "vectors must all have n elements");
        let mut raw = ptr::null_mut();
        ctx.check(unsafe {
            plonk_pk_create(ctx.raw, n.trailing_zeros(), selectors[0].as_ptr(), selectors[1].as_ptr(), selectors[2].as_ptr(),
                            selectors[3].as_ptr(), selectors[4].as_ptr(), sigmas[0].as_ptr(), sigmas[1].as_ptr(),
                            sigmas[2].as_ptr(), k1, k2, public_input_positions.as_ptr(), public_input_positions.len(), &mut raw)
        })?;
        Ok(ProvingKey { raw, n })
    }

    /// The verification key of preprocess.rs:130-178.
    pub fn verification_key(&self, ctx: &Context, srs: &Srs) -> Result<PlonkVk> {
        let mut vk = PlonkVk::default();
        ctx.check(unsafe { plonk_vk_from_pk(ctx.raw, self.raw, srs.raw, &mut vk) })?;
        Ok(vk)
    }
}

impl Drop for ProvingKey {
    fn drop(&mut self) {
        unsafe { plonk_pk_free(self.raw) }
    }
}

/// plonk/prover.rs:24-62.  Borrows key and SRS like the reference's `Prover<'a>`.
pub struct Prover<'a> {
    ctx: &'a Context,
    pk: &'a ProvingKey,
    srs: &'a Srs,
    wires: [&'a [PlonkFr]; 3],
}

impl<'a> Prover<'a> {
    /// `wires` = ConstraintSystem::get_wire_values() (constraint.rs:401-410).
    pub fn new(ctx: &'a Context, wires: [&'a [PlonkFr]; 3], pk: &'a ProvingKey, srs: &'a Srs) -> Self {
        assert!(wires.iter().all(|w| w.len() == pk.n), "Wire values not set");
        Prover { ctx, pk, srs, wires }
    }

    /// prover.rs:56.  `blinding` = b1..b11 in the order the reference draws them from its RNG
    /// (prover.rs:189-194, 310-312, 537-538); the caller supplies the randomness.
    pub fn prove(&self, public_inputs: &[PlonkFr], blinding: &[PlonkFr; 11]) -> Result<PlonkProof> {
        let mut out = PlonkProof::default();
        self.ctx.check(unsafe {
            plonk_prove_3wire(self.ctx.raw, self.pk.raw, self.srs.raw, self.wires[0].as_ptr(), self.wires[1].as_ptr(),
                              self.wires[2].as_ptr(), public_inputs.as_ptr(), public_inputs.len(), blinding.as_ptr(), &mut out)
        })?;
        Ok(out)
    }
}

/// plonk/verifier.rs:24-40.  Of the SRS the verifier needs `g2_tau` only (kzg.rs:34).
pub struct Verifier {
    pub vk: PlonkVk,
    pub public_input_positions: Vec<u64>,
    pub g2_tau: PlonkG2,
}

impl Verifier {
    /// verifier.rs:36.  `Err` for inputs the reference's types cannot hold (points off the curve ...).
    pub fn verify(&self, proof: &PlonkProof, public_inputs: &[PlonkFr]) -> std::result::Result<bool, i32> {
        let mut accepted: c_int = 0;
        let rc = unsafe {
            plonk_verify_3wire(&self.vk, self.public_input_positions.as_ptr(), self.public_input_positions.len(), &self.g2_tau,
                               proof, public_inputs.as_ptr(), public_inputs.len(), &mut accepted)
        };
        if rc != PLONK_OK {
            return Err(rc);
        }
        Ok(accepted != 0)
    }
}

/// serde_json's text of `CompactProof::from(&proof)` (trace.rs:255-332).
pub fn proof_to_json(proof: &PlonkProof) -> String {
    let mut buf = vec![0u8; 8192];
    let mut len = 0usize;
    let rc = unsafe { plonk_proof_to_json(proof, buf.as_mut_ptr() as *mut c_char, buf.len(), &mut len) };
    assert_eq!(rc, PLONK_OK, "plonk_proof_to_json");
    buf.truncate(len);
    String::from_utf8(buf).expect("ASCII")
}

pub fn proof_from_json(json: &str) -> Option<PlonkProof> {
    let mut out = PlonkProof::default();
    let rc = unsafe { plonk_proof_from_json(json.as_ptr() as *const c_char, json.len(), &mut out) };
    (rc == PLONK_OK).then_some(out)
}
