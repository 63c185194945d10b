/* plonk_hip.h -- C ABI of the MI355X (gfx950) PLONK prover hot path.
 *
 * Every entry point replaces one seam of the reference's `backend_new` crate (paths below are
 * relative to the reference tree).  The reference has no FFI of its own: these are the Rust
 * function signatures its protocol layer calls, restated as extern "C" with plain pointers.
 *
 * Conventions
 *  - Fr: 32 bytes = 4 x u64 little-endian limbs in Montgomery form (R = 2^256), the in-memory
 *    image of ark_bls12_381::Fr that backend_new/src/field.rs:18 wraps.
 *  - Fq: 48 bytes = 6 x u64 little-endian limbs, Montgomery form (R = 2^384).
 *  - G1 affine: plonk_g1 = x || y (Fq each) + infinity flag, the fields of ark's G1Affine that
 *    backend_new/src/curve.rs:33 wraps.
 *  - All vectors are natural order in and out (fft.rs:300-318).
 *  - Every function returns PLONK_OK (0) or a negative status and never unwinds; the message is
 *    at plonk_last_error(ctx).  Conditions that panic in the reference (fft.rs:24,44;
 *    kzg.rs:232-238; prover.rs:185,290,447) return PLONK_E_INVALID.
 *  - A ctx owns one HIP device + stream and must be used from one host thread at a time.
 *    Handles (srs, pk) are immutable after creation and may be shared by contexts on the same
 *    device.  The caller owns every host buffer; the library borrows it for the call only.
 *  - Pointers named *_dev are device pointers obtained from plonk_dev_alloc (or any hipMalloc).
 */
#ifndef PLONK_HIP_H
#define PLONK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLONK_OK 0
#define PLONK_E_INVALID (-1) /* bad size / degree / argument (reference: panic) */
#define PLONK_E_HIP (-2)     /* HIP runtime failure */
#define PLONK_E_NOMEM (-3)
#define PLONK_E_INTERNAL (-4)

#define PLONK_NTT_INVERSE 1u /* fft.rs:42 ifft (scaled by 1/n) */
#define PLONK_NTT_COSET 2u   /* fft.rs:121 coset_fft / :140 coset_ifft */

typedef struct plonk_ctx plonk_ctx;
typedef struct plonk_srs plonk_srs;
typedef struct plonk_pk plonk_pk;

typedef struct plonk_fr {
    uint64_t l[4];
} plonk_fr;

typedef struct plonk_g1 {
    uint64_t x[6];
    uint64_t y[6];
    uint8_t infinity;
    uint8_t pad[7];
} plonk_g1;

/* types.rs:74-114 `Proof`: 9 G1 commitments + 6 evaluations, in the reference's field order.
 * Evaluations are Montgomery-form Fr like every other scalar in this ABI. */
typedef struct plonk_proof {
    plonk_g1 a_comm, b_comm, c_comm, z_comm, t_lo_comm, t_mid_comm, t_hi_comm;
    plonk_fr a_eval, b_eval, c_eval, s_sigma1_eval, s_sigma2_eval, z_omega_eval;
    plonk_g1 w_zeta_comm, w_zeta_omega_comm;
} plonk_proof;

#define PLONK_PROOF_WIRE_BYTES 624 /* 9 x 48-byte compressed G1 + 6 x 32-byte canonical LE Fr */

/* ---- context ---- */
int32_t plonk_ctx_create(int32_t device, plonk_ctx** out);
void plonk_ctx_destroy(plonk_ctx* ctx);
const char* plonk_last_error(const plonk_ctx* ctx);
int32_t plonk_sync(plonk_ctx* ctx);
/* The HIP stream every kernel of this ctx is launched on (a hipStream_t). */
void* plonk_ctx_stream(plonk_ctx* ctx);

/* ---- device memory (thin wrappers so an FFI caller needs no HIP binding) ---- */
int32_t plonk_dev_alloc(plonk_ctx* ctx, size_t bytes, void** out_dev);
int32_t plonk_dev_free(plonk_ctx* ctx, void* dev);
int32_t plonk_dev_upload(plonk_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int32_t plonk_dev_download(plonk_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* ---- NTT: replaces fft / ifft / coset_fft / coset_ifft and Domain::{fft,ifft,coset_fft,
 * coset_ifft} (backend_new/src/fft.rs:23,42,121,140,220-248).
 * In place over `batch` vectors of 2^log_n Fr, vector b at data + b*stride (elements).
 * flags: PLONK_NTT_INVERSE, PLONK_NTT_COSET.  `shift` is the coset generator k (Montgomery
 * Fr); NULL means 7, the value prover.rs:367 uses.  The inverse transform scales by 1/n and,
 * with COSET, by k^-i, exactly as fft.rs:52-56,147-152.  log_n in [0, 28] (the field's two-adicity
 * would allow 32; 2^27, the 8n domain of a 2^24-gate circuit, is the largest size exercised). */
int32_t plonk_ntt_dev(plonk_ctx* ctx, plonk_fr* data_dev, uint32_t log_n, uint32_t flags,
                      const plonk_fr* shift, uint32_t batch, size_t stride);
/* Passes over HBM (kernel launches) a transform of 2^log_n elements takes; 0 for a size out of range. */
uint32_t plonk_ntt_sweeps(uint32_t log_n);
/* Same over a host buffer (upload, transform, download). */
int32_t plonk_ntt(plonk_ctx* ctx, plonk_fr* data_host, uint32_t log_n, uint32_t flags,
                  const plonk_fr* shift, uint32_t batch, size_t stride);

/* The same transform of ONE vector for callers that cannot allocate (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  It calls
 * no hipMalloc / hipFree: the intermediate image of the sweeps and the twiddle tables live in a
 * static device array of 24.4 MiB that every GPU the library runs on holds until the process ends.
 * log_n above plonk_ntt_noalloc_max_log_n() (= 18) returns PLONK_E_INVALID.  The array keeps the
 * tables of the last (log_n, flags, shift); a call with another combination regenerates them with a
 * few kernels on the ctx stream.  All calls on one GPU share that GPU's array.  The library orders them:
 * a host mutex per GPU, and each call's stream waits for an event recorded behind the previous call's last
 * kernel (whatever context or thread made it), so successive calls from different contexts need no
 * plonk_sync in between; they run one after the other on the device.  What stays the caller's duty: a
 * captured graph replays against whatever tables the array holds at replay time, so no call with another
 * (log_n, flags, shift) may run on that GPU between capture and the last replay. */
int32_t plonk_ntt_dev_noalloc(plonk_ctx* ctx, plonk_fr* data_dev, uint32_t log_n, uint32_t flags,
                              const plonk_fr* shift);
uint32_t plonk_ntt_noalloc_max_log_n(void);

/* ---- SRS + commit: replaces Srs (kzg.rs:26-35), Commitment::commit (kzg.rs:231-245) and
 * G1Projective::msm (curve.rs:169-173). ---- */
/* Upload g1_powers[0..n) (kzg.rs:28), n < 2^31.  Every point is checked against the curve equation
 * (on the device) and its coordinates against q: PLONK_E_INVALID otherwise; infinity allowed. */
int32_t plonk_srs_upload(plonk_ctx* ctx, const plonk_g1* g1_powers, size_t n, plonk_srs** out);
/* [tau^i]_1 for i in [0, n) generated on the device from a known tau (Montgomery Fr):
 * the group elements of Srs::from_tau (kzg.rs:71-94).  For synthetic benchmarks and tests. */
int32_t plonk_srs_from_tau(plonk_ctx* ctx, const plonk_fr* tau, size_t n, plonk_srs** out);
/* The slice [first, first + count) of the same sequence: one rank's share when the commitment
 * MSMs are sharded by point range over the GPUs of a node. */
int32_t plonk_srs_from_tau_range(plonk_ctx* ctx, const plonk_fr* tau, size_t first, size_t count, plonk_srs** out);
size_t plonk_srs_len(const plonk_srs* srs);
int32_t plonk_srs_download(plonk_ctx* ctx, const plonk_srs* srs, size_t first, size_t count, plonk_g1* out);
void plonk_srs_free(plonk_srs* srs);
/* commit(poly, srs): MSM of coeffs[0..len) against g1_powers[0..len), result affine.
 * len > srs length -> PLONK_E_INVALID (the reference panics, kzg.rs:232-238).  len == 0 or an
 * all-zero polynomial gives the identity. */
int32_t plonk_commit(plonk_ctx* ctx, const plonk_srs* srs, const plonk_fr* coeffs_host, size_t len,
                     plonk_g1* out);
int32_t plonk_commit_dev(plonk_ctx* ctx, const plonk_srs* srs, const plonk_fr* coeffs_dev, size_t len,
                         plonk_g1* out);
/* G1Projective::msm(points, scalars) (curve.rs:169-173) over host arrays, result affine. */
int32_t plonk_msm(plonk_ctx* ctx, const plonk_g1* points_host, const plonk_fr* scalars_host, size_t n,
                  plonk_g1* out);

/* ---- polynomial evaluation, division and KZG openings on the device: the seams of polynomial.rs and kzg.rs that
 * rounds 4 and 5 of the prover (and any caller that opens a committed polynomial) go through.  Coefficient
 * vectors are device-resident Montgomery Fr, low degree first; trailing zeros are allowed (the reference's
 * Polynomial strips them, which changes no value here). ---- */
/* Polynomial::evaluate (polynomial.rs:103-113): *y_out = sum coeffs[i] z^i; 0 for len == 0. */
int32_t plonk_poly_eval_dev(plonk_ctx* ctx, const plonk_fr* coeffs_dev, size_t len, const plonk_fr* z, plonk_fr* y_out);
/* Polynomial::div_by_linear (polynomial.rs:188-204): q_out_dev[0 .. len-1) = quotient of the division by (X - z),
 * the remainder f(z) dropped as the reference's synthetic division drops it.  len < 2 writes nothing (the empty
 * quotient).  q_out_dev must not overlap coeffs_dev. */
int32_t plonk_div_by_linear_dev(plonk_ctx* ctx, const plonk_fr* coeffs_dev, size_t len, const plonk_fr* z,
                                plonk_fr* q_out_dev);
/* OpeningProof::create (kzg.rs:262-276): *y_out = f(z), *proof_out = commit((f - y) / (X - z)).  PLONK_E_INVALID when
 * the quotient is longer than the SRS (the reference panics, kzg.rs:232-238). */
int32_t plonk_open_dev(plonk_ctx* ctx, const plonk_srs* srs, const plonk_fr* coeffs_dev, size_t len, const plonk_fr* z,
                       plonk_fr* y_out, plonk_g1* proof_out);
/* BatchOpening::create (kzg.rs:322-341): evals_out[i] = f_i(z) and the opening at z of g = sum_i gamma^i f_i. */
int32_t plonk_open_batch_dev(plonk_ctx* ctx, const plonk_srs* srs, const plonk_fr* const* polys_dev, const size_t* lens,
                             size_t count, const plonk_fr* z, const plonk_fr* gamma, plonk_fr* evals_out,
                             plonk_g1* proof_out);

/* ---- proving key: replaces preprocess (plonk/preprocess.rs:104-186) and ProvingKey
 * (plonk/types.rs:12).  Inputs are the per-gate selector vectors (constraint.rs:385-393) and the
 * S_sigma evaluation vectors of Permutation::compute (permutation.rs:113-126), each n = 2^log_n
 * Montgomery Fr on the host, log_n in [2, 26].  k1, k2 are the coset generators (permutation.rs:61-62).
 * pi_positions are ConstraintSystem::public_input_positions.  The key keeps, resident on the
 * device, the nine coefficient vectors (q_M..q_C, S_sigma1..3, L1) and their evaluations on
 * the 4n coset 7*H (what prover.rs:380-392 recomputes for every proof). */
int32_t plonk_pk_create(plonk_ctx* ctx, uint32_t log_n, const plonk_fr* q_m, const plonk_fr* q_l,
                        const plonk_fr* q_r, const plonk_fr* q_o, const plonk_fr* q_c,
                        const plonk_fr* s_sigma1, const plonk_fr* s_sigma2, const plonk_fr* s_sigma3,
                        const plonk_fr* k1, const plonk_fr* k2, const uint32_t* pi_positions,
                        size_t n_pi_positions, plonk_pk** out);
void plonk_pk_free(plonk_pk* pk);
/* The eight verification-key commitments of preprocess.rs:130-138, in the order
 * q_m, q_l, q_r, q_o, q_c, s_sigma1, s_sigma2, s_sigma3. */
int32_t plonk_pk_commitments(plonk_ctx* ctx, const plonk_pk* pk, const plonk_srs* srs, plonk_g1 out[8]);
/* Coefficients (normalised length returned in *len) of one key polynomial, index as above
 * plus 8 = L1; for tests. */
int32_t plonk_pk_poly(plonk_ctx* ctx, const plonk_pk* pk, uint32_t which, plonk_fr* out_host, size_t* len);

/* ---- round-3 quotient: replaces the body of Prover::round3 (plonk/prover.rs:364-457 and the
 * coset iNTT at :497).  a, b, c, z are blinded coefficient vectors on the device, zero-padded
 * to 4n Fr each; pi is the public-input polynomial (n coefficients, zero-padded to 4n) or NULL
 * when there are no public inputs.  The inputs are overwritten with their coset evaluations.
 * t_out_dev receives the 4n coefficients of t(X). */
int32_t plonk_quotient_3wire_dev(plonk_ctx* ctx, const plonk_pk* pk, plonk_fr* a_dev, plonk_fr* b_dev,
                                 plonk_fr* c_dev, plonk_fr* z_dev, plonk_fr* pi_dev, const plonk_fr* alpha,
                                 const plonk_fr* beta, const plonk_fr* gamma, plonk_fr* t_out_dev);

/* ---- whole prover: replaces Prover::new + Prover::prove (plonk/prover.rs:39-62).
 * a/b/c_evals: ConstraintSystem::get_wire_values (constraint.rs:401-410), n Montgomery Fr each.
 * blinding[11] = b1..b11 in the order the reference draws them from its RNG (prover.rs:189-194,
 * 310-312, 537-538); the caller supplies the randomness. */
int32_t plonk_prove_3wire(plonk_ctx* ctx, const plonk_pk* pk, const plonk_srs* srs, const plonk_fr* a_evals,
                          const plonk_fr* b_evals, const plonk_fr* c_evals, const plonk_fr* public_inputs,
                          size_t n_public_inputs, const plonk_fr blinding[11], plonk_proof* out);

/* Same with the three wire-evaluation vectors already resident on the device. */
int32_t plonk_prove_3wire_dev(plonk_ctx* ctx, const plonk_pk* pk, const plonk_srs* srs, const plonk_fr* a_evals_dev,
                              const plonk_fr* b_evals_dev, const plonk_fr* c_evals_dev, const plonk_fr* public_inputs,
                              size_t n_public_inputs, const plonk_fr blinding[11], plonk_proof* out);

/* ---- multi-GPU: one process per GPU, the nine commitment MSMs sharded by point range.  No
 * counterpart in the single-process reference.
 * A plonk_comm is this rank's end of an all-gather among the `world` ranks of a proof.  The RCCL
 * form exchanges device buffers over xGMI (ncclAllGather on the ctx stream): rank 0 obtains an id
 * with plonk_comm_unique_id, the launcher hands its bytes to every rank, and all ranks call
 * plonk_comm_create_rccl (a collective call; one GPU per rank).  The callback form stages each record
 * through host memory and calls `allgather(user, send, bytes, recv)`, which must fill recv with the
 * `bytes`-sized records of all ranks in rank order and return 0; it exists to rehearse the
 * multi-rank path where RCCL cannot run (several ranks sharing one GPU, CPU-side collectives). */
#define PLONK_COMM_ID_BYTES 128
typedef struct plonk_comm plonk_comm;
typedef int32_t (*plonk_allgather_fn)(void* user, const void* send, size_t bytes, void* recv);
int32_t plonk_comm_unique_id(uint8_t out[PLONK_COMM_ID_BYTES]);
int32_t plonk_comm_create_rccl(plonk_ctx* ctx, const uint8_t id[PLONK_COMM_ID_BYTES], uint32_t rank, uint32_t world,
                               plonk_comm** out);
int32_t plonk_comm_create_callback(plonk_ctx* ctx, uint32_t rank, uint32_t world, plonk_allgather_fn allgather,
                                   void* user, plonk_comm** out);
/* Measurement aid: one rank of a `world`-rank proof running alone on its GPU.  Its all-gathers return
 * copies of the rank's own record in every slot (device-to-device), and the balanced splits of
 * `total_points` (the first total_points % parts members one point more) for the coverage check: the
 * proof is NOT valid, the rank's device work is what it would be among real peers. */
int32_t plonk_comm_create_loopback(plonk_ctx* ctx, uint32_t rank, uint32_t world, size_t total_points, plonk_comm** out);
void plonk_comm_destroy(plonk_comm* comm);
/* All-gathers issued through this handle and the bytes this rank contributed to them. */
int32_t plonk_comm_stats(const plonk_comm* comm, uint64_t* calls, uint64_t* bytes);
/* Exchange timing (no counterpart in the single-process reference; a measurement aid for multi-GPU runs).
 * Returns in `elapsed_ms` (nullable) the stream time this rank's all-gathers took since the previous call --
 * the wait for the slowest peer included -- measured by an event pair around each of them while timing is on;
 * then switches timing on (enable != 0) or off.  Call it with the proofs' streams drained (after plonk_sync). */
int32_t plonk_comm_timing(plonk_comm* comm, int32_t enable, double* elapsed_ms);
/* Prover::prove with the commitments sharded: `srs_slice` holds g1_powers[shard_lo, shard_lo + len)
 * (plonk_srs_from_tau_range or an upload of that slice) and the ranks' slices must tile the point
 * range in rank order (checked; PLONK_E_INVALID otherwise).  Every rank passes the same witness and
 * key and makes the same call; the ranks' shares of every MSM (the window sums of its reduction)
 * are all-gathered on the device and combined by every rank, so all ranks obtain the same proof.
 * With a power-of-two number of ranks (2, 4, 8) round 3 (prover.rs:339-574) is shared by evaluation point
 * (each rank one residue class of the 4n coset; the quotient slices all-gathered); with two ranks and more
 * the evaluations of round 4 (:577-641) are sums over the ranks' coefficient ranges, and each rank builds and
 * divides the opening numerators of round 5 (:644-837) only on the slice it commits (partial sums all-gathered). */
/* A sharded proof is a sequence of collectives (thirteen all-gathers with two or more ranks): a rank whose call returns anything but PLONK_OK
 * has left that sequence while its peers may already be queued in the next all-gather, so after a failed sharded
 * call the plonk_comm of EVERY rank must be destroyed and a new one created before further proofs (failures
 * that depend on the inputs all ranks share -- a polynomial longer than the SRS, an unsatisfied witness, slices
 * that do not tile the point range -- are detected on every rank alike and leave the ranks in step). */
int32_t plonk_prove_3wire_sharded(plonk_ctx* ctx, const plonk_pk* pk, const plonk_srs* srs_slice, size_t shard_lo,
                                  plonk_comm* comm, const plonk_fr* a_evals_dev, const plonk_fr* b_evals_dev,
                                  const plonk_fr* c_evals_dev, const plonk_fr* public_inputs, size_t n_public_inputs,
                                  const plonk_fr blinding[11], plonk_proof* out);
/* The same with the commitments of a round placed by groups of ranks.  srs[0] / lo[0] is the rank's slice as in
 * plonk_prove_3wire_sharded (used by the rounds that commit one polynomial: all ranks tile the range).  srs[1]
 * and srs[2], each optional (NULL) but present on all ranks or on none, are its slices for the rounds that
 * commit two (the openings) and three (the wires; the quotient parts) polynomials at once: commitment k of such
 * a round is computed by the ranks r with r % b == k alone, whose slices must tile the point range among them
 * (checked) -- each rank runs one MSM over 1/ceil(world / b) of the points instead of b MSMs over 1/world.
 * For circuits whose shares would be latency-bound; needs world >= b.  No counterpart in the single-process
 * reference (prover.rs:181-244, :545-552, :819-820 commit one polynomial after the other). */
typedef struct plonk_shard_slices {
    const plonk_srs* srs[3];
    uint64_t lo[3];
} plonk_shard_slices;
int32_t plonk_prove_3wire_grouped(plonk_ctx* ctx, const plonk_pk* pk, const plonk_shard_slices* slices, plonk_comm* comm,
                                  const plonk_fr* a_evals_dev, const plonk_fr* b_evals_dev, const plonk_fr* c_evals_dev,
                                  const plonk_fr* public_inputs, size_t n_public_inputs, const plonk_fr blinding[11],
                                  plonk_proof* out);
/* Sum of n affine points on the host (combining per-rank partial commitments). */
int32_t plonk_g1_sum(const plonk_g1* points, size_t n, plonk_g1* out);
/* Window width c (bits) the MSM uses for n points: W = ceil(256/c) windows of 2^(c-1) buckets. */
uint32_t plonk_msm_window_bits(size_t n);

/* Device-time breakdown of the last timed plonk_commit / plonk_commit_dev on this context
 * (measurement aid, no counterpart in the reference): the sort of the (point, sign) entries by
 * bucket, the bucket accumulation kernel proper (one lane per bucket), the accumulation of
 * oversized buckets that follows it (one workgroup each), and the reduction of the bucket sums
 * including the copy of the result to the host; `entries` = number of mixed additions the bucket
 * kernels performed. */
typedef struct plonk_msm_timing {
    float sort_ms, big_bucket_ms, bucket_ms, reduce_ms;
    uint64_t entries;
    uint32_t window_bits, bucket_sets;
} plonk_msm_timing;
/* Timing marks cost extra events on the MSM streams, so they are recorded only between
 * plonk_msm_set_timing(ctx, 1) and plonk_msm_set_timing(ctx, 0). */
int32_t plonk_msm_set_timing(plonk_ctx* ctx, int32_t enable);
int32_t plonk_msm_last_timing(plonk_ctx* ctx, plonk_msm_timing* out);
/* Device time of the fused quotient kernel (the loop of prover.rs:415-457 alone, without the
 * transforms around it) of the last plonk_quotient_3wire_dev / plonk_prove_3wire* call made on this
 * context while plonk_msm_set_timing was on. */
int32_t plonk_quotient_last_timing(plonk_ctx* ctx, double* elapsed_ms);

/* ---- encodings (host only, no device work) ---- */
/* G1Affine::to_compressed_bytes (curve.rs:84-88). */
int32_t plonk_g1_compress(const plonk_g1* p, uint8_t out[48]);
/* G1Affine::from_compressed_bytes (curve.rs:90-93; what the Deserialize impl of curve.rs:436-450
 * reads a point from): PLONK_E_INVALID for a cleared compression bit, x >= q, an x with no point
 * on the curve, or a point outside the order-r subgroup. */
int32_t plonk_g1_decompress(const uint8_t in[48], plonk_g1* out);
/* Fr::to_bytes (field.rs:106-114): canonical little-endian. */
int32_t plonk_fr_to_bytes(const plonk_fr* a, uint8_t out[32]);
/* Fr::from_bytes (field.rs:117-125): PLONK_E_INVALID when the integer is >= r. */
int32_t plonk_fr_from_bytes(const uint8_t in[32], plonk_fr* out);
/* 624-byte wire image: a,b,c,z,t_lo,t_mid,t_hi compressed, six evaluations, w_zeta,
 * w_zeta_omega compressed (the field order of trace.rs:255-332 CompactProof). */
int32_t plonk_proof_serialize(const plonk_proof* proof, uint8_t out[PLONK_PROOF_WIRE_BYTES]);
/* Inverse of plonk_proof_serialize (each point through plonk_g1_decompress, each scalar through
 * plonk_fr_from_bytes); PLONK_E_INVALID if any element is rejected, *proof untouched then. */
int32_t plonk_proof_deserialize(const uint8_t in[PLONK_PROOF_WIRE_BYTES], plonk_proof* proof);
/* Transcript::squeeze_challenge chain of the prover (transcript.rs:39-131), exposed so callers
 * can re-derive beta, gamma, alpha, zeta, v, u from a proof: out[6] Montgomery Fr. */
int32_t plonk_proof_challenges(const plonk_proof* proof, const plonk_fr* public_inputs, size_t n_public_inputs,
                               plonk_fr out[6]);

/* ---- verifier and JSON wire format (host only, no device work) ---- */
/* ark's G2Affine (curve.rs:40-48): x, y in Fq2 = c0 + c1 u, Montgomery limbs, + infinity flag. */
typedef struct plonk_g2 {
    uint64_t x_c0[6], x_c1[6], y_c0[6], y_c1[6];
    uint8_t infinity;
    uint8_t pad[7];
} plonk_g2;
/* VerificationKey (plonk/types.rs:42-69); public_input_positions travel beside it as an array. */
typedef struct plonk_vk {
    uint64_t n;
    plonk_g1 q_m_comm, q_l_comm, q_r_comm, q_o_comm, q_c_comm, s_sigma1_comm, s_sigma2_comm, s_sigma3_comm;
    plonk_fr omega, k1, k2;
} plonk_vk;
/* G2Affine::generator (curve.rs:250-254) and scalar multiplication in G2 -- what kzg.rs:86 needs for
 * the verifier's half of the SRS, g2_tau = g2_generator * tau. */
int32_t plonk_g2_generator(plonk_g2* out);
int32_t plonk_g2_mul(const plonk_g2* p, const plonk_fr* k, plonk_g2* out);
/* pairing_check (curve.rs:408-412): *equal = [e(p1, q1) == e(p2, q2)].  PLONK_E_INVALID for a point that is off its
 * curve, has an unreduced coordinate or (G2) lies outside the order-r subgroup. */
int32_t plonk_pairing_check(const plonk_g1* p1, const plonk_g2* q1, const plonk_g1* p2, const plonk_g2* q2, int32_t* equal);
/* OpeningProof::verify (kzg.rs:282-303), host only: *accepted = [e([f]_1 - y [1]_1, [1]_2) == e(proof, [tau]_2 - z [1]_2)].
 * PLONK_E_INVALID for points the reference's types cannot hold or unreduced scalars. */
int32_t plonk_open_verify(const plonk_g1* commitment, const plonk_fr* z, const plonk_fr* y, const plonk_g1* proof,
                          const plonk_g2* g2_tau, int32_t* accepted);
/* The verification key of preprocess (preprocess.rs:130-178) from a device-resident proving key:
 * the eight commitments (computed on the device), n, omega, k1, k2. */
int32_t plonk_vk_from_pk(plonk_ctx* ctx, const plonk_pk* pk, const plonk_srs* srs, plonk_vk* out);
/* Verifier::verify (plonk/verifier.rs:36-343) with the reference's pairing check (:309-314) against
 * g2_tau = [tau]_2 (kzg.rs:34): *accepted = 1 or 0.  PLONK_E_INVALID for inputs the reference's types
 * cannot hold: a point off the curve or outside the order-r subgroup, an unreduced scalar, n not a
 * power of two, a position >= n. */
int32_t plonk_verify_3wire(const plonk_vk* vk, const uint64_t* public_input_positions, size_t n_positions,
                           const plonk_g2* g2_tau, const plonk_proof* proof, const plonk_fr* public_inputs,
                           size_t n_public_inputs, int32_t* accepted);
/* serde_json::to_string(&CompactProof::from(&proof)) (trace.rs:255-332; points as
 * {"compressed","x","y"} curve.rs:418-434, scalars as {"full","short"} field.rs:294-308).  Writes at
 * most cap bytes (no terminator) and the full length to *len; PLONK_E_INVALID if cap is too small. */
int32_t plonk_proof_to_json(const plonk_proof* proof, char* out, size_t cap, size_t* len);
/* serde_json::from_str::<CompactProof> then Proof: reads `compressed` of every point (validated as
 * plonk_g1_decompress does) and `full` of every scalar; PLONK_E_INVALID on malformed input. */
int32_t plonk_proof_from_json(const char* json, size_t len, plonk_proof* proof);

/* ---- benchmark input (host only, no device work; no counterpart in the reference): the synthetic
 * 2^log_n-gate circuit the throughput figures are quoted on -- alternating a*b = c / a+b = c gates
 * chained by the copy constraints a[i+1] = c[i], values from splitmix64 streams seeded by log_n.
 * Fills n = 2^log_n Montgomery Fr per vector: wire values, the five selector vectors
 * (constraint.rs:385-393), the S_sigma evaluation vectors Permutation::compute (permutation.rs:
 * 55-140) yields for these copy constraints; and the eleven blinding scalars, the SRS trapdoor and
 * the coset generators k1, k2 (permutation.rs:307-326).  log_n in [2, 26]. */
int32_t plonk_synthetic_circuit(uint32_t log_n, plonk_fr* a, plonk_fr* b, plonk_fr* c, plonk_fr* q_m, plonk_fr* q_l,
                                plonk_fr* q_r, plonk_fr* q_o, plonk_fr* q_c, plonk_fr* s_sigma1, plonk_fr* s_sigma2,
                                plonk_fr* s_sigma3, plonk_fr blinding[11], plonk_fr* tau, plonk_fr* k1, plonk_fr* k2);

/* ---- measurement helper: achieved Fr / Fq Montgomery multiplications per second of a
 * dependency-chain-free kernel on this device (the integer-VALU ceiling the MSM and NTT are
 * priced against).  which: 0 = Fr, 1 = Fq in the 29-bit-limb lazy form the kernels multiply in;
 * 2 = Fr, 3 = Fq in the saturated 32-bit-limb form; 4 = the raw issue rate of v_mad_u64_u32 in
 * lane-operations per second (a 29-bit-limb Fr / Fq product is at least 162 / 392 of them); 5 = Fq products
 * with one Karatsuba level on the product half (a measured alternative the kernels do not use); 6 = Fq
 * squarings; 7 = Fq products in the proportions of one mixed point addition (7 products, 2 squarings, one
 * two-product reduction per 10), the ceiling that applies to the MSM's bucket accumulation. */
int32_t plonk_bench_field_mul(plonk_ctx* ctx, uint32_t which, double* muls_per_second);
/* HIP-event stopwatch on the ctx stream (the stream every kernel of this ctx runs on): start
 * records an event, stop records a second one, waits for it and returns the elapsed device
 * time between the two in milliseconds. */
int32_t plonk_timer_start(plonk_ctx* ctx);
int32_t plonk_timer_stop(plonk_ctx* ctx, double* elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* PLONK_HIP_H */
