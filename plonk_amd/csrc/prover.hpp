// Host-side declarations of the proving key and prover (prover.hip).
#pragma once
#include <vector>

#include "common.hpp"
#include "fieldx.cuh"
#include "msm.hpp"

namespace plonk {

// ProvingKey of backend_new/src/plonk/types.rs:12, resident on one device.
struct Pk {
    uint64_t id = 0;      // unique per key for the life of the process (an address can be reused after plonk_pk_destroy)
    int device = 0;
    uint32_t log_n = 0;
    size_t n = 0;
    host::fr_h k1, k2, omega;
    std::vector<uint32_t> pi_positions;
    DevBuf<fr_t> coef;    // [9][n]  coefficients: q_m q_l q_r q_o q_c s1 s2 s3 l1
    DevBuf<fr_t> ext;     // [9][4n] evaluations on the coset 7*H_4n; rows q_m / q_l q_r q_o times 2^10 / 2^5
    DevBuf<fr_t> sig_ev;  // [3][n]  S_sigma evaluations on H
    DevBuf<fr_t> w_lo, w_hi, x_lo;
    DevBuf<frx_slot> x_hi;   // true values of the 29-bit-limb form (quotient kernel)
    uint32_t log_lw = 0, log_lx = 0;
    host::fr_h zh_inv[4];
};

// Point-range sharding of the commitment MSMs over the ranks of one node: this rank's SRS handle
// holds g1_powers[lo, lo + srs->n); `ex` all-gathers the ranks' shares of every MSM on the device.
struct ShardInfo {
    size_t lo = 0;
    Exchange* ex = nullptr;
    size_t global_end = 0;   // one past the last SRS point any rank holds (set by the prover)
    // every rank's (first point, points) for its plain slice and its slices for b = 2 and 3, six words per rank
    // (set by the prover from the coverage check: a rank knows which coefficients every other rank works on)
    std::vector<uint64_t> slices;
    // Optional placement by groups: group_srs[b - 2] is this rank's slice for the rounds that commit b = 2 or 3
    // polynomials at once.  Commitment k of such a round is then computed by the ranks r with r % b == k alone,
    // whose slices tile the point range among THEM: each rank runs one larger MSM instead of b small ones.
    const Srs* group_srs[2] = {nullptr, nullptr};
    size_t group_lo[2] = {0, 0};
};

Pk* pk_create(Ctx* ctx, uint32_t log_n, const plonk_fr* const sel[5], const plonk_fr* const sig[3], const host::fr_h& k1,
              const host::fr_h& k2, const uint32_t* pi_pos, size_t n_pi_pos);
void quotient_3wire(Ctx* ctx, const Pk* pk, fr_t* a, fr_t* b, fr_t* c, fr_t* z, fr_t* pi, const host::fr_h& alpha,
                    const host::fr_h& beta, const host::fr_h& gamma, fr_t* t_out);
host::G1AffineH commit_checked(Ctx* ctx, const Srs* srs, const fr_t* coeffs_dev, size_t len);
void prove_3wire(Ctx* ctx, const Pk* pk, const Srs* srs, const plonk_fr* a_ev, const plonk_fr* b_ev, const plonk_fr* c_ev,
                 bool evals_on_device, const plonk_fr* pub, size_t n_pub, const plonk_fr* blinding, plonk_proof* out,
                 const ShardInfo* shard = nullptr);
host::fr_h poly_evaluate(Ctx* ctx, const fr_t* coeffs_dev, size_t len, const host::fr_h& z);
void poly_div_by_linear(Ctx* ctx, const fr_t* coeffs_dev, size_t len, const host::fr_h& z, fr_t* q_out_dev);
host::G1AffineH kzg_open(Ctx* ctx, const Srs* srs, const fr_t* coeffs_dev, size_t len, const host::fr_h& z, host::fr_h* y_out);
host::G1AffineH kzg_open_batch(Ctx* ctx, const Srs* srs, const fr_t* const* polys_dev, const size_t* lens, size_t count,
                               const host::fr_h& z, const host::fr_h& gamma, host::fr_h* evals_out);
host::G1AffineH g1_sum_affine(const plonk_g1* pts, size_t n);
void proof_challenges(const plonk_proof* pr, const plonk_fr* pub, size_t n_pub, plonk_fr out[6]);

}  // namespace plonk
