// extern "C" surface declared in include/plonk_hip.h: context, device memory, NTT, encodings.
// Every entry point catches, records the message in the ctx and returns a status code.
#include <algorithm>

#include "common.hpp"
#include "msm.hpp"
#include "pairing.hpp"
#include "prover.hpp"

namespace plonk {
void synthetic_circuit(uint32_t log_n, plonk_fr* a, plonk_fr* b, plonk_fr* c, plonk_fr* const sel[5], plonk_fr* const sig[3],
                       plonk_fr blinding[11], plonk_fr* tau, plonk_fr* k1_out, plonk_fr* k2_out);
double bench_field_mul(Ctx* ctx, uint32_t which);
int verify_3wire(const plonk_vk* vk, const uint64_t* pi_positions, size_t n_positions, const plonk_g2* g2_tau,
                 const plonk_proof* pr, const plonk_fr* pub, size_t n_pub, bool* accepted);
std::string proof_to_json(const plonk_proof* p);
bool proof_from_json(const char* json, size_t len, plonk_proof* out);
void g2_out(const host::G2AffineH& a, plonk_g2* p);
}  // namespace plonk

using namespace plonk;

struct plonk_ctx {
    Ctx c;
};
struct plonk_srs {
    Srs s;
};
struct plonk_pk {
    Pk k;
};
struct plonk_comm {
    std::unique_ptr<Exchange> ex;
};

// A failed call may leave kernels of an abandoned MSM batch on the context's streams.
static void drain_after_error(Ctx* c) {
    msm_drain(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
}

// HIP's current device is per host thread: every entry point makes the context's device current
// before it allocates or launches anything.
#define API_BEGIN(ctxp) \
    Ctx* C_ = &(ctxp)->c; \
    try { \
        HIP_CHECK(hipSetDevice(C_->device));
#define API_END()                                   \
        return PLONK_OK;                            \
    } catch (const plonk::Error& e) {               \
        C_->last_error = e.what();                  \
        drain_after_error(C_);                      \
        return e.code;                              \
    } catch (const std::bad_alloc&) {               \
        C_->last_error = "out of host memory";      \
        drain_after_error(C_);                      \
        return PLONK_E_NOMEM;                       \
    } catch (const std::exception& e) {             \
        C_->last_error = e.what();                  \
        drain_after_error(C_);                      \
        return PLONK_E_INTERNAL;                    \
    }

extern "C" {

int32_t plonk_ctx_create(int32_t device, plonk_ctx** out) {
    if (!out) return PLONK_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PLONK_E_HIP;
    if (device < 0 || device >= count) return PLONK_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return PLONK_E_HIP;
    plonk_ctx* ctx = new (std::nothrow) plonk_ctx();
    if (!ctx) return PLONK_E_NOMEM;
    ctx->c.device = device;
    if (hipStreamCreateWithFlags(&ctx->c.stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return PLONK_E_HIP;
    }
    try {
        ntt_noalloc_init(device);   // address of this device's static workspace (plonk_ntt_dev_noalloc)
        ctx->c.ntt_lds_attr_set = true;   // ... which also raised the sweeps' LDS limit on this device
    } catch (const std::exception&) {
        delete ctx;
        return PLONK_E_HIP;
    }
    *out = ctx;
    return PLONK_OK;
}

void plonk_ctx_destroy(plonk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->c.device);
    (void)hipStreamSynchronize(ctx->c.stream);
    delete ctx;
}

const char* plonk_last_error(const plonk_ctx* ctx) { return ctx ? ctx->c.last_error.c_str() : "null ctx"; }

void* plonk_ctx_stream(plonk_ctx* ctx) { return ctx ? (void*)ctx->c.stream : nullptr; }

int32_t plonk_sync(plonk_ctx* ctx) {
    if (!ctx) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    HIP_CHECK(hipStreamSynchronize(C_->stream));
    API_END()
}

int32_t plonk_dev_alloc(plonk_ctx* ctx, size_t bytes, void** out_dev) {
    if (!ctx || !out_dev) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    *out_dev = nullptr;
    if (bytes) HIP_CHECK(hipMalloc(out_dev, bytes));
    API_END()
}

int32_t plonk_dev_free(plonk_ctx* ctx, void* dev) {
    if (!ctx) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    HIP_CHECK(hipStreamSynchronize(C_->stream));
    if (dev) HIP_CHECK(hipFree(dev));
    API_END()
}

int32_t plonk_dev_upload(plonk_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    if (!ctx || (bytes && (!dst_dev || !src_host))) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    if (bytes) {
        HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, C_->stream));
        HIP_CHECK(hipStreamSynchronize(C_->stream));
    }
    API_END()
}

int32_t plonk_dev_download(plonk_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    if (!ctx || (bytes && (!dst_host || !src_dev))) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    if (bytes) {
        HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, C_->stream));
        HIP_CHECK(hipStreamSynchronize(C_->stream));
    }
    API_END()
}

static void check_ntt_args(const void* data, uint32_t log_n, uint32_t flags, uint32_t batch, size_t stride) {
    PLONK_REQUIRE(data != nullptr, "ntt: null data");
    PLONK_REQUIRE(log_n <= 28, "ntt: size must be a power of two no larger than 2^28");
    PLONK_REQUIRE((flags & ~3u) == 0, "ntt: unknown flag");
    PLONK_REQUIRE(batch >= 1, "ntt: batch must be >= 1");
    PLONK_REQUIRE(batch == 1 || stride >= ((size_t)1 << log_n), "ntt: batch stride smaller than the transform");
}

static host::fr_h shift_or_default(const plonk_fr* shift) {
    if (!shift) return host::Fr::from_u64(7);
    host::fr_h s;
    memcpy(s.l, shift->l, 32);
    PLONK_REQUIRE(!host::Fr::geq_p(s) && !host::Fr::is_zero(s), "ntt: coset shift must be a non-zero reduced Fr");
    return s;
}

int32_t plonk_ntt_dev(plonk_ctx* ctx, plonk_fr* data_dev, uint32_t log_n, uint32_t flags, const plonk_fr* shift,
                      uint32_t batch, size_t stride) {
    if (!ctx) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    check_ntt_args(data_dev, log_n, flags, batch, stride);
    ntt_run_shift(C_, reinterpret_cast<fr_t*>(data_dev), log_n, flags, shift_or_default(shift), batch, stride);
    API_END()
}

int32_t plonk_ntt_dev_noalloc(plonk_ctx* ctx, plonk_fr* data_dev, uint32_t log_n, uint32_t flags, const plonk_fr* shift) {
    if (!ctx) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    check_ntt_args(data_dev, log_n, flags, 1, 0);
    ntt_run_noalloc(C_, reinterpret_cast<fr_t*>(data_dev), log_n, flags, shift_or_default(shift));
    API_END()
}

uint32_t plonk_ntt_noalloc_max_log_n(void) { return ntt_noalloc_max_log_n(); }

uint32_t plonk_ntt_sweeps(uint32_t log_n) {
    try {
        return log_n <= 28 ? ntt_sweeps(log_n) : 0;
    } catch (const std::exception&) {
        return 0;
    }
}

int32_t plonk_ntt(plonk_ctx* ctx, plonk_fr* data_host, uint32_t log_n, uint32_t flags, const plonk_fr* shift,
                  uint32_t batch, size_t stride) {
    if (!ctx) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    check_ntt_args(data_host, log_n, flags, batch, stride);
    const size_t n = (size_t)1 << log_n;
    const size_t total = (size_t)(batch - 1) * stride + n;
    DevBuf<fr_t> buf(total);
    HIP_CHECK(hipMemcpyAsync(buf.p, data_host, total * sizeof(fr_t), hipMemcpyHostToDevice, C_->stream));
    ntt_run_shift(C_, buf.p, log_n, flags, shift_or_default(shift), batch, stride);
    HIP_CHECK(hipMemcpyAsync(data_host, buf.p, total * sizeof(fr_t), hipMemcpyDeviceToHost, C_->stream));
    HIP_CHECK(hipStreamSynchronize(C_->stream));
    API_END()
}


// ---- SRS / commit / MSM ----
static host::fr_h fr_in(const plonk_fr* x, const char* what) {
    PLONK_REQUIRE(x != nullptr, std::string(what) + ": null");
    host::fr_h h;
    memcpy(h.l, x->l, 32);
    PLONK_REQUIRE(!host::Fr::geq_p(h), std::string(what) + ": not a reduced Fr");
    return h;
}

int32_t plonk_srs_upload(plonk_ctx* ctx, const plonk_g1* g1_powers, size_t n, plonk_srs** out) {
    if (!ctx || !out || (n && !g1_powers)) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    *out = nullptr;
    *out = reinterpret_cast<plonk_srs*>(srs_upload(C_, g1_powers, n));
    API_END()
}

int32_t plonk_srs_from_tau(plonk_ctx* ctx, const plonk_fr* tau, size_t n, plonk_srs** out) {
    if (!ctx || !out) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    *out = nullptr;
    *out = reinterpret_cast<plonk_srs*>(srs_from_tau(C_, fr_in(tau, "srs tau"), 0, n));
    API_END()
}

int32_t plonk_srs_from_tau_range(plonk_ctx* ctx, const plonk_fr* tau, size_t first, size_t count, plonk_srs** out) {
    if (!ctx || !out) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    *out = nullptr;
    *out = reinterpret_cast<plonk_srs*>(srs_from_tau(C_, fr_in(tau, "srs tau"), first, count));
    API_END()
}

size_t plonk_srs_len(const plonk_srs* srs) { return srs ? reinterpret_cast<const Srs*>(srs)->n : 0; }

int32_t plonk_srs_download(plonk_ctx* ctx, const plonk_srs* srs, size_t first, size_t count, plonk_g1* out) {
    if (!ctx || !srs || (count && !out)) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const Srs* s = reinterpret_cast<const Srs*>(srs);
    PLONK_REQUIRE(first <= s->n && count <= s->n - first, "srs: range out of bounds");
    std::vector<g1_affine_t> tmp(count);
    if (count) HIP_CHECK(hipMemcpy(tmp.data(), s->points.p + first, count * sizeof(g1_affine_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < count; ++i) g1_unpack(tmp[i], out + i);
    API_END()
}

void plonk_srs_free(plonk_srs* srs) {
    if (!srs) return;
    Srs* s = reinterpret_cast<Srs*>(srs);
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    delete s;
}

int32_t plonk_commit_dev(plonk_ctx* ctx, const plonk_srs* srs, const plonk_fr* coeffs_dev, size_t len, plonk_g1* out) {
    if (!ctx || !srs || !out || (len && !coeffs_dev)) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const Srs* s = reinterpret_cast<const Srs*>(srs);
    PLONK_REQUIRE(s->device == C_->device, "commit: SRS lives on another device");
    g1_from_host(commit_checked(C_, s, reinterpret_cast<const fr_t*>(coeffs_dev), len), out);
    API_END()
}

int32_t plonk_commit(plonk_ctx* ctx, const plonk_srs* srs, const plonk_fr* coeffs_host, size_t len, plonk_g1* out) {
    if (!ctx || !srs || !out || (len && !coeffs_host)) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const Srs* s = reinterpret_cast<const Srs*>(srs);
    PLONK_REQUIRE(s->device == C_->device, "commit: SRS lives on another device");
    // Polynomial::from_coeffs normalises (polynomial.rs:47-51): trailing zeros do not count
    // towards the degree that kzg.rs:232 checks.
    while (len && (coeffs_host[len - 1].l[0] | coeffs_host[len - 1].l[1] | coeffs_host[len - 1].l[2] |
                   coeffs_host[len - 1].l[3]) == 0)
        --len;
    PLONK_REQUIRE(len <= s->n, "commit: polynomial degree exceeds SRS max degree");
    DevBuf<fr_t> buf(len);
    if (len) HIP_CHECK(hipMemcpyAsync(buf.p, coeffs_host, len * sizeof(fr_t), hipMemcpyHostToDevice, C_->stream));
    g1_from_host(msm_run(C_, msm_points(s), buf.p, len), out);
    API_END()
}

int32_t plonk_msm(plonk_ctx* ctx, const plonk_g1* points_host, const plonk_fr* scalars_host, size_t n, plonk_g1* out) {
    if (!ctx || !out || (n && (!points_host || !scalars_host))) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    std::unique_ptr<Srs> tmp(srs_upload(C_, points_host, n, /*with_table=*/false));
    DevBuf<fr_t> sc(n);
    if (n) HIP_CHECK(hipMemcpyAsync(sc.p, scalars_host, n * sizeof(fr_t), hipMemcpyHostToDevice, C_->stream));
    g1_from_host(msm_run(C_, msm_points(tmp.get()), sc.p, n), out);
    API_END()
}

// ---- proving key ----
int32_t plonk_pk_create(plonk_ctx* ctx, uint32_t log_n, const plonk_fr* q_m, const plonk_fr* q_l, const plonk_fr* q_r,
                        const plonk_fr* q_o, const plonk_fr* q_c, const plonk_fr* s_sigma1, const plonk_fr* s_sigma2,
                        const plonk_fr* s_sigma3, const plonk_fr* k1, const plonk_fr* k2, const uint32_t* pi_positions,
                        size_t n_pi_positions, plonk_pk** out) {
    if (!ctx || !out) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    *out = nullptr;
    const plonk_fr* sel[5] = {q_m, q_l, q_r, q_o, q_c};
    const plonk_fr* sig[3] = {s_sigma1, s_sigma2, s_sigma3};
    for (auto p : sel) PLONK_REQUIRE(p != nullptr, "pk: null selector vector");
    for (auto p : sig) PLONK_REQUIRE(p != nullptr, "pk: null sigma vector");
    PLONK_REQUIRE(n_pi_positions == 0 || pi_positions != nullptr, "pk: null public input positions");
    *out = reinterpret_cast<plonk_pk*>(pk_create(C_, log_n, sel, sig, fr_in(k1, "pk k1"), fr_in(k2, "pk k2"), pi_positions,
                                                 n_pi_positions));
    API_END()
}

void plonk_pk_free(plonk_pk* pk) {
    if (!pk) return;
    Pk* k = reinterpret_cast<Pk*>(pk);
    (void)hipSetDevice(k->device);
    (void)hipDeviceSynchronize();
    delete k;
}

int32_t plonk_pk_commitments(plonk_ctx* ctx, const plonk_pk* pk, const plonk_srs* srs, plonk_g1 out[8]) {
    if (!ctx || !pk || !srs || !out) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const Pk* k = reinterpret_cast<const Pk*>(pk);
    const Srs* s = reinterpret_cast<const Srs*>(srs);
    PLONK_REQUIRE(k->device == C_->device && s->device == C_->device, "pk commitments: key or SRS lives on another device");
    for (int q = 0; q < 8; ++q) g1_from_host(commit_checked(C_, s, k->coef.p + q * k->n, k->n), out + q);
    API_END()
}

int32_t plonk_pk_poly(plonk_ctx* ctx, const plonk_pk* pk, uint32_t which, plonk_fr* out_host, size_t* len) {
    if (!ctx || !pk || !out_host || !len || which > 8) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const Pk* k = reinterpret_cast<const Pk*>(pk);
    PLONK_REQUIRE(k->device == C_->device, "pk poly: proving key lives on another device");
    HIP_CHECK(hipMemcpy(out_host, k->coef.p + which * k->n, k->n * sizeof(fr_t), hipMemcpyDeviceToHost));
    size_t l = k->n;
    while (l && (out_host[l - 1].l[0] | out_host[l - 1].l[1] | out_host[l - 1].l[2] | out_host[l - 1].l[3]) == 0) --l;
    *len = l;
    API_END()
}

int32_t plonk_quotient_3wire_dev(plonk_ctx* ctx, const plonk_pk* pk, plonk_fr* a_dev, plonk_fr* b_dev, plonk_fr* c_dev,
                                 plonk_fr* z_dev, plonk_fr* pi_dev, const plonk_fr* alpha, const plonk_fr* beta,
                                 const plonk_fr* gamma, plonk_fr* t_out_dev) {
    if (!ctx || !pk || !a_dev || !b_dev || !c_dev || !z_dev || !t_out_dev) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const Pk* k = reinterpret_cast<const Pk*>(pk);
    PLONK_REQUIRE(k->device == C_->device, "quotient: proving key lives on another device");
    quotient_3wire(C_, k, reinterpret_cast<fr_t*>(a_dev), reinterpret_cast<fr_t*>(b_dev), reinterpret_cast<fr_t*>(c_dev),
                   reinterpret_cast<fr_t*>(z_dev), reinterpret_cast<fr_t*>(pi_dev), fr_in(alpha, "alpha"),
                   fr_in(beta, "beta"), fr_in(gamma, "gamma"), reinterpret_cast<fr_t*>(t_out_dev));
    API_END()
}

int32_t plonk_prove_3wire(plonk_ctx* ctx, const plonk_pk* pk, const plonk_srs* srs, const plonk_fr* a_evals,
                          const plonk_fr* b_evals, const plonk_fr* c_evals, const plonk_fr* public_inputs,
                          size_t n_public_inputs, const plonk_fr blinding[11], plonk_proof* out) {
    if (!ctx || !pk || !srs || !a_evals || !b_evals || !c_evals || !blinding || !out) return PLONK_E_INVALID;
    if (n_public_inputs && !public_inputs) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const Pk* k = reinterpret_cast<const Pk*>(pk);
    const Srs* s = reinterpret_cast<const Srs*>(srs);
    PLONK_REQUIRE(k->device == C_->device && s->device == C_->device, "prove: key or SRS lives on another device");
    prove_3wire(C_, k, s, a_evals, b_evals, c_evals, false, public_inputs, n_public_inputs, blinding, out);
    API_END()
}

int32_t plonk_prove_3wire_dev(plonk_ctx* ctx, const plonk_pk* pk, const plonk_srs* srs, const plonk_fr* a_evals_dev,
                              const plonk_fr* b_evals_dev, const plonk_fr* c_evals_dev, const plonk_fr* public_inputs,
                              size_t n_public_inputs, const plonk_fr blinding[11], plonk_proof* out) {
    if (!ctx || !pk || !srs || !a_evals_dev || !b_evals_dev || !c_evals_dev || !blinding || !out) return PLONK_E_INVALID;
    if (n_public_inputs && !public_inputs) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const Pk* k = reinterpret_cast<const Pk*>(pk);
    const Srs* s = reinterpret_cast<const Srs*>(srs);
    PLONK_REQUIRE(k->device == C_->device && s->device == C_->device, "prove: key or SRS lives on another device");
    prove_3wire(C_, k, s, a_evals_dev, b_evals_dev, c_evals_dev, true, public_inputs, n_public_inputs, blinding, out);
    API_END()
}

int32_t plonk_comm_unique_id(uint8_t out[PLONK_COMM_ID_BYTES]) {
    if (!out) return PLONK_E_INVALID;
    try {
        comm_unique_id(out);
    } catch (const plonk::Error& e) {
        return e.code;
    } catch (const std::exception&) {
        return PLONK_E_INTERNAL;
    }
    return PLONK_OK;
}

int32_t plonk_comm_create_rccl(plonk_ctx* ctx, const uint8_t id[PLONK_COMM_ID_BYTES], uint32_t rank, uint32_t world,
                               plonk_comm** out) {
    if (!ctx || !id || !out) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    *out = nullptr;
    std::unique_ptr<plonk_comm> c(new plonk_comm());
    c->ex.reset(comm_create_rccl(C_->device, id, rank, world));
    *out = c.release();
    API_END()
}

int32_t plonk_comm_create_callback(plonk_ctx* ctx, uint32_t rank, uint32_t world, plonk_allgather_fn allgather, void* user,
                                   plonk_comm** out) {
    if (!ctx || !out) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    *out = nullptr;
    std::unique_ptr<plonk_comm> c(new plonk_comm());
    c->ex.reset(comm_create_callback(C_->device, rank, world, allgather, user));
    *out = c.release();
    API_END()
}

int32_t plonk_comm_create_loopback(plonk_ctx* ctx, uint32_t rank, uint32_t world, size_t total_points, plonk_comm** out) {
    if (!ctx || !out) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    *out = nullptr;
    std::unique_ptr<plonk_comm> c(new plonk_comm());
    c->ex.reset(comm_create_loopback(C_->device, rank, world, total_points));
    *out = c.release();
    API_END()
}

void plonk_comm_destroy(plonk_comm* comm) { delete comm; }

int32_t plonk_comm_stats(const plonk_comm* comm, uint64_t* calls, uint64_t* bytes) {
    if (!comm || !calls || !bytes) return PLONK_E_INVALID;
    *calls = comm->ex->calls;
    *bytes = comm->ex->bytes;
    return PLONK_OK;
}

int32_t plonk_comm_timing(plonk_comm* comm, int32_t enable, double* elapsed_ms) {
    if (!comm) return PLONK_E_INVALID;
    try {
        (void)hipSetDevice(comm->ex->device);
        const double ms = comm->ex->take_elapsed_ms();
        if (elapsed_ms) *elapsed_ms = ms;
        comm->ex->set_timing(enable != 0);
        return PLONK_OK;
    } catch (const std::exception&) {
        return PLONK_E_HIP;
    }
}

int32_t plonk_prove_3wire_sharded(plonk_ctx* ctx, const plonk_pk* pk, const plonk_srs* srs_slice, size_t shard_lo,
                                  plonk_comm* comm, const plonk_fr* a_evals_dev, const plonk_fr* b_evals_dev,
                                  const plonk_fr* c_evals_dev, const plonk_fr* public_inputs, size_t n_public_inputs,
                                  const plonk_fr blinding[11], plonk_proof* out) {
    if (!ctx || !pk || !srs_slice || !comm || !a_evals_dev || !b_evals_dev || !c_evals_dev || !blinding || !out)
        return PLONK_E_INVALID;
    if (n_public_inputs && !public_inputs) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const Pk* k = reinterpret_cast<const Pk*>(pk);
    const Srs* s = reinterpret_cast<const Srs*>(srs_slice);
    PLONK_REQUIRE(k->device == C_->device && s->device == C_->device, "prove: key or SRS lives on another device");
    PLONK_REQUIRE(comm->ex->world > 1 || shard_lo == 0, "prove: a single rank holds the whole point range");
    ShardInfo sh;
    sh.lo = shard_lo;
    sh.ex = comm->ex.get();
    prove_3wire(C_, k, s, a_evals_dev, b_evals_dev, c_evals_dev, true, public_inputs, n_public_inputs, blinding, out, &sh);
    API_END()
}

int32_t plonk_prove_3wire_grouped(plonk_ctx* ctx, const plonk_pk* pk, const plonk_shard_slices* slices, plonk_comm* comm,
                                  const plonk_fr* a_evals_dev, const plonk_fr* b_evals_dev, const plonk_fr* c_evals_dev,
                                  const plonk_fr* public_inputs, size_t n_public_inputs, const plonk_fr blinding[11],
                                  plonk_proof* out) {
    if (!ctx || !pk || !slices || !slices->srs[0] || !comm || !a_evals_dev || !b_evals_dev || !c_evals_dev || !blinding || !out)
        return PLONK_E_INVALID;
    if (n_public_inputs && !public_inputs) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const Pk* k = reinterpret_cast<const Pk*>(pk);
    const Srs* s = reinterpret_cast<const Srs*>(slices->srs[0]);
    PLONK_REQUIRE(k->device == C_->device && s->device == C_->device, "prove: key or SRS lives on another device");
    PLONK_REQUIRE(comm->ex->world > 1 || slices->lo[0] == 0, "prove: a single rank holds the whole point range");
    ShardInfo sh;
    sh.lo = slices->lo[0];
    sh.ex = comm->ex.get();
    for (int g = 0; g < 2; ++g) {
        if (!slices->srs[g + 1]) continue;
        sh.group_srs[g] = reinterpret_cast<const Srs*>(slices->srs[g + 1]);
        sh.group_lo[g] = slices->lo[g + 1];
        PLONK_REQUIRE(sh.group_srs[g]->device == C_->device, "prove: a group slice of the SRS lives on another device");
    }
    prove_3wire(C_, k, s, a_evals_dev, b_evals_dev, c_evals_dev, true, public_inputs, n_public_inputs, blinding, out, &sh);
    API_END()
}

// ---- stand-alone polynomial / opening seam ----
int32_t plonk_poly_eval_dev(plonk_ctx* ctx, const plonk_fr* coeffs_dev, size_t len, const plonk_fr* z, plonk_fr* y_out) {
    if (!ctx || !y_out || (len && !coeffs_dev)) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const host::fr_h y = poly_evaluate(C_, reinterpret_cast<const fr_t*>(coeffs_dev), len, fr_in(z, "evaluate: point"));
    memcpy(y_out->l, y.l, 32);
    API_END()
}

int32_t plonk_div_by_linear_dev(plonk_ctx* ctx, const plonk_fr* coeffs_dev, size_t len, const plonk_fr* z, plonk_fr* q_out_dev) {
    if (!ctx || (len && !coeffs_dev) || (len > 1 && !q_out_dev)) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    poly_div_by_linear(C_, reinterpret_cast<const fr_t*>(coeffs_dev), len, fr_in(z, "div_by_linear: point"),
                       reinterpret_cast<fr_t*>(q_out_dev));
    API_END()
}

int32_t plonk_open_dev(plonk_ctx* ctx, const plonk_srs* srs, const plonk_fr* coeffs_dev, size_t len, const plonk_fr* z,
                       plonk_fr* y_out, plonk_g1* proof_out) {
    if (!ctx || !srs || !y_out || !proof_out || (len && !coeffs_dev)) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const Srs* s = reinterpret_cast<const Srs*>(srs);
    PLONK_REQUIRE(s->device == C_->device, "open: SRS lives on another device");
    host::fr_h y;
    g1_from_host(kzg_open(C_, s, reinterpret_cast<const fr_t*>(coeffs_dev), len, fr_in(z, "open: point"), &y), proof_out);
    memcpy(y_out->l, y.l, 32);
    API_END()
}

int32_t plonk_open_batch_dev(plonk_ctx* ctx, const plonk_srs* srs, const plonk_fr* const* polys_dev, const size_t* lens,
                             size_t count, const plonk_fr* z, const plonk_fr* gamma, plonk_fr* evals_out, plonk_g1* proof_out) {
    if (!ctx || !srs || !proof_out || (count && (!polys_dev || !lens || !evals_out))) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const Srs* s = reinterpret_cast<const Srs*>(srs);
    PLONK_REQUIRE(s->device == C_->device, "open batch: SRS lives on another device");
    std::vector<const fr_t*> ps(count);
    for (size_t k = 0; k < count; ++k) ps[k] = reinterpret_cast<const fr_t*>(polys_dev[k]);
    std::vector<host::fr_h> ys(count);
    g1_from_host(kzg_open_batch(C_, s, ps.data(), lens, count, fr_in(z, "open batch: point"), fr_in(gamma, "open batch: gamma"),
                                ys.data()), proof_out);
    for (size_t k = 0; k < count; ++k) memcpy(evals_out[k].l, ys[k].l, 32);
    API_END()
}

int32_t plonk_g1_sum(const plonk_g1* points, size_t n, plonk_g1* out) {
    if (!out || (n && !points)) return PLONK_E_INVALID;
    g1_from_host(g1_sum_affine(points, n), out);
    return PLONK_OK;
}

uint32_t plonk_msm_window_bits(size_t n) { return msm_window_bits(n); }

int32_t plonk_msm_set_timing(plonk_ctx* ctx, int32_t enable) {
    if (!ctx) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    msm_set_timing(C_, enable != 0);
    C_->kernel_timing = enable != 0;
    API_END()
}

int32_t plonk_quotient_last_timing(plonk_ctx* ctx, double* elapsed_ms) {
    if (!ctx || !elapsed_ms) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    PLONK_REQUIRE(C_->q_timed && C_->q_ev1, "quotient timing: no timed quotient evaluation has run on this context");
    HIP_CHECK(hipEventSynchronize(C_->q_ev1));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, C_->q_ev0, C_->q_ev1));
    *elapsed_ms = ms;
    API_END()
}

int32_t plonk_msm_last_timing(plonk_ctx* ctx, plonk_msm_timing* out) {
    if (!ctx || !out) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    MsmTiming t;
    msm_last_timing(C_, &t);
    out->sort_ms = t.sort_ms; out->big_bucket_ms = t.big_bucket_ms; out->bucket_ms = t.bucket_ms; out->reduce_ms = t.reduce_ms;
    out->entries = t.entries; out->window_bits = t.window_bits; out->bucket_sets = t.bucket_sets;
    API_END()
}

int32_t plonk_timer_start(plonk_ctx* ctx) {
    if (!ctx) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    if (!C_->ev0) {
        HIP_CHECK(hipEventCreate(&C_->ev0));
        HIP_CHECK(hipEventCreate(&C_->ev1));
    }
    HIP_CHECK(hipEventRecord(C_->ev0, C_->stream));
    API_END()
}

int32_t plonk_timer_stop(plonk_ctx* ctx, double* elapsed_ms) {
    if (!ctx || !elapsed_ms) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    PLONK_REQUIRE(C_->ev0 != nullptr, "timer: stop without start");
    HIP_CHECK(hipEventRecord(C_->ev1, C_->stream));
    HIP_CHECK(hipEventSynchronize(C_->ev1));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, C_->ev0, C_->ev1));
    *elapsed_ms = ms;
    API_END()
}

int32_t plonk_proof_serialize(const plonk_proof* proof, uint8_t out[PLONK_PROOF_WIRE_BYTES]) {
    if (!proof || !out) return PLONK_E_INVALID;
    const plonk_g1* g[9] = {&proof->a_comm, &proof->b_comm, &proof->c_comm, &proof->z_comm, &proof->t_lo_comm,
                            &proof->t_mid_comm, &proof->t_hi_comm, &proof->w_zeta_comm, &proof->w_zeta_omega_comm};
    const plonk_fr* f[6] = {&proof->a_eval, &proof->b_eval, &proof->c_eval, &proof->s_sigma1_eval,
                            &proof->s_sigma2_eval, &proof->z_omega_eval};
    uint8_t* p = out;
    for (int k = 0; k < 7; ++k, p += 48) plonk_g1_compress(g[k], p);
    for (int k = 0; k < 6; ++k, p += 32) plonk_fr_to_bytes(f[k], p);
    for (int k = 7; k < 9; ++k, p += 48) plonk_g1_compress(g[k], p);
    return PLONK_OK;
}

int32_t plonk_proof_challenges(const plonk_proof* proof, const plonk_fr* public_inputs, size_t n_public_inputs,
                               plonk_fr out[6]) {
    if (!proof || !out || (n_public_inputs && !public_inputs)) return PLONK_E_INVALID;
    proof_challenges(proof, public_inputs, n_public_inputs, out);
    return PLONK_OK;
}

int32_t plonk_fr_to_bytes(const plonk_fr* a, uint8_t out[32]) {
    if (!a || !out) return PLONK_E_INVALID;
    host::fr_h h;
    memcpy(h.l, a->l, 32);
    host::fr_to_bytes_le(h, out);
    return PLONK_OK;
}

int32_t plonk_fr_from_bytes(const uint8_t in[32], plonk_fr* out) {
    if (!in || !out) return PLONK_E_INVALID;
    host::fr_h h;
    if (!host::fr_from_bytes_le(in, &h)) return PLONK_E_INVALID;
    memcpy(out->l, h.l, 32);
    return PLONK_OK;
}

int32_t plonk_g1_compress(const plonk_g1* p, uint8_t out[48]) {
    if (!p || !out) return PLONK_E_INVALID;
    host::G1AffineH a;
    memcpy(a.x.l, p->x, 48);
    memcpy(a.y.l, p->y, 48);
    a.inf = p->infinity != 0;
    host::g1_compress(a, out);
    return PLONK_OK;
}

int32_t plonk_g1_decompress(const uint8_t in[48], plonk_g1* out) {
    if (!in || !out) return PLONK_E_INVALID;
    host::G1AffineH a;
    if (!host::g1_decompress(in, &a)) return PLONK_E_INVALID;
    g1_from_host(a, out);
    return PLONK_OK;
}

int32_t plonk_proof_deserialize(const uint8_t in[PLONK_PROOF_WIRE_BYTES], plonk_proof* proof) {
    if (!in || !proof) return PLONK_E_INVALID;
    plonk_proof pr;
    memset(&pr, 0, sizeof(pr));
    plonk_g1* g[9] = {&pr.a_comm, &pr.b_comm, &pr.c_comm, &pr.z_comm, &pr.t_lo_comm,
                      &pr.t_mid_comm, &pr.t_hi_comm, &pr.w_zeta_comm, &pr.w_zeta_omega_comm};
    plonk_fr* f[6] = {&pr.a_eval, &pr.b_eval, &pr.c_eval, &pr.s_sigma1_eval, &pr.s_sigma2_eval, &pr.z_omega_eval};
    const uint8_t* p = in;
    for (int k = 0; k < 7; ++k, p += 48)
        if (plonk_g1_decompress(p, g[k]) != PLONK_OK) return PLONK_E_INVALID;
    for (int k = 0; k < 6; ++k, p += 32)
        if (plonk_fr_from_bytes(p, f[k]) != PLONK_OK) return PLONK_E_INVALID;
    for (int k = 7; k < 9; ++k, p += 48)
        if (plonk_g1_decompress(p, g[k]) != PLONK_OK) return PLONK_E_INVALID;
    *proof = pr;
    return PLONK_OK;
}

// ---- verifier / JSON (host only) ----
static host::G2AffineH g2_from(const plonk_g2& p) {
    host::G2AffineH a;
    memcpy(a.x.c0.l, p.x_c0, 48);
    memcpy(a.x.c1.l, p.x_c1, 48);
    memcpy(a.y.c0.l, p.y_c0, 48);
    memcpy(a.y.c1.l, p.y_c1, 48);
    a.inf = p.infinity != 0;
    return a;
}
// What ark's G2Affine can hold after deserialisation: reduced coordinates, on the curve, and in the order-r subgroup
// (the cofactor of G2 is large, so an arbitrary curve point is almost never in it): [r] Q = O.
static bool g2_ok(const host::G2AffineH& a) {
    if (a.inf) return true;
    if (host::Fq::geq_p(a.x.c0) || host::Fq::geq_p(a.x.c1) || host::Fq::geq_p(a.y.c0) || host::Fq::geq_p(a.y.c1) ||
        !host::g2_on_curve(a))
        return false;
    return host::g2_in_subgroup(a);
}
static bool g1_ok(const plonk_g1& p, host::G1AffineH* out) {
    memcpy(out->x.l, p.x, 48);
    memcpy(out->y.l, p.y, 48);
    out->inf = p.infinity != 0;
    return out->inf || (!host::Fq::geq_p(out->x) && !host::Fq::geq_p(out->y) && host::g1_on_curve(*out));
}

int32_t plonk_g2_generator(plonk_g2* out) {
    if (!out) return PLONK_E_INVALID;
    g2_out(host::g2_generator(), out);
    return PLONK_OK;
}

int32_t plonk_g2_mul(const plonk_g2* p, const plonk_fr* k, plonk_g2* out) {
    if (!p || !k || !out) return PLONK_E_INVALID;
    const host::G2AffineH a = g2_from(*p);
    host::fr_h s;
    memcpy(s.l, k->l, 32);
    if (!g2_ok(a) || host::Fr::geq_p(s)) return PLONK_E_INVALID;
    const host::fr_h c = host::Fr::from_mont(s);
    g2_out(host::g2_mul_limbs(a, c.l, 4), out);
    return PLONK_OK;
}

int32_t plonk_pairing_check(const plonk_g1* p1, const plonk_g2* q1, const plonk_g1* p2, const plonk_g2* q2, int32_t* equal) {
    if (!p1 || !q1 || !p2 || !q2 || !equal) return PLONK_E_INVALID;
    host::G1AffineH a, b;
    const host::G2AffineH c = g2_from(*q1), d = g2_from(*q2);
    if (!g1_ok(*p1, &a) || !g1_ok(*p2, &b) || !g2_ok(c) || !g2_ok(d)) return PLONK_E_INVALID;
    *equal = host::pairing_check(a, c, b, d) ? 1 : 0;
    return PLONK_OK;
}

int32_t plonk_open_verify(const plonk_g1* commitment, const plonk_fr* z, const plonk_fr* y, const plonk_g1* proof,
                          const plonk_g2* g2_tau, int32_t* accepted) {
    if (!commitment || !z || !y || !proof || !g2_tau || !accepted) return PLONK_E_INVALID;
    *accepted = 0;
    host::G1AffineH c, w;
    const host::G2AffineH t2 = g2_from(*g2_tau);
    host::fr_h zz, yy;
    memcpy(zz.l, z->l, 32);
    memcpy(yy.l, y->l, 32);
    if (!g1_ok(*commitment, &c) || !g1_ok(*proof, &w) || !g2_ok(t2) || host::Fr::geq_p(zz) || host::Fr::geq_p(yy))
        return PLONK_E_INVALID;
    // [f]_1 - y [1]_1 and [tau]_2 - z [1]_2 (kzg.rs:289-294)
    const host::fr_h yc = host::Fr::from_mont(yy), zc = host::Fr::from_mont(zz);
    const host::G1XyzzH yg = host::g1_mul_limbs(host::g1_from_affine(host::g1_generator()), yc.l, 4);
    const host::G1AffineH lhs = host::g1_to_affine(host::g1_add(host::g1_from_affine(c), host::g1_neg(yg)));
    const host::G2AffineH zg = host::g2_mul_limbs(host::g2_generator(), zc.l, 4);
    const host::G2AffineH rhs = host::g2_add(t2, host::g2_neg(zg));
    *accepted = host::pairing_check(lhs, host::g2_generator(), w, rhs) ? 1 : 0;   // kzg.rs:297-302
    return PLONK_OK;
}

int32_t plonk_vk_from_pk(plonk_ctx* ctx, const plonk_pk* pk, const plonk_srs* srs, plonk_vk* out) {
    if (!ctx || !pk || !srs || !out) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    const Pk* k = reinterpret_cast<const Pk*>(pk);
    const Srs* s = reinterpret_cast<const Srs*>(srs);
    PLONK_REQUIRE(k->device == C_->device && s->device == C_->device, "vk: key or SRS lives on another device");
    memset(out, 0, sizeof(*out));
    out->n = k->n;
    plonk_g1* dst[8] = {&out->q_m_comm, &out->q_l_comm, &out->q_r_comm, &out->q_o_comm, &out->q_c_comm,
                        &out->s_sigma1_comm, &out->s_sigma2_comm, &out->s_sigma3_comm};
    for (int q = 0; q < 8; ++q) g1_from_host(commit_checked(C_, s, k->coef.p + q * k->n, k->n), dst[q]);
    memcpy(out->omega.l, k->omega.l, 32);
    memcpy(out->k1.l, k->k1.l, 32);
    memcpy(out->k2.l, k->k2.l, 32);
    API_END()
}

int32_t plonk_verify_3wire(const plonk_vk* vk, const uint64_t* public_input_positions, size_t n_positions,
                           const plonk_g2* g2_tau, const plonk_proof* proof, const plonk_fr* public_inputs,
                           size_t n_public_inputs, int32_t* accepted) {
    if (!vk || !g2_tau || !proof || !accepted || (n_positions && !public_input_positions) ||
        (n_public_inputs && !public_inputs))
        return PLONK_E_INVALID;
    *accepted = 0;
    try {
        bool ok = false;
        const int rc = verify_3wire(vk, public_input_positions, n_positions, g2_tau, proof, public_inputs, n_public_inputs, &ok);
        if (rc != PLONK_OK) return rc;
        *accepted = ok ? 1 : 0;
    } catch (const std::bad_alloc&) {
        return PLONK_E_NOMEM;
    } catch (const std::exception&) {
        return PLONK_E_INTERNAL;
    }
    return PLONK_OK;
}

int32_t plonk_proof_to_json(const plonk_proof* proof, char* out, size_t cap, size_t* len) {
    if (!proof || !len || (cap && !out)) return PLONK_E_INVALID;
    try {
        const std::string s = proof_to_json(proof);
        *len = s.size();
        if (s.size() > cap) return PLONK_E_INVALID;
        memcpy(out, s.data(), s.size());
    } catch (const std::exception&) {
        return PLONK_E_NOMEM;
    }
    return PLONK_OK;
}

int32_t plonk_proof_from_json(const char* json, size_t len, plonk_proof* proof) {
    if (!json || !proof) return PLONK_E_INVALID;
    try {
        return proof_from_json(json, len, proof) ? PLONK_OK : PLONK_E_INVALID;
    } catch (const std::exception&) {
        return PLONK_E_NOMEM;
    }
}

int32_t plonk_synthetic_circuit(uint32_t log_n, plonk_fr* a, plonk_fr* b, plonk_fr* c, plonk_fr* q_m, plonk_fr* q_l, plonk_fr* q_r,
                                plonk_fr* q_o, plonk_fr* q_c, plonk_fr* s_sigma1, plonk_fr* s_sigma2, plonk_fr* s_sigma3,
                                plonk_fr blinding[11], plonk_fr* tau, plonk_fr* k1, plonk_fr* k2) {
    plonk_fr* const sel[5] = {q_m, q_l, q_r, q_o, q_c};
    plonk_fr* const sig[3] = {s_sigma1, s_sigma2, s_sigma3};
    if (!a || !b || !c || !blinding || !tau || !k1 || !k2) return PLONK_E_INVALID;
    for (auto p : sel)
        if (!p) return PLONK_E_INVALID;
    for (auto p : sig)
        if (!p) return PLONK_E_INVALID;
    try {
        synthetic_circuit(log_n, a, b, c, sel, sig, blinding, tau, k1, k2);
    } catch (const plonk::Error& e) {
        return e.code;
    } catch (const std::exception&) {
        return PLONK_E_INTERNAL;
    }
    return PLONK_OK;
}

int32_t plonk_bench_field_mul(plonk_ctx* ctx, uint32_t which, double* muls_per_second) {
    if (!ctx || !muls_per_second || which > 11) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    *muls_per_second = bench_field_mul(C_, which);
    API_END()
}

}  // extern "C"
