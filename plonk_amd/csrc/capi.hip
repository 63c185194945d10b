// extern "C" surface declared in include/plonk_hip.h: context, device memory, NTT, encodings.
// Every entry point catches, records the message in the ctx and returns a status code.
#include <algorithm>

#include "common.hpp"

namespace plonk {
double bench_field_mul(Ctx* ctx, uint32_t which);
void ntt_run_shift(Ctx* ctx, fr_t* data_dev, uint32_t log_n, uint32_t flags, const host::fr_h& shift, uint32_t batch,
                   size_t batch_stride);
}  // namespace plonk

using namespace plonk;

struct plonk_ctx {
    Ctx c;
};

#define API_BEGIN(ctxp) \
    Ctx* C_ = &(ctxp)->c; \
    try {
#define API_END()                                   \
        return PLONK_OK;                            \
    } catch (const plonk::Error& e) {               \
        C_->last_error = e.what();                  \
        return e.code;                              \
    } catch (const std::bad_alloc&) {               \
        C_->last_error = "out of host memory";      \
        return PLONK_E_NOMEM;                       \
    } catch (const std::exception& e) {             \
        C_->last_error = e.what();                  \
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
    PLONK_REQUIRE(log_n <= 32, "ntt: size must be a power of two no larger than 2^32 (field two-adicity)");
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

int32_t plonk_bench_field_mul(plonk_ctx* ctx, uint32_t which, double* muls_per_second) {
    if (!ctx || !muls_per_second || which > 1) return PLONK_E_INVALID;
    API_BEGIN(ctx)
    *muls_per_second = bench_field_mul(C_, which);
    API_END()
}

}  // extern "C"
