// Shared host-side declarations for the library: context, error handling, device buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/plonk_hip.h"
#include "field.cuh"
#include "host_field.hpp"

namespace plonk {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define HIP_CHECK(expr)                                                                             \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            throw ::plonk::Error(PLONK_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

#define PLONK_REQUIRE(cond, msg)                                          \
    do {                                                                  \
        if (!(cond)) throw ::plonk::Error(PLONK_E_INVALID, (msg));        \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() {}
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        release();
        if (count) HIP_CHECK(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
    }
    void ensure(size_t count) { if (count > n) alloc(count); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
    }
};

struct NttPlan;

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;               // plonk_timer_start / _stop
    hipEvent_t q_ev0 = nullptr, q_ev1 = nullptr;           // around the last quotient kernel (while MSM timing is on)
    bool kernel_timing = false, q_timed = false;
    std::map<uint64_t, std::shared_ptr<NttPlan>> plans;   // key: log_n | inverse<<8 | coset<<9 | generator slot<<16
    DevBuf<fr_t> ntt_tmp;                                  // intermediate image of a multi-pass NTT
    bool ntt_lds_attr_set = false;                         // dynamic-LDS limit of ntt_pass_kernel raised
    std::shared_ptr<void> msm_ws;                          // MsmWorkspace (msm.hip)
    std::shared_ptr<void> prover_ws;                       // ProverWorkspace (prover.hip)
    ~Ctx();
};

inline fr_t to_dev(const host::fr_h& h) {
    fr_t d;
    memcpy(d.l, h.l, 32);
    return d;
}
inline host::fr_h to_host(const fr_t& d) {
    host::fr_h h;
    memcpy(h.l, d.l, 32);
    return h;
}

// ntt.hip
constexpr uint32_t NTT_INVERSE = PLONK_NTT_INVERSE;
constexpr uint32_t NTT_COSET = PLONK_NTT_COSET;
// in_len: every vector is zero from that index on (a zero-padded coefficient vector); the first sweep then
// neither reads nor scales the padding, which may hold anything.  0 = the whole vector is input.
void ntt_run(Ctx* ctx, fr_t* data_dev, uint32_t log_n, uint32_t flags, uint32_t batch, size_t batch_stride, size_t in_len = 0);
void ntt_run_on(Ctx* ctx, hipStream_t stream, fr_t* data_dev, uint32_t log_n, uint32_t flags, uint32_t batch,
                size_t batch_stride, size_t in_len = 0);
NttPlan* ntt_get_plan(Ctx* ctx, uint32_t log_n, uint32_t flags);
void ntt_run_shift(Ctx* ctx, fr_t* data_dev, uint32_t log_n, uint32_t flags, const host::fr_h& shift, uint32_t batch,
                   size_t batch_stride);
// ... with any coset generator, on any stream of the context (same ordering duty as ntt_run_on)
void ntt_run_shift_on(Ctx* ctx, hipStream_t stream, fr_t* data_dev, uint32_t log_n, uint32_t flags, const host::fr_h& shift,
                      uint32_t batch, size_t batch_stride);
void ntt_noalloc_init(int device);
uint32_t ntt_noalloc_max_log_n();
uint32_t ntt_sweeps(uint32_t log_n);
void ntt_run_noalloc(Ctx* ctx, fr_t* data_dev, uint32_t log_n, uint32_t flags, const host::fr_h& shift);

}  // namespace plonk
