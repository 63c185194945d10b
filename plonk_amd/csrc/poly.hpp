// Host-side declarations of the polynomial helper kernels (poly.hip).
#pragma once
#include "common.hpp"

namespace plonk {

struct BlindArgs {
    fr_t bl[3];
    uint32_t k;
};

struct PermArgs {
    const fr_t *a, *b, *c, *s1, *s2, *s3, *w_lo, *w_hi;
    fr_t *num, *den;
    fr_t beta, gamma, k1, k2;
    uint32_t n, log_l;
};

static constexpr uint32_t EVAL_MAX = 8;
struct EvalArgs {
    const fr_t* coeffs[EVAL_MAX];
    uint32_t len[EVAL_MAX];
    fr_t x[EVAL_MAX];
};

static constexpr uint32_t LINCOMB_MAX = 16;
struct LincombArgs {
    const fr_t* p[LINCOMB_MAX];
    uint32_t len[LINCOMB_MAX];
    fr_t s[LINCOMB_MAX];
    uint32_t count;
};

void poly_blind(Ctx* ctx, fr_t* coeffs, uint32_t n, const host::fr_h* bl, uint32_t k);
void poly_poke(Ctx* ctx, fr_t* coeffs, uint32_t idx, const host::fr_h& v, bool subtract);
void poly_copy(Ctx* ctx, fr_t* dst, const fr_t* src, size_t n);   // same-device copy by a kernel
void perm_terms(Ctx* ctx, const PermArgs& p);
void perm_ratio(Ctx* ctx, const fr_t* num, fr_t* den, uint32_t n, fr_t* scratch, uint32_t* flag);
void scan_prefix_product(Ctx* ctx, const fr_t* x, fr_t* out, uint32_t n, DevBuf<fr_t>& totals);
void poly_eval_batch(Ctx* ctx, const EvalArgs& a, uint32_t count, fr_t* out_dev, DevBuf<fr_t>& partial);
void poly_lincomb(Ctx* ctx, const LincombArgs& a, fr_t* out, uint32_t out_len);
void poly_div_linear(Ctx* ctx, const fr_t* c, uint32_t len, const host::fr_h& z, fr_t* tmp, fr_t* q_out,
                     DevBuf<fr_t>& totals);
void poly_nonzero_check(Ctx* ctx, const fr_t* x, uint32_t lo, uint32_t hi, uint32_t* flag);

}  // namespace plonk
