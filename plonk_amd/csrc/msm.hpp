// Host-side declarations of the MSM / SRS module (msm.hip).
#pragma once
#include "common.hpp"
#include "g1.cuh"

namespace plonk {

// Srs of backend_new/src/kzg.rs:26-35: g1_powers resident on one device.
struct Srs {
    int device = 0;
    size_t n = 0;
    DevBuf<g1_affine_t> points;
};

uint32_t msm_window_bits(size_t n);
host::G1AffineH msm_run(Ctx* ctx, const g1_affine_t* points_dev, const fr_t* scalars_dev, size_t n);
Srs* srs_upload(Ctx* ctx, const plonk_g1* pts, size_t n);
Srs* srs_from_tau(Ctx* ctx, const host::fr_h& tau, size_t first, size_t n);
void g1_unpack(const g1_affine_t& in, plonk_g1* out);
void g1_from_host(const host::G1AffineH& a, plonk_g1* out);

}  // namespace plonk
