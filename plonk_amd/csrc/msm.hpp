// Host-side declarations of the MSM / SRS module (msm.hip).
#pragma once
#include "common.hpp"
#include "g1.cuh"
#include "g1x.cuh"

namespace plonk {

// Srs of backend_new/src/kzg.rs:26-35: g1_powers resident on one device.
struct Srs {
    int device = 0;
    size_t n = 0;
    DevBuf<g1_affine_t> points;   // saturated Montgomery (x, y): the ABI image
    DevBuf<g1x_affine_t> px;      // the same points in the radix-2^29 form the MSM kernels use
};

uint32_t msm_window_bits(size_t n);
host::G1AffineH msm_run(Ctx* ctx, const g1x_affine_t* points_dev, const fr_t* scalars_dev, size_t n);
void msm_run_batch(Ctx* ctx, const g1x_affine_t* points_dev, const fr_t* const* scalars_dev, const size_t* n, int count,
                   host::G1AffineH* out);
void g1_convert_to_xform(Ctx* ctx, const g1_affine_t* in, size_t n, g1x_affine_t* out);
Srs* srs_upload(Ctx* ctx, const plonk_g1* pts, size_t n);
Srs* srs_from_tau(Ctx* ctx, const host::fr_h& tau, size_t first, size_t n);
void g1_unpack(const g1_affine_t& in, plonk_g1* out);
void g1_from_host(const host::G1AffineH& a, plonk_g1* out);

}  // namespace plonk
