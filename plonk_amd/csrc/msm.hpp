// Host-side declarations of the MSM / SRS module (msm.hip).
#pragma once
#include <functional>

#include "comm.hpp"
#include "common.hpp"
#include "g1.cuh"
#include "g1x.cuh"

namespace plonk {

// Srs of backend_new/src/kzg.rs:26-35: g1_powers resident on one device.
struct Srs {
    int device = 0;
    size_t n = 0;
    DevBuf<g1_affine_t> points;   // saturated Montgomery (x, y): the ABI image
    DevBuf<g1x_affine_t> px;      // radix-2^29 form for the MSM kernels: `levels` rows of n points,
                                  // row w = 2^(table_c w) P_i (a single row when table_c == 0)
    uint32_t table_c = 0;         // window width the table was built for, 0 = no table
    uint32_t levels = 1;
};

// What an MSM reads its bases from.
struct MsmPoints {
    const g1x_affine_t* px;
    size_t level_stride;          // points per table row (= Srs::n)
    uint32_t table_c;             // 0: plain Pippenger, one bucket set per window
};
inline MsmPoints msm_points(const Srs* s) { return MsmPoints{s->px.p, s->n, s->table_c}; }

// SRS sizes from which the window table is built (below it the per-window bucket sets are cheap).
constexpr size_t MSM_TABLE_MIN_POINTS = (size_t)1 << 4;

uint32_t msm_window_bits(size_t n);
uint32_t msm_table_bits(size_t n);
// Work for the end of a batch: called once with the stream of the bucket kernels, after the last of
// them is enqueued, so what it launches there runs beside the final (latency-bound) reduction instead
// of after it.  msm_run / msm_run_batch return with ctx->stream ordered behind that work.
typedef std::function<void(hipStream_t)> MsmFiller;
// With an Exchange the MSM is one rank's share of a sharded one: the per-window sums of all ranks are
// all-gathered on the device and every rank combines them, so every rank returns the whole sum.  All
// ranks must then make the same sequence of MSM calls (an empty share included).
host::G1AffineH msm_run(Ctx* ctx, const MsmPoints& pts, const fr_t* scalars_dev, size_t n, const MsmFiller& filler = nullptr,
                        Exchange* ex = nullptr);
void msm_run_batch(Ctx* ctx, const MsmPoints& pts, const fr_t* const* scalars_dev, const size_t* n, int count,
                   host::G1AffineH* out, const MsmFiller& filler = nullptr, Exchange* ex = nullptr);
// The same in two steps, for a caller that prepares the scalar vectors one after the other.
constexpr int MSM_BATCH_MAX = 3;
struct MsmBatch {
    MsmPoints pts;
    int count = 0;
    Exchange* ex = nullptr;
    const fr_t* scalars[MSM_BATCH_MAX];
    size_t n[MSM_BATCH_MAX];
    bool held[MSM_BATCH_MAX] = {false, false, false};
};
// hold_buckets: only the sort is started; the bucket kernels (which fill the chip and leave whatever is enqueued
// beside them crawling) wait for msm_batch_release, or for msm_batch_finish.
void msm_batch_add(Ctx* ctx, MsmBatch& b, const fr_t* scalars_dev, size_t n, bool hold_buckets = false);
void msm_batch_release(Ctx* ctx, MsmBatch& b);
void msm_batch_finish(Ctx* ctx, MsmBatch& b, host::G1AffineH* out, const MsmFiller& filler = nullptr);
struct MsmTiming {
    float sort_ms, big_bucket_ms, bucket_ms, reduce_ms;
    uint64_t entries;
    uint32_t window_bits, bucket_sets;
};
void msm_drain(Ctx* ctx);
void msm_set_timing(Ctx* ctx, bool on);
void msm_last_timing(Ctx* ctx, MsmTiming* out);
void g1_convert_to_xform(Ctx* ctx, const g1_affine_t* in, size_t n, g1x_affine_t* out);
Srs* srs_upload(Ctx* ctx, const plonk_g1* pts, size_t n, bool with_table = true);
Srs* srs_from_tau(Ctx* ctx, const host::fr_h& tau, size_t first, size_t n, bool with_table = true);
void g1_unpack(const g1_affine_t& in, plonk_g1* out);
void g1_from_host(const host::G1AffineH& a, plonk_g1* out);

}  // namespace plonk
