// Multi-scalar multiplication over BLS12-381 G1 for gfx950: the device side of
// G1Projective::msm (backend_new/src/curve.rs:169-173) and Commitment::commit (kzg.rs:231-245).
//
// Pippenger bucket method with signed c-bit digits in [-2^(c-1), 2^(c-1)], W = ceil(256/c) windows:
//   1. sort     : the (point, sign) entries of all windows are grouped by bucket with a two-level
//                 counting sort in LDS (coarse bins per slab of scalars, then each bin by its low
//                 bits); digits are recomputed from the scalars, never stored.
//   2. order    : buckets are listed by decreasing size, so the lanes of a wave run equally long.
//   3. buckets  : one lane per bucket sums its points with XYZZ mixed additions; buckets far above
//                 the mean size are summed by a whole workgroup, the largest in 64 pieces (skewed
//                 scalars such as 0/1/-1 selector vectors put most points of a window into one bucket).
//   4. reduce   : sum_b b * B_b: lanes take chunks of L buckets (running sums A_t, R_t), then
//                 sum A_t and the bit-masked sums of R_t by LDS trees (sum_t t R_t = sum_i 2^i G_i).
//   5. the few resulting points are combined and normalised on the host: O(100) point operations.
// An SRS of >= 16 points carries a table of its points times 2^(c w): all windows then share ONE
// bucket set, so steps 3-4 run over 2^(c-1) buckets once instead of once per window.
// Integer VALU work throughout; bytes moved are negligible next to the field multiplications.
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "g1.cuh"
#include "g1x.cuh"
#include "msm.hpp"

namespace plonk {

static constexpr uint32_t SIGN_BIT = 0x80000000u;
static constexpr uint32_t BIG_BUCKET = 192;   // points; above max(this, 4 x the mean size) a bucket gets a workgroup
static constexpr uint32_t HUGE_BUCKET = 8192; // points; from max(this, 32 x the mean) on it is cut into pieces
static constexpr uint32_t HUGE_PIECES = 64;   // pieces per huge bucket, a workgroup each
static constexpr uint32_t ORDER_BINS = 256;   // "huge", "big", then bucket sizes from the big threshold down to 0

// thresholds of one MSM: a bucket is big above `big`, huge from `huge`; sizes are binned by >> shift
struct BucketClasses {
    uint32_t big, huge, shift;
};
static constexpr uint32_t CHUNK_LOG_BIG = 3;    // log2 of the buckets per lane in the running-sum step ...
static constexpr uint32_t CHUNK_LOG_SMALL = 0;  // ... none for small bucket sets, where only the depth of the reduction counts:
                                                //     the buckets themselves are the chunks (A_t = R_t = B_{t+1})

// signed c-bit digits of a canonical scalar, least significant window first
struct DigitIter {
    fr_t s;
    uint32_t c, mask, half, carry;
    __device__ __forceinline__ DigitIter(const fr_t& scalar_mont, uint32_t c_) : c(c_) {
        s = f_from_mont<FrP>(scalar_mont);
        mask = (1u << c) - 1u;
        half = 1u << (c - 1);
        carry = 0;
    }
    __device__ __forceinline__ int32_t next() {
        const uint32_t raw = (s.l[0] & mask) + carry;
        int32_t d;
        if (raw > half) { d = (int32_t)raw - (int32_t)(1u << c); carry = 1; }
        else { d = (int32_t)raw; carry = 0; }
#pragma unroll
        for (int k = 0; k < 7; ++k) s.l[k] = (s.l[k] >> c) | (s.l[k + 1] << (32 - c));
        s.l[7] >>= c;
        return d;
    }
};

// ---- 1-3. group the (point, sign) entries by bucket.  A bucket id is gid = set * 2^(c-1) + |digit| - 1,
//          where `shared` mode has a single set (all windows feed ONE bucket set: the points of
//          window w are then the precomputed 2^(c w) P_i of the SRS table) and otherwise window w
//          is set w.  Two levels, both counting in LDS and never with global atomics:
//            coarse: slabs of 4 scalars per lane; workgroup g counts its entries per coarse bin
//                    (gid >> low_bits) into wg_counts[bin][g]; a scan of that matrix gives every
//                    (bin, g) its own output run; the entries go there as (payload, gid low bits).
//            fine:   one workgroup per coarse bin sorts its run by the low bits in LDS and emits
//                    the per-bucket counts / offsets and the final entry list. ----
constexpr uint32_t SORT_LANES = 1024;      // lanes of a coarse workgroup for large inputs, 4 scalars per lane: few, long-lived
                                           // workgroups, so the 8-byte pieces of an output line meet in one L2 before the line
                                           // is evicted.  Small inputs use 256-lane workgroups: they fit beside a bucket kernel
                                           // that has left a CU half empty, where a 16-wave workgroup waits for it to drain.
constexpr uint32_t SORT_PER_LANE = 4;
constexpr uint32_t SORT_MAX_COARSE = 2048; // coarse bins (LDS counters)
constexpr uint32_t SORT_MAX_LOW = 11;      // low bits sorted by the fine pass (2048 LDS counters)

__device__ __forceinline__ uint32_t digit_gid(int32_t d, uint32_t w, uint32_t shared, uint32_t nbk) {
    return (shared ? 0u : w) * nbk + (uint32_t)(d < 0 ? -d : d) - 1u;
}

__global__ __launch_bounds__(SORT_LANES) void msm_coarse_hist_kernel(const fr_t* __restrict__ scalars, uint32_t n, uint32_t c,
                                                               uint32_t W, uint32_t nbk, uint32_t shared, uint32_t low_bits,
                                                               uint32_t n_coarse, uint32_t* __restrict__ wg_counts) {
    __shared__ uint32_t h[SORT_MAX_COARSE];
    for (uint32_t k = threadIdx.x; k < n_coarse; k += blockDim.x) h[k] = 0;
    __syncthreads();
    for (uint32_t r = 0; r < SORT_PER_LANE; ++r) {
        const uint32_t i = (blockIdx.x * SORT_PER_LANE + r) * blockDim.x + threadIdx.x;
        if (i >= n) break;
        DigitIter it(ld(scalars + i), c);
        for (uint32_t w = 0; w < W; ++w) {
            const int32_t d = it.next();
            if (d != 0) atomicAdd(&h[digit_gid(d, w, shared, nbk) >> low_bits], 1u);
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_coarse; k += blockDim.x) wg_counts[(size_t)k * gridDim.x + blockIdx.x] = h[k];
}

// ---- exclusive scan of a vector in tiles of SCAN_TILE entries:
//      tile sums -> scan of the tile sums (one workgroup) -> scan inside each tile ----
constexpr uint32_t SCAN_TILE = 4096;   // 256 lanes x 16 consecutive entries

__global__ __launch_bounds__(256) void msm_scan_tiles_kernel(const uint32_t* __restrict__ counts, uint32_t nb,
                                                              uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t part[256];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    uint32_t sum = 0;
    for (uint32_t k = tile * SCAN_TILE + tid; k < min(nb, (tile + 1) * SCAN_TILE); k += 256) sum += counts[k];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) part[tid] += part[tid + s];
        __syncthreads();
    }
    if (tid == 0) tile_sums[tile] = part[0];
}

__global__ __launch_bounds__(1024) void msm_scan_sums_kernel(uint32_t* __restrict__ v, uint32_t tiles) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (tiles + 1023) / 1024;
    const uint32_t lo = min(tiles, tid * per), hi = min(tiles, lo + per);
    uint32_t sum = 0;
    for (uint32_t k = lo; k < hi; ++k) sum += v[k];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t t = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;  // exclusive
    for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t t = v[k];
        v[k] = run;
        run += t;
    }
}

__global__ __launch_bounds__(256) void msm_scan_kernel(const uint32_t* __restrict__ counts, uint32_t nb,
                                                        const uint32_t* __restrict__ tile_sums,
                                                        uint32_t* __restrict__ offsets) {
    __shared__ uint32_t part[256];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    const uint32_t lo = min(nb, tile * SCAN_TILE + tid * 16), hi = min(nb, lo + 16);
    uint32_t v[16], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        v[k] = lo + k < hi ? counts[lo + k] : 0u;
        sum += v[k];
    }
    part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        uint32_t t = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    uint32_t run = tile_sums[tile] + part[tid] - sum;
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
        if (lo + k < hi) offsets[lo + k] = run;
        run += v[k];
    }
}

// coarse scatter: wg_offsets is the exclusive scan of wg_counts (bin-major)
__global__ __launch_bounds__(SORT_LANES) void msm_coarse_scatter_kernel(const fr_t* __restrict__ scalars, uint32_t n, uint32_t c,
                                                                  uint32_t W, uint32_t nbk, uint32_t shared,
                                                                  uint32_t level_stride, uint32_t low_bits, uint32_t n_coarse,
                                                                  const uint32_t* __restrict__ wg_offsets,
                                                                  uint2* __restrict__ coarse) {
    __shared__ uint32_t cur[SORT_MAX_COARSE];
    for (uint32_t k = threadIdx.x; k < n_coarse; k += blockDim.x) cur[k] = wg_offsets[(size_t)k * gridDim.x + blockIdx.x];
    __syncthreads();
    const uint32_t low_mask = (1u << low_bits) - 1u;
    for (uint32_t r = 0; r < SORT_PER_LANE; ++r) {
        const uint32_t i = (blockIdx.x * SORT_PER_LANE + r) * blockDim.x + threadIdx.x;
        if (i >= n) break;
        DigitIter it(ld(scalars + i), c);
        for (uint32_t w = 0; w < W; ++w) {
            const int32_t d = it.next();
            if (d == 0) continue;
            const uint32_t gid = digit_gid(d, w, shared, nbk);
            const uint32_t pos = atomicAdd(&cur[gid >> low_bits], 1u);
            const uint32_t point = shared ? w * level_stride + i : i;   // row w of the SRS table
            coarse[pos] = make_uint2(point | (d < 0 ? SIGN_BIT : 0u), gid & low_mask);
        }
    }
}

// fine sort of coarse bin blockIdx.x, whose run is [wg_offsets[bin * num_wg], wg_offsets[(bin + 1) * num_wg])
// (wg_offsets has one entry past the matrix, the total).  The run is cut into at most FINE_MAX_PARTS
// parts of about FINE_PART entries, one workgroup (blockIdx.y) each, so that a bin swollen by skewed
// scalars (many equal or small values) still spreads over the chip: the first kernel counts every
// part by low bits, the second turns the counts into cursors and moves the entries.
constexpr uint32_t FINE_PART = 8192;
constexpr uint32_t FINE_MAX_PARTS = 16;

struct FinePart {
    uint32_t lo, hi, parts, sub_lo, sub_hi;
};
__device__ __forceinline__ FinePart fine_part(const uint32_t* wg_offsets, uint32_t num_wg, uint32_t bin, uint32_t part) {
    FinePart f;
    f.lo = wg_offsets[(size_t)bin * num_wg];
    f.hi = wg_offsets[(size_t)(bin + 1) * num_wg];
    const uint32_t size = f.hi - f.lo;
    f.parts = min(FINE_MAX_PARTS, max(1u, (size + FINE_PART - 1) / FINE_PART));
    const uint32_t len = (size + f.parts - 1) / f.parts;
    f.sub_lo = min(f.hi, f.lo + part * len);
    f.sub_hi = min(f.hi, f.sub_lo + len);
    return f;
}

__global__ __launch_bounds__(1024) void msm_fine_hist_kernel(const uint2* __restrict__ coarse,
                                                              const uint32_t* __restrict__ wg_offsets, uint32_t num_wg,
                                                              uint32_t low_bits, uint32_t* __restrict__ part_hist) {
    __shared__ uint32_t h[1u << SORT_MAX_LOW];
    const uint32_t bin = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    const FinePart f = fine_part(wg_offsets, num_wg, bin, part);
    if (part >= f.parts) return;
    const uint32_t nlow = 1u << low_bits;
    for (uint32_t k = tid; k < nlow; k += 1024) h[k] = 0;
    __syncthreads();
    for (uint32_t e = f.sub_lo + tid; e < f.sub_hi; e += 1024) atomicAdd(&h[coarse[e].y], 1u);
    __syncthreads();
    uint32_t* dst = part_hist + (((size_t)bin * FINE_MAX_PARTS + part) << low_bits);
    for (uint32_t k = tid; k < nlow; k += 1024) dst[k] = h[k];
}

__global__ __launch_bounds__(1024) void msm_fine_scatter_kernel(const uint2* __restrict__ coarse,
                                                                 const uint32_t* __restrict__ wg_offsets, uint32_t num_wg,
                                                                 uint32_t n_bins, uint32_t low_bits, uint32_t total_buckets,
                                                                 const uint32_t* __restrict__ part_hist,
                                                                 uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets,
                                                                 uint32_t* __restrict__ sorted) {
    __shared__ uint32_t h[1u << SORT_MAX_LOW];
    __shared__ uint32_t part_sum[1024];
    // workgroup ids go round the 8 XCDs: the parts of one bin take ids 8 apart, so they run at about
    // the same time on ONE XCD and their 4-byte stores into the bin's output meet in that L2
    const uint32_t group = blockIdx.x / (8 * FINE_MAX_PARTS), r = blockIdx.x % (8 * FINE_MAX_PARTS);
    const uint32_t bin = group * 8 + (r & 7u), part = r >> 3, tid = threadIdx.x;
    if (bin >= n_bins) return;
    const FinePart f = fine_part(wg_offsets, num_wg, bin, part);
    if (part >= f.parts) return;
    const uint32_t nlow = 1u << low_bits;
    // per low value: entries of the whole bin (tot) and of the parts before this one (pre)
    const uint32_t per = (nlow + 1023) / 1024;   // 1 or 2
    const uint32_t k0 = min(nlow, tid * per), k1 = min(nlow, k0 + per);
    uint32_t tot[2] = {0, 0}, pre[2] = {0, 0}, sum = 0;
    for (uint32_t k = k0; k < k1; ++k) {
        const uint32_t* src = part_hist + (((size_t)bin * FINE_MAX_PARTS) << low_bits) + k;
        for (uint32_t q = 0; q < f.parts; ++q) {
            const uint32_t v = src[(size_t)q << low_bits];
            tot[k - k0] += v;
            if (q < part) pre[k - k0] += v;
        }
        sum += tot[k - k0];
    }
    part_sum[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t t = tid >= off ? part_sum[tid - off] : 0;
        __syncthreads();
        part_sum[tid] += t;
        __syncthreads();
    }
    uint32_t run = f.lo + part_sum[tid] - sum;
    for (uint32_t k = k0; k < k1; ++k) {
        const uint32_t gid = (bin << low_bits) + k;
        if (part == 0 && gid < total_buckets) {
            counts[gid] = tot[k - k0];
            offsets[gid] = run;
        }
        h[k] = run + pre[k - k0];   // this part's cursor
        run += tot[k - k0];
    }
    __syncthreads();
    for (uint32_t e = f.sub_lo + tid; e < f.sub_hi; e += 1024) {
        const uint2 v = coarse[e];
        sorted[atomicAdd(&h[v.y], 1u)] = v.x;
    }
}

// ---- 4. bucket order: buckets sorted by decreasing size, so the lanes of a wave run equally long ----
__device__ __forceinline__ uint32_t order_bin(uint32_t cnt, BucketClasses k) {  // bin 0 = huge, 1 = big, then by size
    return cnt >= k.huge ? 0u : (cnt > k.big ? 1u : 2u + ((k.big - cnt) >> k.shift));
}
// A workgroup covers ORDER_PER_LANE * 256 buckets: the per-bin totals of a workgroup meet the other
// workgroups' in global atomics on the same few dozen addresses, so there should be few workgroups.
static constexpr uint32_t ORDER_PER_LANE = 8;
__global__ __launch_bounds__(256) void msm_order_hist_kernel(const uint32_t* __restrict__ counts, uint32_t total_buckets,
                                                              BucketClasses classes, uint32_t* __restrict__ bins) {
    __shared__ uint32_t h[ORDER_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t first = blockIdx.x * (256 * ORDER_PER_LANE) + threadIdx.x;
#pragma unroll
    for (uint32_t j = 0; j < ORDER_PER_LANE; ++j) {
        const uint32_t gid = first + j * 256;
        if (gid < total_buckets) atomicAdd(&h[order_bin(counts[gid], classes)], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&bins[threadIdx.x], h[threadIdx.x]);
}
__global__ __launch_bounds__(256) void msm_order_scan_kernel(uint32_t* __restrict__ bins, uint32_t* __restrict__ n_big) {
    __shared__ uint32_t part[ORDER_BINS];
    const uint32_t tid = threadIdx.x;
    const uint32_t v = bins[tid];
    part[tid] = v;
    __syncthreads();
    for (uint32_t off = 1; off < ORDER_BINS; off <<= 1) {
        uint32_t t = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    bins[tid] = part[tid] - v;  // exclusive start of each bin
    if (tid == 0) n_big[1] = v;        // number of huge buckets
    if (tid == 1) n_big[0] = part[1];  // huge + big: where the one-lane-per-bucket part of `order` starts
}
__global__ __launch_bounds__(256) void msm_order_scatter_kernel(const uint32_t* __restrict__ counts,
                                                                 uint32_t total_buckets, BucketClasses classes,
                                                                 uint32_t* __restrict__ bins, uint32_t* __restrict__ order) {
    __shared__ uint32_t h[ORDER_BINS];
    __shared__ uint32_t base[ORDER_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t first = blockIdx.x * (256 * ORDER_PER_LANE) + threadIdx.x;
    uint32_t bin[ORDER_PER_LANE], rank[ORDER_PER_LANE];
#pragma unroll
    for (uint32_t j = 0; j < ORDER_PER_LANE; ++j) {
        const uint32_t gid = first + j * 256;
        bin[j] = 0; rank[j] = 0;
        if (gid < total_buckets) {
            bin[j] = order_bin(counts[gid], classes);
            rank[j] = atomicAdd(&h[bin[j]], 1u);
        }
    }
    __syncthreads();
    if (h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&bins[threadIdx.x], h[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < ORDER_PER_LANE; ++j) {
        const uint32_t gid = first + j * 256;
        if (gid < total_buckets) order[base[bin[j]] + rank[j]] = gid;
    }
}

// an entry is a point index with the sign of its digit in the top bit
__device__ __forceinline__ AffX load_point(const g1x_affine_t* points, uint32_t entry) {
    return ld_affx(points + (entry & ~SIGN_BIT));
}
__device__ __forceinline__ bool entry_neg(uint32_t entry) { return (entry & SIGN_BIT) != 0; }

// ---- 5a. small buckets: one lane per bucket, visited in size order (order[0..n_big) are the big ones) ----
__global__ __launch_bounds__(256) void msm_bucket_kernel(const g1x_affine_t* __restrict__ points,
                                                          const uint32_t* __restrict__ sorted,
                                                          const uint32_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ counts,
                                                          uint32_t total_buckets, const uint32_t* __restrict__ order,
                                                          const uint32_t* __restrict__ n_big, g1x_xyzz_t* __restrict__ buckets) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x + n_big[0];
    if (slot >= total_buckets) return;
    const uint32_t gid = order[slot];
    const uint32_t cnt = counts[gid];
    const uint32_t* src = sorted + offsets[gid];
    PtX acc = ptx_identity();
    const AffX no_point = AffX{x_zero<FqX>(), x_zero<FqX>()};
    // the next point is fetched (a random 128-byte read) while the current one is added
    uint32_t e_next = cnt ? src[0] : 0u;
    AffX nxt = cnt ? load_point(points, e_next) : no_point;
    bool neg_next = entry_neg(e_next);
    e_next = cnt > 1 ? src[1] : 0u;
    uint32_t k = 0;
    if (cnt >= 2) {
        // the first two points are both affine: their sum costs about half a mixed addition.  Should it not apply
        // (an identity, equal x) the lane simply starts the loop below at its first point.
        const AffX second = load_point(points, e_next);
        const uint32_t e2 = cnt > 2 ? src[2] : 0u;
        const AffX third = cnt > 2 ? load_point(points, e2) : no_point;   // on its way while the first sum is taken
        if (ptx_from_two(acc, nxt, neg_next, second, entry_neg(e_next))) {
            k = 2;
            nxt = third;
            neg_next = entry_neg(e2);
            e_next = cnt > 3 ? src[3] : 0u;
        }
    }
    for (; k < cnt; ++k) {
        const AffX p = nxt;
        const bool neg = neg_next;
        if (k + 1 < cnt) {
            nxt = load_point(points, e_next);
            neg_next = entry_neg(e_next);
            if (k + 2 < cnt) e_next = src[k + 2];
        }
        ptx_madd(acc, p, neg);
    }
    st_ptx(buckets + gid, acc);
}

// ---- 5a'. the same for a small bucket set: with fewer buckets than the chip has lanes, 2^LOG_LPB
//          neighbouring lanes share a bucket (entries k = sub, sub + LPB, ...) and fold their sums
//          through LDS, so the dependent chain of additions per lane is LPB times shorter ----
template <uint32_t LOG_LPB>
__global__ __launch_bounds__(256) void msm_bucket_split_kernel(const g1x_affine_t* __restrict__ points,
                                                                const uint32_t* __restrict__ sorted,
                                                                const uint32_t* __restrict__ offsets,
                                                                const uint32_t* __restrict__ counts,
                                                                uint32_t total_buckets, const uint32_t* __restrict__ order,
                                                                const uint32_t* __restrict__ n_big,
                                                                g1x_xyzz_t* __restrict__ buckets) {
    constexpr uint32_t LPB = 1u << LOG_LPB;
    __shared__ g1x_xyzz_t red[256];
    const uint32_t tid = threadIdx.x, sub = tid & (LPB - 1);
    const uint32_t slot = ((blockIdx.x * 256 + tid) >> LOG_LPB) + n_big[0];
    const bool live = slot < total_buckets;
    const uint32_t gid = live ? order[slot] : 0u;
    const uint32_t cnt = live ? counts[gid] : 0u;
    const uint32_t* src = sorted + (live ? offsets[gid] : 0u);
    PtX acc = ptx_identity();
    for (uint32_t k = sub; k < cnt; k += LPB) {
        const uint32_t e = src[k];
        const AffX p = load_point(points, e);
        ptx_madd(acc, p, entry_neg(e));
    }
    // fold the LPB sums of a bucket, every addition run by a pair of its lanes (pair_add)
    const uint32_t pid = tid >> 1, psub = pid & (LPB / 2 - 1);
    const bool odd = (tid & 1u) != 0;
    st_ptx(&red[tid], acc);
    __syncthreads();
    PairPt a = ld_pair(&red[2 * pid], odd);
    const PairPt b = ld_pair(&red[2 * pid + 1], odd);
    __syncthreads();
    pair_add(a, b, odd);
#pragma unroll 1
    for (uint32_t half = LPB >> 2; half > 0; half >>= 1) {
        st_pair(&red[pid], a, odd);
        __syncthreads();
        PairPt o = pair_identity();
        if (psub < half) o = ld_pair(&red[pid + half], odd);
        __syncthreads();
        pair_add(a, o, odd);
    }
    if (live && psub == 0) st_pair(buckets + gid, a, odd);
}

// ---- 5a''. for the smallest inputs even that is a few long dependent chains per SIMD.  Here a PAIR of lanes runs each addition
//          (pair_madd: half the instruction stream), 2^LOG_PPB pairs share a bucket (entries
//          k = sub, sub + PPB, ...) and fold their sums through LDS ----
template <uint32_t LOG_PPB>
__global__ __launch_bounds__(256) void msm_bucket_pair_kernel(const g1x_affine_t* __restrict__ points,
                                                               const uint32_t* __restrict__ sorted,
                                                               const uint32_t* __restrict__ offsets,
                                                               const uint32_t* __restrict__ counts,
                                                               uint32_t total_buckets, const uint32_t* __restrict__ order,
                                                               const uint32_t* __restrict__ n_big,
                                                               g1x_xyzz_t* __restrict__ buckets) {
    constexpr uint32_t PPB = 1u << LOG_PPB;
    __shared__ g1x_xyzz_t red[128];
    const uint32_t tid = threadIdx.x, pid = tid >> 1, sub = pid & (PPB - 1);
    const bool odd = (tid & 1u) != 0;
    const uint32_t slot = ((blockIdx.x * 128 + pid) >> LOG_PPB) + n_big[0];
    const bool live = slot < total_buckets;
    const uint32_t gid = live ? order[slot] : 0u;
    const uint32_t cnt = live ? counts[gid] : 0u;
    const uint32_t* src = sorted + (live ? offsets[gid] : 0u);
    PairPt acc = pair_identity();
    for (uint32_t k = sub; k < cnt; k += PPB) {
        const uint32_t e = src[k];
        const g1x_affine_t* pt = points + (e & ~SIGN_BIT);
        const fqx c = ldx(odd ? &pt->y : &pt->x);
        pair_madd(acc, c, entry_neg(e), odd);
    }
#pragma unroll 1
    for (uint32_t half = PPB >> 1; half > 0; half >>= 1) {
        st_pair(&red[pid], acc, odd);
        __syncthreads();
        PairPt o = pair_identity();
        if (sub < half) o = ld_pair(&red[pid + half], odd);
        __syncthreads();
        pair_add(acc, o, odd);
    }
    if (live && sub == 0) st_pair(buckets + gid, acc, odd);
}

// Sum over the workgroup of the 256 lanes' points, folded by lane pairs (pair_add); the result is
// written to *dst by lanes 0 and 1.
__device__ __forceinline__ void block_sum_256(const PtX& acc, g1x_xyzz_t* red, uint32_t tid, g1x_xyzz_t* dst) {
    const uint32_t pid = tid >> 1;
    const bool odd = (tid & 1u) != 0;
    st_ptx(&red[tid], acc);
    __syncthreads();
    PairPt a = ld_pair(&red[2 * pid], odd);
    const PairPt b = ld_pair(&red[2 * pid + 1], odd);
    __syncthreads();
    pair_add(a, b, odd);
#pragma unroll 1
    for (uint32_t half = 64; half > 0; half >>= 1) {
        st_pair(&red[pid], a, odd);
        __syncthreads();
        PairPt o = pair_identity();
        if (pid < half) o = ld_pair(&red[pid + half], odd);
        __syncthreads();
        pair_add(a, o, odd);
    }
    if (tid < 2) st_pair(dst, a, odd);
}

// ---- 5b. huge buckets (order[0..n_huge)): HUGE_PIECES workgroups each sum a slice of the bucket's
//          entries; skewed scalars (0/1/-1 selector-like vectors, or a narrow top window) put a
//          large share of all points into a handful of buckets ----
__global__ __launch_bounds__(256) void msm_huge_bucket_kernel(const g1x_affine_t* __restrict__ points,
                                                               const uint32_t* __restrict__ sorted,
                                                               const uint32_t* __restrict__ offsets,
                                                               const uint32_t* __restrict__ counts,
                                                               const uint32_t* __restrict__ order,
                                                               const uint32_t* __restrict__ n_big, uint32_t max_huge,
                                                               g1x_xyzz_t* __restrict__ pieces) {
    __shared__ g1x_xyzz_t red[256];
    const uint32_t tid = threadIdx.x, piece = blockIdx.x;
    const uint32_t nhuge = min(n_big[1], max_huge);
    for (uint32_t item = blockIdx.y; item < nhuge; item += gridDim.y) {
        const uint32_t gid = order[item];
        const uint32_t cnt = counts[gid];
        const uint32_t* src = sorted + offsets[gid];
        const uint32_t len = (cnt + HUGE_PIECES - 1) / HUGE_PIECES;
        const uint32_t lo = min(cnt, piece * len), hi = min(cnt, lo + len);
        PtX acc = ptx_identity();
        for (uint32_t k = lo + tid; k < hi; k += 256) {
            const uint32_t e = src[k];
            const AffX p = load_point(points, e);
            ptx_madd(acc, p, entry_neg(e));
        }
        block_sum_256(acc, red, tid, pieces + (size_t)item * HUGE_PIECES + piece);
        __syncthreads();
    }
}

// ---- 5c. big buckets: one workgroup per bucket; for a huge bucket it adds up the pieces ----
__global__ __launch_bounds__(256) void msm_big_bucket_kernel(const g1x_affine_t* __restrict__ points,
                                                              const uint32_t* __restrict__ sorted,
                                                              const uint32_t* __restrict__ offsets,
                                                              const uint32_t* __restrict__ counts,
                                                              const uint32_t* __restrict__ order,
                                                              const uint32_t* __restrict__ n_big, uint32_t max_huge,
                                                              const g1x_xyzz_t* __restrict__ pieces,
                                                              g1x_xyzz_t* __restrict__ buckets) {
    __shared__ g1x_xyzz_t red[256];
    const uint32_t tid = threadIdx.x;
    const uint32_t nbig = n_big[0], nhuge = min(n_big[1], max_huge);
    for (uint32_t item = blockIdx.x; item < nbig; item += gridDim.x) {
        const uint32_t gid = order[item];
        PtX acc = ptx_identity();
        if (item < nhuge) {
            if (tid < HUGE_PIECES) acc = ld_ptx(pieces + (size_t)item * HUGE_PIECES + tid);
        } else {
            const uint32_t cnt = counts[gid];
            const uint32_t* src = sorted + offsets[gid];
            for (uint32_t k = tid; k < cnt; k += 256) {
                const uint32_t e = src[k];
                const AffX p = load_point(points, e);
                ptx_madd(acc, p, entry_neg(e));
            }
        }
        block_sum_256(acc, red, tid, buckets + gid);
        __syncthreads();
    }
}

// ---- 6a. running sums over chunks of L = 2^chunk_log buckets: lane (w, t) covers b in (tL, (t+1)L]
//          A_t = sum (b - tL) B_b,  R_t = sum B_b.   sum_b b B_b = sum_t A_t + L sum_t t R_t ----
__global__ __launch_bounds__(256) void msm_chunk_kernel(const g1x_xyzz_t* __restrict__ buckets, uint32_t nbk,
                                                         uint32_t chunk_log, uint32_t chunks_per_window, uint32_t total_chunks,
                                                         g1x_xyzz_t* __restrict__ A, g1x_xyzz_t* __restrict__ R) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total_chunks) return;
    const uint32_t w = gid / chunks_per_window, t = gid % chunks_per_window;
    const g1x_xyzz_t* B = buckets + (size_t)w * nbk + ((size_t)t << chunk_log);
    PtX running = ptx_identity(), acc = ptx_identity();
    // running += B_j, then acc += running: one expansion of the addition serves both (its ~55 KB of
    // straight-line code is most of the instruction cache), the operands are switched around it.
    // The next bucket is fetched while acc += running runs: a lane is alone on its SIMD here.
    PtX next = ld_ptx(B + ((1u << chunk_log) - 1u));
#pragma unroll 1
    for (uint32_t step = 0; step < (2u << chunk_log); ++step) {
        const bool take = (step & 1u) == 0;
        PtX x = take ? running : acc;
        PtX y = take ? next : running;
        if (!take && step + 1 < (2u << chunk_log)) next = ld_ptx(B + ((1u << chunk_log) - 2u - (step >> 1)));
        ptx_add<MulInline>(x, y);
        if (take) running = x;
        else acc = x;
    }
    st_ptx(A + gid, acc);
    st_ptx(R + gid, running);
}

// ---- 6b/6c. sum_t A_t and the bit-masked sums G_i = sum of R_t over the chunks t with bit i set (so
//          that sum_t t R_t = sum_i 2^i G_i), through the view t = hi * 2^ml + lo of the chunk index:
//          a low bit of t is a bit of lo, so its G_i is a masked sum of the 2^ml column sums
//          Col_lo = sum_hi R[hi][lo]; a high bit is a bit of hi and goes through the row sums
//          Row_hi = sum_lo R[hi][lo].  Two adds per chunk instead of one per chunk and bit.
//          6b: one workgroup per row (R and A) or column; 6c: one workgroup per quantity. ----
// Sum of `count` points src[index(k)] over a workgroup of LANES lanes = LANES / 4 quads, every
// addition run by a quad (quad_add): each quad first adds its share, then the quads' sums are folded
// through LDS.  Both phases go through ONE expansion of the addition: in these short kernels a good
// part of the time is the first pass over its straight-line code, so a second copy would cost
// nearly as much again.  Result in quad 0 (lanes 0..3, lane j holding coordinate j).
template <uint32_t LANES, class Index>
__device__ __forceinline__ fqx reduce_points(const g1x_xyzz_t* __restrict__ src, uint32_t count, Index index,
                                             g1x_xyzz_t* red, uint32_t tid) {
    constexpr uint32_t QUADS = LANES / 4;
    const uint32_t qid = tid >> 2, j = tid & 3u;
    const uint32_t n_serial = (count + QUADS - 1) / QUADS;
    uint32_t levels = 0;
    while ((1u << levels) < QUADS) ++levels;
    fqx acc = quad_identity();
#pragma unroll 1
    for (uint32_t step = 0; step < n_serial + levels; ++step) {
        fqx o = quad_identity();
        if (step < n_serial) {
            const uint32_t k = qid + step * QUADS;
            if (k < count) o = ld_quad(src + index(k), j);
        } else {
            const uint32_t half = QUADS >> (step - n_serial + 1);
            st_quad(&red[qid], acc, j);
            __syncthreads();
            if (qid < half) o = ld_quad(&red[qid + half], j);
            __syncthreads();
        }
        quad_add(acc, o, j);   // adding the identity returns at once
    }
    return acc;
}

constexpr uint32_t ROWCOL_LANES = 128, MASKED_LANES = 512;

__global__ __launch_bounds__(ROWCOL_LANES) void msm_rowcol_kernel(const g1x_xyzz_t* __restrict__ A, const g1x_xyzz_t* __restrict__ R,
                                                                   uint32_t m, uint32_t ml, g1x_xyzz_t* __restrict__ row,
                                                                   g1x_xyzz_t* __restrict__ arow, g1x_xyzz_t* __restrict__ col) {
    __shared__ g1x_xyzz_t red[ROWCOL_LANES / 4];
    const uint32_t rows = 1u << (m - ml), cols = 1u << ml, chunks = 1u << m;
    const uint32_t w = blockIdx.y, tid = threadIdx.x;
    uint32_t job = blockIdx.x;
    const g1x_xyzz_t* src;
    g1x_xyzz_t* dst;
    uint32_t first, stride, count;
    if (job < 2 * rows) {            // a row of R (job < rows) or of A: 2^ml consecutive chunks
        const bool of_a = job >= rows;
        if (of_a) job -= rows;
        src = (of_a ? A : R) + (size_t)w * chunks;
        dst = (of_a ? arow : row) + (size_t)w * rows + job;
        first = job << ml; stride = 1; count = cols;
    } else {                         // a column of R: every 2^ml-th chunk
        job -= 2 * rows;
        src = R + (size_t)w * chunks;
        dst = col + (size_t)w * cols + job;
        first = job; stride = cols; count = rows;
    }
    const fqx acc = reduce_points<ROWCOL_LANES>(src, count, [=](uint32_t k) { return first + (size_t)k * stride; }, red, tid);
    if (tid < 4) st_quad(dst, acc, tid);
}

__global__ __launch_bounds__(MASKED_LANES) void msm_masked_kernel(const g1x_xyzz_t* __restrict__ row, const g1x_xyzz_t* __restrict__ arow,
                                                                   const g1x_xyzz_t* __restrict__ col, uint32_t m, uint32_t ml,
                                                                   g1_xyzz_t* __restrict__ out) {
    __shared__ g1x_xyzz_t red[MASKED_LANES / 4];
    const uint32_t rows = 1u << (m - ml), cols = 1u << ml;
    const uint32_t q = blockIdx.x, nq = gridDim.x, w = blockIdx.y, tid = threadIdx.x;
    const g1x_xyzz_t* src;
    uint32_t count, bit = 0;
    if (q == 0) {
        src = arow + (size_t)w * rows; count = rows;
    } else if (q - 1 < ml) {
        src = col + (size_t)w * cols; count = cols >> 1; bit = q - 1;
    } else {
        src = row + (size_t)w * rows; count = rows >> 1; bit = q - 1 - ml;
    }
    // q > 0: the k-th index with bit `bit` set
    const fqx acc = reduce_points<MASKED_LANES>(src, count, [=](uint32_t k) -> size_t {
        return q == 0 ? k : (((k >> bit) << (bit + 1)) | (1u << bit) | (k & ((1u << bit) - 1u)));
    }, red, tid);
    if (tid >= 4) return;
    // back to the saturated form for the host: lane j holds coordinate j
    g1_xyzz_t* dst = out + (size_t)w * nq + q;
    st(&dst->x + tid, x_to_sat<FqX>(acc));
}

// saturated affine -> x-form affine
__global__ __launch_bounds__(256) void g1_to_xform_kernel(const g1_affine_t* __restrict__ in, uint32_t n,
                                                           g1x_affine_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const g1_affine_t p = ld_affine(in + i);
    AffX q;
    if (affine_is_inf(p)) {
        q.x = x_zero<FqX>();
        q.y = x_zero<FqX>();
    } else {
        q.x = x_from_sat<FqX>(p.x);
        q.y = x_from_sat<FqX>(p.y);
    }
    stx(&out[i].x, q.x);
    stx(&out[i].y, q.y);
}

// ------------------------------------------------------------------------------------------
uint32_t msm_window_bits(size_t n) {
    uint32_t lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) ++lg;
    int c = (int)lg - 4;
    return (uint32_t)std::max(4, std::min(16, c));
}

static host::G1XyzzH xyzz_to_host(const g1_xyzz_t& d) {
    host::G1XyzzH h;
    memcpy(h.x.l, d.x.l, 48); memcpy(h.y.l, d.y.l, 48); memcpy(h.zz.l, d.zz.l, 48); memcpy(h.zzz.l, d.zzz.l, 48);
    return h;
}

// One in-flight MSM: its own buffers and events.  The three phases of an MSM run on three streams
// shared by all slots -- sort (histogram, scan, scatter, bucket order: atomics and scattered stores),
// buckets (integer-VALU bound) and tail (latency-bound reductions) -- so the sort of MSM k+1 and the
// tail of MSM k-1 run underneath the bucket accumulation of MSM k.
struct MsmSlot {
    hipEvent_t sorted_ev = nullptr, buckets_ev = nullptr, done_ev = nullptr;   // order the phases
    hipEvent_t t_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // timing marks (only when enabled)
    bool timed = false;
    uint32_t* host_entries = nullptr;   // pinned: (point, sign) entries of the last run
    DevBuf<uint32_t> sorted, counts, offsets, wg_counts, wg_offsets, tile_sums, part_hist, order, bins;
    DevBuf<uint2> coarse;
    DevBuf<g1x_xyzz_t> buckets, pieces, A, R, row, arow, col;
    DevBuf<g1_xyzz_t> sums;
    g1_xyzz_t* host_sums = nullptr;   // pinned, so the device-to-host copy stays asynchronous
    size_t host_cap = 0;
    // sharded MSM: this rank's record (header + sums), the records of all ranks, and their pinned landing buffer
    DevBuf<unsigned char> record, gathered;
    unsigned char* host_gathered = nullptr;
    uint32_t* host_hdr = nullptr;     // pinned staging of the record header
    size_t host_gathered_cap = 0;
    uint32_t world = 0;               // ranks whose records host_gathered holds; 0 = a local MSM
    uint32_t c = 0, W = 0, nq = 0, log_chunks = 0, chunk_log = 0;
    bool empty = true;
    ~MsmSlot() {
        if (host_sums) (void)hipHostFree(host_sums);
        if (host_gathered) (void)hipHostFree(host_gathered);
        if (host_hdr) (void)hipHostFree(host_hdr);
        if (host_entries) (void)hipHostFree(host_entries);
        for (auto& e : t_ev)
            if (e) (void)hipEventDestroy(e);
        if (sorted_ev) (void)hipEventDestroy(sorted_ev);
        if (buckets_ev) (void)hipEventDestroy(buckets_ev);
        if (done_ev) (void)hipEventDestroy(done_ev);
    }
};

struct MsmWorkspace {
    static constexpr int SLOTS = 3;
    bool timing = false;   // record the timing marks (msm_set_timing)
    MsmSlot slot[SLOTS];
    hipStream_t sort_stream = nullptr, bucket_stream = nullptr;
    hipEvent_t ready = nullptr, filled = nullptr;
    ~MsmWorkspace() {
        if (ready) (void)hipEventDestroy(ready);
        if (filled) (void)hipEventDestroy(filled);
        if (sort_stream) (void)hipStreamDestroy(sort_stream);
        if (bucket_stream) (void)hipStreamDestroy(bucket_stream);
    }
};

static MsmWorkspace* msm_workspace(Ctx* ctx) {
    if (!ctx->msm_ws) {
        MsmWorkspace* ws = new MsmWorkspace();
        ctx->msm_ws = std::shared_ptr<void>(ws, [](void* p) { delete static_cast<MsmWorkspace*>(p); });
        HIP_CHECK(hipStreamCreateWithFlags(&ws->sort_stream, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&ws->bucket_stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&ws->ready, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ws->filled, hipEventDisableTiming));
        for (auto& s : ws->slot) {
            HIP_CHECK(hipEventCreateWithFlags(&s.sorted_ev, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&s.buckets_ev, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&s.done_ev, hipEventDisableTiming));
            for (auto& e : s.t_ev) HIP_CHECK(hipEventCreate(&e));
            HIP_CHECK(hipHostMalloc((void**)&s.host_entries, sizeof(uint32_t), hipHostMallocDefault));
        }
    }
    return static_cast<MsmWorkspace*>(ctx->msm_ws.get());
}

// One rank's share of a sharded MSM as it is all-gathered: a 64-byte header (empty flag, window width,
// bucket sets, sums per set, log2 chunks, log2 buckets per chunk) and the sums of the reduction,
// at most 256 of them (W <= 64 windows of at most 4 sums at c = 4, one set of at most 20 with the
// window table).  The ranks' shares may differ in size and hence in geometry, so each record
// describes itself.
static constexpr size_t MSM_REC_HDR = 64;
static constexpr size_t MSM_REC_SUMS = 256;
static constexpr size_t MSM_REC_BYTES = MSM_REC_HDR + MSM_REC_SUMS * sizeof(g1_xyzz_t);

static void msm_exchange(Ctx* ctx, MsmSlot& s, Exchange* ex, bool empty) {
    hipStream_t st = ctx->stream;
    if (!s.host_hdr) HIP_CHECK(hipHostMalloc((void**)&s.host_hdr, MSM_REC_HDR, hipHostMallocDefault));
    s.record.ensure(MSM_REC_BYTES);
    s.gathered.ensure(MSM_REC_BYTES * ex->world);
    if (MSM_REC_BYTES * ex->world > s.host_gathered_cap) {
        if (s.host_gathered) (void)hipHostFree(s.host_gathered);
        s.host_gathered = nullptr;
        HIP_CHECK(hipHostMalloc((void**)&s.host_gathered, MSM_REC_BYTES * ex->world, hipHostMallocDefault));
        s.host_gathered_cap = MSM_REC_BYTES * ex->world;
    }
    memset(s.host_hdr, 0, MSM_REC_HDR);
    s.host_hdr[0] = empty ? 1u : 0u;
    s.host_hdr[1] = s.c; s.host_hdr[2] = s.W; s.host_hdr[3] = s.nq; s.host_hdr[4] = s.log_chunks; s.host_hdr[5] = s.chunk_log;
    HIP_CHECK(hipMemcpyAsync(s.record.p, s.host_hdr, MSM_REC_HDR, hipMemcpyHostToDevice, st));
    ex->allgather(s.record.p, s.gathered.p, MSM_REC_BYTES, st);
    HIP_CHECK(hipMemcpyAsync(s.host_gathered, s.gathered.p, MSM_REC_BYTES * ex->world, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipEventRecord(s.done_ev, st));
}

// Enqueue one MSM on slot `si`.  The scalars must be ready on ctx->stream (an event orders the sort
// stream behind it).
enum MsmPhase { PH_SORT = 1, PH_BUCKETS = 2, PH_TAIL = 4 };   // an MSM is enqueued phase by phase (sort and buckets usually together)
static void msm_enqueue(Ctx* ctx, MsmWorkspace* ws, int si, const MsmPoints& pts, const fr_t* scalars, size_t n,
                        uint32_t phases, Exchange* ex) {
    const bool tail_phase = (phases & PH_TAIL) != 0;
    const g1x_affine_t* points = pts.px;
    // the shared bucket set is used unless the input is a sliver of the SRS (n < 2^(table_c - 12))
    const uint32_t shared = (pts.table_c && (n >> (pts.table_c > 12 ? pts.table_c - 12 : 0)) != 0) ? 1u : 0u;
    MsmSlot& s = ws->slot[si];
    s.empty = n == 0;
    s.world = ex ? ex->world : 0;
    if (n == 0) {
        if (ex && tail_phase) msm_exchange(ctx, s, ex, true);   // an empty share still takes part in the all-gather
        return;
    }
    PLONK_REQUIRE(n < ((size_t)1 << 31), "msm: too many points");
    const uint32_t c = shared ? pts.table_c : msm_window_bits(n);
    const uint32_t W = (256 + c - 1) / c;
    PLONK_REQUIRE((uint64_t)W * n < ((uint64_t)1 << 32), "msm: too many (point, window) entries for 32-bit offsets");
    const uint32_t S = shared ? 1u : W;   // bucket sets
    const uint32_t nbk = 1u << (c - 1);
    const uint32_t total_buckets = S * nbk;
    // large bucket sets: the reduction is work-bound (8 buckets per lane, 2048 chunks per tree
    // workgroup); small ones: it is a chain of dependent additions, kept as short as possible
    const bool small_set = total_buckets < (1u << 18);
    const uint32_t chunk_log = std::min(c - 1, small_set ? CHUNK_LOG_SMALL : CHUNK_LOG_BIG);
    const uint32_t chunks = nbk >> chunk_log, total_chunks = S * chunks;
    uint32_t log_chunks = 0;
    while ((1u << log_chunks) < chunks) ++log_chunks;
    const uint32_t nq = 1 + log_chunks;
    s.c = c; s.W = S; s.nq = nq; s.log_chunks = log_chunks; s.chunk_log = chunk_log;
    // sort geometry: low_bits of the bucket id are sorted per coarse bin, the rest select the bin
    uint32_t kbits = 0;
    while ((1u << kbits) < total_buckets) ++kbits;
    const uint32_t low_bits = std::min(SORT_MAX_LOW, std::max(2u, kbits > 9 ? kbits - 9 : 0u));
    const uint32_t n_coarse = (total_buckets + (1u << low_bits) - 1) >> low_bits;
    PLONK_REQUIRE(n_coarse <= SORT_MAX_COARSE, "msm: too many buckets for the sort");
    const uint32_t sort_lanes = n < ((size_t)1 << 19) ? 256u : SORT_LANES, sort_slab = sort_lanes * SORT_PER_LANE;
    const uint32_t num_wg = (uint32_t)((n + sort_slab - 1) / sort_slab);
    const uint32_t matrix = n_coarse * num_wg + 1;   // + the total
    const uint32_t tiles = (matrix + SCAN_TILE - 1) / SCAN_TILE;
    s.sorted.ensure((size_t)W * n);
    s.coarse.ensure((size_t)W * n);
    s.wg_counts.ensure(matrix);
    s.wg_offsets.ensure(matrix);
    s.tile_sums.ensure(tiles);
    s.part_hist.ensure(((size_t)n_coarse * FINE_MAX_PARTS) << low_bits);
    s.counts.ensure(total_buckets);
    s.offsets.ensure(total_buckets);
    s.order.ensure(total_buckets);
    s.bins.ensure(ORDER_BINS + 4);
    s.buckets.ensure(total_buckets);
    // size classes relative to the mean bucket size; at most entries / huge buckets can be huge
    const uint32_t mean = (uint32_t)(((size_t)W * n) / total_buckets);
    BucketClasses classes;
    classes.big = std::max(BIG_BUCKET, 4 * mean);
    classes.huge = std::max(HUGE_BUCKET, 32 * mean);
    classes.shift = 0;
    while ((classes.big >> classes.shift) > ORDER_BINS - 3) ++classes.shift;
    const uint32_t max_huge = (uint32_t)(((size_t)W * n) / classes.huge) + 1;
    s.pieces.ensure((size_t)max_huge * HUGE_PIECES);
    s.A.ensure(total_chunks);
    s.R.ensure(total_chunks);
    const uint32_t ml = (log_chunks + 1) / 2, rows = 1u << (log_chunks - ml), cols = 1u << ml;
    s.row.ensure((size_t)S * rows);
    s.arow.ensure((size_t)S * rows);
    s.col.ensure((size_t)S * cols);
    s.sums.ensure((size_t)S * nq);
    PLONK_REQUIRE(!ex || (size_t)S * nq <= MSM_REC_SUMS, "msm: too many window sums for the exchange record");
    if ((size_t)S * nq > s.host_cap) {
        if (s.host_sums) (void)hipHostFree(s.host_sums);
        s.host_sums = nullptr;
        HIP_CHECK(hipHostMalloc((void**)&s.host_sums, (size_t)S * nq * sizeof(g1_xyzz_t), hipHostMallocDefault));
        s.host_cap = (size_t)S * nq;
    }
    const uint32_t bblk = (total_buckets + 255) / 256;
    uint32_t* n_big = s.bins.p + ORDER_BINS;
    hipStream_t st;
    if (phases & PH_SORT) {
    // ---- sort phase (the caller has ordered the sort stream behind the scalars) ----
    st = ws->sort_stream;
    s.timed = ws->timing;
    if (s.timed) HIP_CHECK(hipEventRecord(s.t_ev[0], st));
    HIP_CHECK(hipMemsetAsync(s.wg_counts.p + (matrix - 1), 0, sizeof(uint32_t), st));
    HIP_CHECK(hipMemsetAsync(s.bins.p, 0, (ORDER_BINS + 4) * sizeof(uint32_t), st));
    hipLaunchKernelGGL(msm_coarse_hist_kernel, dim3(num_wg), dim3(sort_lanes), 0, st, scalars, (uint32_t)n, c, W, nbk, shared, low_bits,
                       n_coarse, s.wg_counts.p);
    hipLaunchKernelGGL(msm_scan_tiles_kernel, dim3(tiles), dim3(256), 0, st, s.wg_counts.p, matrix, s.tile_sums.p);
    hipLaunchKernelGGL(msm_scan_sums_kernel, dim3(1), dim3(1024), 0, st, s.tile_sums.p, tiles);
    hipLaunchKernelGGL(msm_scan_kernel, dim3(tiles), dim3(256), 0, st, s.wg_counts.p, matrix, s.tile_sums.p, s.wg_offsets.p);
    hipLaunchKernelGGL(msm_coarse_scatter_kernel, dim3(num_wg), dim3(sort_lanes), 0, st, scalars, (uint32_t)n, c, W, nbk, shared,
                       (uint32_t)pts.level_stride, low_bits, n_coarse, s.wg_offsets.p, s.coarse.p);
    hipLaunchKernelGGL(msm_fine_hist_kernel, dim3(n_coarse, FINE_MAX_PARTS), dim3(1024), 0, st, s.coarse.p, s.wg_offsets.p,
                       num_wg, low_bits, s.part_hist.p);
    hipLaunchKernelGGL(msm_fine_scatter_kernel, dim3(((n_coarse + 7) / 8) * 8 * FINE_MAX_PARTS), dim3(1024), 0, st, s.coarse.p,
                       s.wg_offsets.p, num_wg, n_coarse, low_bits, total_buckets, s.part_hist.p, s.counts.p, s.offsets.p, s.sorted.p);
    const uint32_t oblk = (total_buckets + 256 * ORDER_PER_LANE - 1) / (256 * ORDER_PER_LANE);
    hipLaunchKernelGGL(msm_order_hist_kernel, dim3(oblk), dim3(256), 0, st, s.counts.p, total_buckets, classes, s.bins.p);
    hipLaunchKernelGGL(msm_order_scan_kernel, dim3(1), dim3(256), 0, st, s.bins.p, n_big);
    hipLaunchKernelGGL(msm_order_scatter_kernel, dim3(oblk), dim3(256), 0, st, s.counts.p, total_buckets, classes, s.bins.p,
                       s.order.p);
    if (s.timed) {
        HIP_CHECK(hipMemcpyAsync(s.host_entries, s.wg_offsets.p + (matrix - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipEventRecord(s.t_ev[1], st));
    }
    HIP_CHECK(hipEventRecord(s.sorted_ev, st));
    }
    if (phases & PH_BUCKETS) {
    // ---- bucket phase ----
    st = ws->bucket_stream;
    HIP_CHECK(hipStreamWaitEvent(st, s.sorted_ev, 0));
    if (s.timed) HIP_CHECK(hipEventRecord(s.t_ev[2], st));
    if (total_buckets < (1u << 17) && (size_t)W * n <= ((size_t)1 << 19)) {
        // few entries: all latency.  Pairs per bucket: up to 2^18 lanes in all, at least ~4 additions per pair
        uint32_t log_ppb = 0;
        while (log_ppb < 3 && (total_buckets << (log_ppb + 2)) <= (1u << 18) && (mean >> (log_ppb + 1)) >= 4) ++log_ppb;
        const uint32_t pblk = (uint32_t)((((size_t)total_buckets << log_ppb) + 127) / 128);
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(pblk), dim3(256), 0, st, points, s.sorted.p, s.offsets.p, s.counts.p, total_buckets,
                               s.order.p, n_big, s.buckets.p);
        };
        if (log_ppb == 0) launch(msm_bucket_pair_kernel<0>);
        else if (log_ppb == 1) launch(msm_bucket_pair_kernel<1>);
        else if (log_ppb == 2) launch(msm_bucket_pair_kernel<2>);
        else launch(msm_bucket_pair_kernel<3>);
    } else {
        // lanes per bucket: a small bucket set with long buckets is spread over more lanes
        uint32_t log_lpb = 0;
        while (log_lpb < 3 && (total_buckets << log_lpb) < (1u << 18) && (mean >> log_lpb) >= 16) ++log_lpb;
        const uint32_t sblk = (uint32_t)((((size_t)total_buckets << log_lpb) + 255) / 256);
        if (log_lpb == 0)
            hipLaunchKernelGGL(msm_bucket_kernel, dim3(bblk), dim3(256), 0, st, points, s.sorted.p, s.offsets.p, s.counts.p,
                               total_buckets, s.order.p, n_big, s.buckets.p);
        else if (log_lpb == 1)
            hipLaunchKernelGGL(msm_bucket_split_kernel<1>, dim3(sblk), dim3(256), 0, st, points, s.sorted.p, s.offsets.p,
                               s.counts.p, total_buckets, s.order.p, n_big, s.buckets.p);
        else if (log_lpb == 2)
            hipLaunchKernelGGL(msm_bucket_split_kernel<2>, dim3(sblk), dim3(256), 0, st, points, s.sorted.p, s.offsets.p,
                               s.counts.p, total_buckets, s.order.p, n_big, s.buckets.p);
        else
            hipLaunchKernelGGL(msm_bucket_split_kernel<3>, dim3(sblk), dim3(256), 0, st, points, s.sorted.p, s.offsets.p,
                               s.counts.p, total_buckets, s.order.p, n_big, s.buckets.p);
    }
    if (s.timed) HIP_CHECK(hipEventRecord(s.t_ev[3], st));
    // The oversized buckets come AFTER the lane-per-bucket kernel (they write other buckets, so the
    // order is free): two nearly empty launches ahead of it would each have to drain through a chip
    // that the previous MSM's reduction is using, and hold the big kernel back by that long.
    hipLaunchKernelGGL(msm_huge_bucket_kernel, dim3(HUGE_PIECES, 16), dim3(256), 0, st, points, s.sorted.p, s.offsets.p,
                       s.counts.p, s.order.p, n_big, max_huge, s.pieces.p);
    hipLaunchKernelGGL(msm_big_bucket_kernel, dim3(512), dim3(256), 0, st, points, s.sorted.p, s.offsets.p, s.counts.p,
                       s.order.p, n_big, max_huge, s.pieces.p, s.buckets.p);
    if (s.timed) HIP_CHECK(hipEventRecord(s.t_ev[5], st));
    HIP_CHECK(hipEventRecord(s.buckets_ev, st));
    }
    if (!tail_phase) return;
    // ---- tail phase, on the ctx stream (idle while its commitments are in flight) ----
    st = ctx->stream;
    HIP_CHECK(hipStreamWaitEvent(st, s.buckets_ev, 0));
    const g1x_xyzz_t *chunk_a = s.buckets.p, *chunk_r = s.buckets.p;
    if (chunk_log != 0) {
        hipLaunchKernelGGL(msm_chunk_kernel, dim3((total_chunks + 255) / 256), dim3(256), 0, st, s.buckets.p, nbk, chunk_log, chunks,
                           total_chunks, s.A.p, s.R.p);
        chunk_a = s.A.p;
        chunk_r = s.R.p;
    }
    hipLaunchKernelGGL(msm_rowcol_kernel, dim3(2 * rows + cols, S), dim3(ROWCOL_LANES), 0, st, chunk_a, chunk_r, log_chunks, ml, s.row.p,
                       s.arow.p, s.col.p);
    g1_xyzz_t* sums_out = s.sums.p;
    if (ex) {
        s.record.ensure(MSM_REC_BYTES);
        sums_out = reinterpret_cast<g1_xyzz_t*>(s.record.p + MSM_REC_HDR);
    }
    hipLaunchKernelGGL(msm_masked_kernel, dim3(nq, S), dim3(MASKED_LANES), 0, st, s.row.p, s.arow.p, s.col.p, log_chunks, ml, sums_out);
    HIP_CHECK(hipGetLastError());
    if (ex) {
        msm_exchange(ctx, s, ex, false);
        if (s.timed) HIP_CHECK(hipEventRecord(s.t_ev[4], st));
        return;
    }
    HIP_CHECK(hipMemcpyAsync(s.host_sums, s.sums.p, (size_t)S * nq * sizeof(g1_xyzz_t), hipMemcpyDeviceToHost, st));
    if (s.timed) HIP_CHECK(hipEventRecord(s.t_ev[4], st));
    HIP_CHECK(hipEventRecord(s.done_ev, st));
}

// S_w = A_w + L * sum_i 2^i G_{w,i}; result = sum_w 2^(c w) S_w, from the sums of one reduction
static host::G1XyzzH combine_sums(const host::G1XyzzH* sums, uint32_t c, uint32_t W, uint32_t nq, uint32_t log_chunks,
                                  uint32_t chunk_log) {
    host::G1XyzzH acc = host::g1_identity();
    for (int w = (int)W - 1; w >= 0; --w) {
        if (w != (int)W - 1)
            for (uint32_t k = 0; k < c; ++k) acc = host::g1_double(acc);
        const host::G1XyzzH* q = sums + (size_t)w * nq;
        host::G1XyzzH g = host::g1_identity();
        for (int i = (int)log_chunks - 1; i >= 0; --i) {
            g = host::g1_double(g);
            g = host::g1_add(g, q[1 + i]);
        }
        for (uint32_t k = 0; k < chunk_log; ++k) g = host::g1_double(g);
        g = host::g1_add(g, q[0]);
        acc = host::g1_add(acc, g);
    }
    return acc;
}
static host::G1XyzzH combine_sums(const g1_xyzz_t* sums, uint32_t c, uint32_t W, uint32_t nq, uint32_t log_chunks,
                                  uint32_t chunk_log) {
    std::vector<host::G1XyzzH> h((size_t)W * nq);
    for (size_t i = 0; i < h.size(); ++i) h[i] = xyzz_to_host(sums[i]);
    return combine_sums(h.data(), c, W, nq, log_chunks, chunk_log);
}

// Wait for slot `si` and combine on the host; a sharded MSM adds up the shares of all ranks, in rank order.
static host::G1AffineH msm_collect(MsmWorkspace* ws, int si) {
    MsmSlot& s = ws->slot[si];
    host::G1AffineH ident;
    ident.x = host::Fq::zero(); ident.y = host::Fq::zero(); ident.inf = true;
    if (s.world == 0) {
        if (s.empty) return ident;
        HIP_CHECK(hipEventSynchronize(s.done_ev));
        return host::g1_to_affine(combine_sums(s.host_sums, s.c, s.W, s.nq, s.log_chunks, s.chunk_log));
    }
    HIP_CHECK(hipEventSynchronize(s.done_ev));
    // The combination is linear in the window sums: shares reduced with the same geometry (the usual case: equal
    // slices) are first added sum by sum, and the doublings of the combination are paid once for all of them.
    struct Group {
        uint32_t hdr[6];
        std::vector<host::G1XyzzH> sums;
    };
    std::vector<Group> groups;
    for (uint32_t r = 0; r < s.world; ++r) {
        const unsigned char* rec = s.host_gathered + (size_t)r * MSM_REC_BYTES;
        uint32_t hdr[6];
        memcpy(hdr, rec, sizeof(hdr));
        if (hdr[0]) continue;
        if (hdr[1] < 1 || hdr[1] > 31 || hdr[2] < 1 || hdr[3] < 1 || (size_t)hdr[2] * hdr[3] > MSM_REC_SUMS ||
            hdr[4] + 1 != hdr[3] || hdr[5] > 31)
            throw Error(PLONK_E_INTERNAL, "msm: malformed share in the all-gathered records");
        const g1_xyzz_t* sums = reinterpret_cast<const g1_xyzz_t*>(rec + MSM_REC_HDR);
        const size_t count = (size_t)hdr[2] * hdr[3];
        Group* g = nullptr;
        for (Group& have : groups)
            if (memcmp(have.hdr + 1, hdr + 1, 5 * sizeof(uint32_t)) == 0) g = &have;
        if (!g) {
            groups.emplace_back();
            g = &groups.back();
            memcpy(g->hdr, hdr, sizeof(hdr));
            g->sums.resize(count);
            for (size_t i = 0; i < count; ++i) g->sums[i] = xyzz_to_host(sums[i]);
        } else {
            for (size_t i = 0; i < count; ++i) g->sums[i] = host::g1_add(g->sums[i], xyzz_to_host(sums[i]));
        }
    }
    host::G1XyzzH acc = host::g1_identity();
    for (const Group& g : groups) acc = host::g1_add(acc, combine_sums(g.sums.data(), g.hdr[1], g.hdr[2], g.hdr[3], g.hdr[4], g.hdr[5]));
    return host::g1_to_affine(acc);
}

host::G1AffineH msm_run(Ctx* ctx, const MsmPoints& pts, const fr_t* scalars_dev, size_t n, const MsmFiller& filler,
                        Exchange* ex) {
    host::G1AffineH out;
    msm_run_batch(ctx, pts, &scalars_dev, &n, 1, &out, filler, ex);
    return out;
}

// Up to MSM_BATCH_MAX independent MSMs against the same points, pipelined on the device.  add() starts
// one (its sort and bucket phases) as soon as its scalars are ready on ctx->stream, so the caller can
// go on preparing the next scalars; finish() runs the reductions and collects the results.
void msm_batch_add(Ctx* ctx, MsmBatch& b, const fr_t* scalars_dev, size_t n, bool hold_buckets) {
    PLONK_REQUIRE(b.count < MSM_BATCH_MAX, "msm: batch is full");
    PLONK_REQUIRE(n <= b.pts.level_stride, "msm: more scalars than points");
    MsmWorkspace* ws = msm_workspace(ctx);
    HIP_CHECK(hipEventRecord(ws->ready, ctx->stream));
    HIP_CHECK(hipStreamWaitEvent(ws->sort_stream, ws->ready, 0));
    b.scalars[b.count] = scalars_dev;
    b.n[b.count] = n;
    b.held[b.count] = hold_buckets;
    msm_enqueue(ctx, ws, b.count, b.pts, scalars_dev, n, hold_buckets ? PH_SORT : (PH_SORT | PH_BUCKETS), b.ex);
    ++b.count;
}

// Starts the bucket kernels of the MSMs added with hold_buckets, behind what ctx->stream holds now.
void msm_batch_release(Ctx* ctx, MsmBatch& b) {
    MsmWorkspace* ws = msm_workspace(ctx);
    bool any = false;
    for (int k = 0; k < b.count; ++k) any = any || b.held[k];
    if (!any) return;
    HIP_CHECK(hipEventRecord(ws->ready, ctx->stream));
    HIP_CHECK(hipStreamWaitEvent(ws->bucket_stream, ws->ready, 0));
    for (int k = 0; k < b.count; ++k) {
        if (!b.held[k]) continue;
        msm_enqueue(ctx, ws, k, b.pts, b.scalars[k], b.n[k], PH_BUCKETS, b.ex);
        b.held[k] = false;
    }
}

void msm_batch_finish(Ctx* ctx, MsmBatch& b, host::G1AffineH* out, const MsmFiller& filler) {
    MsmWorkspace* ws = msm_workspace(ctx);
    msm_batch_release(ctx, b);
    for (int k = 0; k < b.count; ++k) msm_enqueue(ctx, ws, k, b.pts, b.scalars[k], b.n[k], PH_TAIL, b.ex);
    if (filler) {
        // behind what ctx->stream held when the last MSM was added (not behind the reductions above)
        if (b.count == 0) HIP_CHECK(hipEventRecord(ws->ready, ctx->stream));
        HIP_CHECK(hipStreamWaitEvent(ws->bucket_stream, ws->ready, 0));
        filler(ws->bucket_stream);
        HIP_CHECK(hipEventRecord(ws->filled, ws->bucket_stream));
        HIP_CHECK(hipStreamWaitEvent(ctx->stream, ws->filled, 0));
    }
    for (int k = 0; k < b.count; ++k) out[k] = msm_collect(ws, k);
    b.count = 0;
}

void msm_run_batch(Ctx* ctx, const MsmPoints& pts, const fr_t* const* scalars_dev, const size_t* n, int count,
                   host::G1AffineH* out, const MsmFiller& filler, Exchange* ex) {
    static_assert(MSM_BATCH_MAX == MsmWorkspace::SLOTS, "one workspace slot per MSM of a batch");
    for (int k = 0; k < count; ++k) PLONK_REQUIRE(n[k] <= pts.level_stride, "msm: more scalars than points");
    MsmBatch b;
    b.pts = pts;
    b.ex = ex;
    for (int base = 0; base < count; base += MSM_BATCH_MAX) {
        const int m = std::min(MSM_BATCH_MAX, count - base);
        for (int k = 0; k < m; ++k) msm_batch_add(ctx, b, scalars_dev[base + k], n[base + k]);
        msm_batch_finish(ctx, b, out + base, base + m == count ? filler : MsmFiller());
    }
    if (count == 0 && filler) msm_batch_finish(ctx, b, out, filler);
}

// After a failed call: wait for whatever the MSM streams still hold, so that the next call does not
// rewrite a slot's buffers under a kernel of the abandoned batch.
void msm_drain(Ctx* ctx) {
    if (!ctx->msm_ws) return;
    MsmWorkspace* ws = static_cast<MsmWorkspace*>(ctx->msm_ws.get());
    if (ws->sort_stream) (void)hipStreamSynchronize(ws->sort_stream);
    if (ws->bucket_stream) (void)hipStreamSynchronize(ws->bucket_stream);
}

// Timing marks are recorded only while enabled (extra events on the MSM streams).
void msm_set_timing(Ctx* ctx, bool on) { msm_workspace(ctx)->timing = on; }

// Phase durations of the most recent MSM that ran on slot 0 with timing enabled, and the number of
// (point, sign) entries it accumulated.
void msm_last_timing(Ctx* ctx, MsmTiming* out) {
    MsmWorkspace* ws = msm_workspace(ctx);
    MsmSlot& s = ws->slot[0];
    PLONK_REQUIRE(!s.empty && s.c != 0 && s.timed, "msm timing: no timed MSM has run on this context");
    HIP_CHECK(hipEventSynchronize(s.t_ev[4]));
    HIP_CHECK(hipEventElapsedTime(&out->sort_ms, s.t_ev[0], s.t_ev[1]));
    HIP_CHECK(hipEventElapsedTime(&out->bucket_ms, s.t_ev[2], s.t_ev[3]));
    HIP_CHECK(hipEventElapsedTime(&out->big_bucket_ms, s.t_ev[3], s.t_ev[5]));
    HIP_CHECK(hipEventElapsedTime(&out->reduce_ms, s.t_ev[5], s.t_ev[4]));
    out->entries = *s.host_entries;
    out->window_bits = s.c;
    out->bucket_sets = s.W;
}

void g1_convert_to_xform(Ctx* ctx, const g1_affine_t* in, size_t n, g1x_affine_t* out) {
    if (n == 0) return;
    hipLaunchKernelGGL(g1_to_xform_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, in, (uint32_t)n, out);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// SRS: g1_powers resident on the device as (x, y) Montgomery pairs, (0, 0) for the identity.

static void g1_pack(const plonk_g1* in, size_t n, std::vector<g1_affine_t>& out) {
    out.resize(n);
    for (size_t i = 0; i < n; ++i) {
        if (in[i].infinity) {
            memset(&out[i], 0, sizeof(g1_affine_t));
        } else {
            memcpy(out[i].x.l, in[i].x, 48);
            memcpy(out[i].y.l, in[i].y, 48);
        }
    }
}

void g1_unpack(const g1_affine_t& in, plonk_g1* out) {
    memset(out, 0, sizeof(*out));
    bool zero = true;
    for (int k = 0; k < 12; ++k) zero = zero && in.x.l[k] == 0 && in.y.l[k] == 0;
    if (zero) {
        out->infinity = 1;
        return;
    }
    memcpy(out->x, in.x.l, 48);
    memcpy(out->y, in.y.l, 48);
}

void g1_from_host(const host::G1AffineH& a, plonk_g1* out) {
    memset(out, 0, sizeof(*out));
    if (a.inf) {
        out->infinity = 1;
        return;
    }
    memcpy(out->x, a.x.l, 48);
    memcpy(out->y, a.y.l, 48);
}

// c doublings of every point: level w of the window table from level w-1
__global__ __launch_bounds__(256) void g1_shift_level_kernel(const g1_affine_t* __restrict__ in, uint32_t n, uint32_t c,
                                                              g1_xyzz_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const g1_affine_t p = ld_affine(in + i);
    g1_xyzz_t acc = xyzz_double_affine(p);
    for (uint32_t k = 1; k < c; ++k) acc = xyzz_double(acc);
    st_xyzz(out + i, acc);
}

static void xyzz_to_affine(Ctx* ctx, const g1_xyzz_t* in, size_t n, fq_t* prefix, g1_affine_t* out);

uint32_t msm_table_bits(size_t n) {
    // 8, 13, 16 or 20: widths whose top window of a 255-bit scalar is still 7, 8, 15 and 15 bits wide, so
    // no window crowds its digits into a handful of buckets.  The small widths keep the reduction of
    // a small SRS shallow; ceil(256/c) is 13 for c = 20 and for c = 21, so a wider window would only
    // add buckets, and larger inputs put more points into each of the 2^19 buckets.
    uint32_t lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) ++lg;
    return lg <= 8 ? 8u : (lg <= 12 ? 13u : (lg <= 17 ? 16u : 20u));
}

// Builds srs->px as the window table [W][n]: row w holds 2^(c w) P_i in x-form.  With the table
// every window of a scalar feeds the same bucket set, so an MSM reduces 2^(c-1) buckets once
// instead of W times, and c can grow to log2(n).  Costs W x the SRS in HBM (1.75 GB at n = 2^20).
static void srs_build_points(Ctx* ctx, Srs* s, bool with_table) {
    const size_t n = s->n;
    s->table_c = 0;
    s->levels = 1;
    if (!with_table || n < MSM_TABLE_MIN_POINTS) {
        s->px.alloc(n);
        g1_convert_to_xform(ctx, s->points.p, n, s->px.p);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return;
    }
    const uint32_t c = msm_table_bits(n), W = (256 + c - 1) / c;
    PLONK_REQUIRE((size_t)W * n < ((size_t)1 << 31), "srs: window table too large");
    s->px.alloc((size_t)W * n);
    g1_convert_to_xform(ctx, s->points.p, n, s->px.p);
    DevBuf<g1_affine_t> cur(n), nxt(n);
    DevBuf<g1_xyzz_t> tmp(n);
    DevBuf<fq_t> prefix(n);
    HIP_CHECK(hipMemcpyAsync(cur.p, s->points.p, n * sizeof(g1_affine_t), hipMemcpyDeviceToDevice, ctx->stream));
    for (uint32_t w = 1; w < W; ++w) {
        hipLaunchKernelGGL(g1_shift_level_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, cur.p,
                           (uint32_t)n, c, tmp.p);
        xyzz_to_affine(ctx, tmp.p, n, prefix.p, nxt.p);
        g1_convert_to_xform(ctx, nxt.p, n, s->px.p + (size_t)w * n);
        std::swap(cur.p, nxt.p);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    s->table_c = c;
    s->levels = W;
}

// *flag |= 1 if some point (other than the (0, 0) image of the identity) is off y^2 = x^3 + 4
__global__ __launch_bounds__(256) void g1_on_curve_kernel(const g1_affine_t* __restrict__ pts, uint32_t n, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const g1_affine_t p = ld_affine(pts + i);
    if (affine_is_inf(p)) return;
    fq_t four = f_one<FqP>();
    four = fq_add(four, four);
    four = fq_add(four, four);
    const fq_t rhs = fq_add(fq_mul(fq_mul(p.x, p.x), p.x), four);
    if (!f_eq<FqP>(fq_mul(p.y, p.y), rhs)) atomicOr(flag, 1u);
}

Srs* srs_upload(Ctx* ctx, const plonk_g1* pts, size_t n, bool with_table) {
    PLONK_REQUIRE(n < ((size_t)1 << 31), "srs: too many points");
    for (size_t i = 0; i < n; ++i) {
        if (pts[i].infinity) continue;
        host::fq_h x, y;
        memcpy(x.l, pts[i].x, 48);
        memcpy(y.l, pts[i].y, 48);
        PLONK_REQUIRE(!host::Fq::geq_p(x) && !host::Fq::geq_p(y), "srs: coordinate not reduced");
    }
    std::unique_ptr<Srs> s(new Srs());
    s->device = ctx->device;
    s->n = n;
    std::vector<g1_affine_t> packed;
    g1_pack(pts, n, packed);
    s->points.alloc(n);
    if (n) {
        HIP_CHECK(hipMemcpy(s->points.p, packed.data(), n * sizeof(g1_affine_t), hipMemcpyHostToDevice));
        // every point must satisfy the curve equation (checked where the points now are)
        DevBuf<uint32_t> flag(1);
        HIP_CHECK(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(g1_on_curve_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, s->points.p, (uint32_t)n,
                           flag.p);
        HIP_CHECK(hipGetLastError());
        uint32_t bad = 0;
        HIP_CHECK(hipMemcpyAsync(&bad, flag.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        PLONK_REQUIRE(bad == 0, "srs: point not on the curve");
    }
    srs_build_points(ctx, s.get(), with_table);
    return s.release();
}

// ---- fixed-base generation of [tau^i]_1 ----
// table[w][d-1] = d * 2^(8w) * G, w < 32, d in 1..255 (affine)
static const std::vector<g1_affine_t>& fixed_base_table() {
    static std::vector<g1_affine_t> table;
    if (!table.empty()) return table;
    using namespace host;
    std::vector<g1_affine_t> t(32 * 255);
    G1XyzzH base = g1_from_affine(g1_generator());
    for (int w = 0; w < 32; ++w) {
        G1XyzzH acc = g1_identity();
        for (int d = 1; d <= 255; ++d) {
            acc = g1_add(acc, base);
            G1AffineH a = g1_to_affine(acc);
            memcpy(t[w * 255 + d - 1].x.l, a.x.l, 48);
            memcpy(t[w * 255 + d - 1].y.l, a.y.l, 48);
        }
        base = g1_add(acc, base);  // 256 * base
    }
    table.swap(t);
    return table;
}

__global__ __launch_bounds__(256) void srs_powers_kernel(fr_t tau, uint64_t first, uint32_t n,
                                                          const g1_affine_t* __restrict__ table,
                                                          g1_xyzz_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // tau^(first + i) by square-and-multiply
    fr_t r = f_one<FrP>(), b = tau;
    for (uint64_t e = first + i; e; e >>= 1) {
        if (e & 1u) r = fr_mul(r, b);
        b = fr_mul(b, b);
    }
    fr_t s = f_from_mont<FrP>(r);
    g1_xyzz_t acc = xyzz_identity();
    for (uint32_t w = 0; w < 32; ++w) {
        const uint32_t d = s.l[0] & 0xffu;
#pragma unroll
        for (int k = 0; k < 7; ++k) s.l[k] = (s.l[k] >> 8) | (s.l[k + 1] << 24);
        s.l[7] >>= 8;
        if (d) {
            const g1_affine_t p = ld_affine(table + w * 255 + d - 1);
            xyzz_madd(acc, p);
        }
    }
    st_xyzz(out + i, acc);
}

__device__ __noinline__ fq_t fq_inverse(const fq_t& a) {
    // a^(q-2): q-2 differs from q only in limb 0 (0xffffaaab - 2)
    fq_t r = f_one<FqP>();
#pragma unroll 1
    for (int limb = 11; limb >= 0; --limb) {
        uint32_t e = FqP::p(0);
        switch (limb) {
            case 11: e = FqP::p(11); break; case 10: e = FqP::p(10); break; case 9: e = FqP::p(9); break;
            case 8: e = FqP::p(8); break; case 7: e = FqP::p(7); break; case 6: e = FqP::p(6); break;
            case 5: e = FqP::p(5); break; case 4: e = FqP::p(4); break; case 3: e = FqP::p(3); break;
            case 2: e = FqP::p(2); break; case 1: e = FqP::p(1); break; default: e = FqP::p(0) - 2u; break;
        }
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            r = fq_mul(r, r);
            if ((e >> bit) & 1u) r = fq_mul(r, a);
        }
    }
    return r;
}

// XYZZ -> affine for chunks of CH consecutive points per lane (Montgomery batch inversion through
// a global scratch array of prefix products).
__global__ __launch_bounds__(256) void xyzz_to_affine_kernel(const g1_xyzz_t* __restrict__ in, uint32_t n, uint32_t ch,
                                                              fq_t* __restrict__ prefix, g1_affine_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo = t * ch;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + ch);
    fq_t run = f_one<FqP>();
    for (uint32_t i = lo; i < hi; ++i) {
        st(prefix + i, run);  // product of the non-identity entries before i
        const fq_t zz = ld(&in[i].zz);
        if (!f_is_zero<FqP>(zz)) run = fq_mul(run, fq_mul(zz, ld(&in[i].zzz)));
    }
    fq_t inv = fq_inverse(run);
    for (uint32_t i = hi; i-- > lo;) {
        const fq_t zz = ld(&in[i].zz);
        g1_affine_t a;
        if (f_is_zero<FqP>(zz)) {
            a.x = f_zero<FqP>();
            a.y = f_zero<FqP>();
        } else {
            const fq_t zzz = ld(&in[i].zzz);
            const fq_t tinv = fq_mul(inv, ld(prefix + i));  // 1 / (zz * zzz)
            inv = fq_mul(inv, fq_mul(zz, zzz));
            a.x = fq_mul(ld(&in[i].x), fq_mul(tinv, zzz));
            a.y = fq_mul(ld(&in[i].y), fq_mul(tinv, zz));
        }
        st(&out[i].x, a.x);
        st(&out[i].y, a.y);
    }
}

static void xyzz_to_affine(Ctx* ctx, const g1_xyzz_t* in, size_t n, fq_t* prefix, g1_affine_t* out) {
    const uint32_t ch = 16;
    const uint32_t lanes = (uint32_t)((n + ch - 1) / ch);
    hipLaunchKernelGGL(xyzz_to_affine_kernel, dim3((lanes + 255) / 256), dim3(256), 0, ctx->stream, in, (uint32_t)n, ch,
                       prefix, out);
}

Srs* srs_from_tau(Ctx* ctx, const host::fr_h& tau, size_t first, size_t n, bool with_table) {
    PLONK_REQUIRE(n < ((size_t)1 << 31), "srs: too many points");
    std::unique_ptr<Srs> s(new Srs());
    s->device = ctx->device;
    s->n = n;
    s->points.alloc(n);
    if (n == 0) return s.release();
    const std::vector<g1_affine_t>& table = fixed_base_table();
    DevBuf<g1_affine_t> table_dev(table.size());
    HIP_CHECK(hipMemcpy(table_dev.p, table.data(), table.size() * sizeof(g1_affine_t), hipMemcpyHostToDevice));
    DevBuf<g1_xyzz_t> tmp(n);
    DevBuf<fq_t> prefix(n);
    const uint32_t nblk = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(srs_powers_kernel, dim3(nblk), dim3(256), 0, ctx->stream, to_dev(tau), (uint64_t)first, (uint32_t)n,
                       table_dev.p, tmp.p);
    xyzz_to_affine(ctx, tmp.p, n, prefix.p, s->points.p);
    HIP_CHECK(hipGetLastError());
    srs_build_points(ctx, s.get(), with_table);
    return s.release();
}

}  // namespace plonk
