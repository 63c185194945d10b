// Number-theoretic transform over BLS12-381 Fr for gfx950.
//
// Replaces fft / ifft / coset_fft / coset_ifft of backend_new/src/fft.rs:23-157 (natural order
// in and out; inverse scaled by 1/n; coset variants scale coefficient i by k^i before / k^-i
// after).  The reference runs log2(N) radix-2 sweeps over the whole vector; here a transform of
// N = N_1 * ... * N_P points is P sweeps over HBM.  Sweep p loads tiles of N_p x C elements
// into LDS (C consecutive elements per row, so every global access is a C*32-byte run), runs
// log2(N_p) radix-2 stages out of LDS, applies the step twiddles and stores.  The first sweep
// reads `data` and writes the intermediate image, the last reads that image and writes natural
// order back to `data`.
//
// Index algebra: address = hi * (N_p * S_p) + d * S_p + lo with S_p = N_{p+1}...N_P; sweep p
// replaces digit d = i_p by k_p.  hi holds k_1..k_{p-1} (k_1 most significant), lo holds
// i_{p+1}..i_P.  With M_p = N_1...N_{p-1} the input index is sum_q i_q S_q, the output index
// sum_p k_p M_p, and w_N^(input * output) factors into the sweeps' own transforms (q = p) and, for
// every pair p < q, the step twiddle w_N^(k_p M_p * i_q S_q): after sweep p an element is multiplied
// by one table entry [i_q][k_p] for each later sweep q.  Coset factors g^(i_q S_q) / g^(-k_p M_p)
// depend on one digit only and ride on the tables of neighbouring sweeps.
#include <algorithm>

#include <mutex>

#include "common.hpp"
#include "fieldx.cuh"

namespace plonk {

static constexpr int NTT_MAX_PASSES = 4;
static constexpr int NTT_TILE_LOG = 11;      // elements per LDS tile (72 KiB of 9-limb elements)
static constexpr int NTT_SUBTW_LOG = 11;     // sub_tw holds w_{2^11}^j, j < 2^10
static constexpr int NTT_THREADS = 512;

// Arithmetic runs in the radix-2^29 lazy Montgomery form of fieldx.cuh.  The transform is linear, so
// the saturated Montgomery words of an input are only regrouped into 29-bit limbs (the value read
// that way is the element times 2^-5, and the outputs carry the same factor back); only the
// twiddle tables are true x-form values.  Elements in LDS and in the intermediate image stay
// below 2p.
struct NttPassArgs {
    const fr_t* in;
    fr_t* out;
    const frx_slot* pre;     // [2^R] multiplies input digit d (nullable)
    const frx_slot* post;    // [2^R] multiplies output digit k (nullable)
    const frx_slot* sub_tw;  // w_{2^11}^j
    // step twiddles of a non-last sweep p: one table per later sweep q, entry [i_q][k_p] =
    // w_N^(k_p M_p * i_q S_q) (times the coset factors that depend on i_q or k_p alone)
    const frx_slot* pair_tab[NTT_MAX_PASSES - 1];
    // the product of the tables from index merged_from on, entry [k_p][low merged_bits bits of the lower
    // index] -- consecutive outputs read consecutive entries: one product per element instead of one per
    // later sweep (nullable)
    const frx_slot* merged_tab;
    uint32_t merged_from, merged_bits;
    uint32_t pair_shift[NTT_MAX_PASSES - 1];   // log2 S_q: where i_q sits in the lower index
    uint32_t pair_mask[NTT_MAX_PASSES - 1];    // N_q - 1
    uint32_t n_pairs;
    uint32_t log_n, R, log_s, log_c, log_h;
    uint32_t n_prev;              // number of earlier sweeps
    uint32_t r_prev[NTT_MAX_PASSES];
    uint64_t batch_stride;
    uint64_t in_len;              // first sweep: elements from this index on are taken as zero and not read (0 = all are read)
    uint32_t is_last;
    uint32_t xcd_remap;           // narrow tiles: column-adjacent tiles are given to workgroups of one XCD (see the kernel)
};

__device__ __forceinline__ uint32_t brev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// k_1 + N_1 k_2 + ... from hi = k_1 * (N_2..N_{q}) + ... + k_q  (mixed-radix digit reversal)
__device__ __forceinline__ uint32_t digits_rev(uint32_t hi, uint32_t log_h, uint32_t n_prev, const uint32_t* r_prev) {
    uint32_t K = 0, out_shift = 0, rem = log_h;
    for (uint32_t q = 0; q < n_prev; ++q) {
        rem -= r_prev[q];
        uint32_t digit = (hi >> rem) & ((1u << r_prev[q]) - 1u);
        K |= digit << out_shift;
        out_shift += r_prev[q];
    }
    return K;
}

// an LDS tile: 9 limbs per element in three planes (16 + 16 + 4 bytes)
struct LdsTile {
    uint4* a;
    uint4* b;
    uint32_t* c;
    __device__ __forceinline__ frx get(uint32_t i) const {
        const uint4 x = a[i], y = b[i];
        frx r;
        r.l[0] = x.x; r.l[1] = x.y; r.l[2] = x.z; r.l[3] = x.w; r.l[4] = y.x; r.l[5] = y.y; r.l[6] = y.z; r.l[7] = y.w;
        r.l[8] = c[i];
        return r;
    }
    __device__ __forceinline__ void put(uint32_t i, const frx& v) const {
        a[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
        b[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
        c[i] = v.l[8];
    }
};

// TWG: the sweep's own twiddles are read from sub_tw in memory (L2) instead of an LDS copy (a compile-time choice: a
// pointer picked at run time between the two would make every twiddle read a generic-address load).
// The kernel must stay within 128 VGPRs (four waves per SIMD): two workgroups then share a CU, and one computes while
// the other loads or stores; with 131 a transform of 2^22 elements takes 8 % longer.
template <bool TWG>
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(NttPassArgs p) {
    const bool LAST = p.is_last != 0;
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const uint32_t R = p.R, log_c = p.log_c;
    const uint32_t Np = 1u << R, C = 1u << log_c;
    const uint32_t T = Np << log_c;
    // last sweep: rows are contiguous in memory -> image [c][d] with one element of padding per
    // row; other sweeps: image [d][c].
    const bool rowmajor = LAST;
    const uint32_t plane = rowmajor ? (C * (Np + 1)) : T;
    LdsTile tile_mem;
    tile_mem.a = lds;
    tile_mem.b = lds + plane;
    tile_mem.c = reinterpret_cast<uint32_t*>(lds + 2 * plane);
    // w_{Np}^e, e < Np/2, behind the data planes
    LdsTile tw_mem;
    {
        const uint32_t half = Np >> 1;
        uint4* base = lds + ((plane * 36 + 15) >> 4);
        tw_mem.a = base;
        tw_mem.b = base + half;
        tw_mem.c = reinterpret_cast<uint32_t*>(base + 2 * half);
    }
    const uint32_t tid = threadIdx.x;
    // Workgroup ids go round the 8 XCDs, each with an L2 of its own.  A tile of one or two columns touches 32 or 64
    // bytes of every 128-byte line it reads and writes; the tiles that share those lines are neighbours in tile
    // order, so with narrow tiles four neighbouring tiles go to four consecutive workgroups of ONE XCD (ids 8
    // apart), which run at about the same time: the line is then fetched, and written back, once per XCD instead
    // of once per tile.  (The grid is a multiple of 32 workgroups whenever tiles are narrow.)
    uint32_t tile = blockIdx.x;
    if (p.xcd_remap) {
        const uint32_t xcd = tile & 7u, slot = tile >> 3;
        tile = ((slot >> 2) << 5) + (xcd << 2) + (slot & 3u);
    }
    constexpr bool tw_global = TWG;
    if (!tw_global)
        for (uint32_t e = tid; e < (Np >> 1); e += NTT_THREADS) tw_mem.put(e, ld_tw(p.sub_tw + ((size_t)e << (NTT_SUBTW_LOG - R))));
    auto tw = [&](uint32_t e) -> frx { return tw_global ? ld_tw(p.sub_tw + ((size_t)e << (NTT_SUBTW_LOG - R))) : tw_mem.get(e); };
    const fr_t* in = p.in + (size_t)blockIdx.y * p.batch_stride;
    fr_t* out = p.out + (size_t)blockIdx.y * p.batch_stride;

    // ---- tile geometry ----
    uint32_t hi = 0, lo_base = 0, k1_base = 0, rest = 0, log_h_rest = 0;
    if (!LAST) {
        const uint32_t tiles_per_hi_log = p.log_s - log_c;
        hi = tile >> tiles_per_hi_log;
        lo_base = (tile & ((1u << tiles_per_hi_log) - 1u)) << log_c;
    } else if (p.n_prev) {
        const uint32_t r1 = p.r_prev[0];
        log_h_rest = p.log_h - r1;
        const uint32_t k1_tiles_log = r1 - log_c;
        k1_base = (tile & ((1u << k1_tiles_log) - 1u)) << log_c;
        rest = tile >> k1_tiles_log;
    }
    auto lds_index = [&](uint32_t d, uint32_t c) -> uint32_t { return rowmajor ? (c * (Np + 1) + d) : ((d << log_c) + c); };

    // ---- load: canonical input on the first sweep, below 2p afterwards ----
    for (uint32_t e = tid; e < T; e += NTT_THREADS) {
        uint32_t c, d;
        size_t addr;
        if (rowmajor) {
            d = e & (Np - 1);
            c = e >> R;
            const uint32_t hi_c = ((k1_base + c) << log_h_rest) + rest;
            addr = ((size_t)hi_c << R) + d;
        } else {
            c = e & (C - 1);
            d = e >> log_c;
            addr = ((size_t)hi << (R + p.log_s)) + ((size_t)d << p.log_s) + lo_base + c;
        }
        frx x = x_zero<FrX>();
        if (!p.in_len || addr < p.in_len) {   // a zero-padded coefficient vector: the padding costs neither a read nor a product
            x = x_repack_in<FrX>(ld(in + addr));
            if (p.pre) x = x_mul<FrX>(x, ld_tw(p.pre + d));
        }
        tile_mem.put(lds_index(d, c), x);
    }
    __syncthreads();

    // ---- log2(Np) decimation-in-time stages on natural-order input (the twiddle of a butterfly
    //      depends on its block, in bit-reversed order); the result lands bit-reversed in d.
    //      Two stages at a time in registers; the stage s = R-1 has twiddle 1 throughout.
    //      Only the multiplied operand of a butterfly is reduced (below 2p): an element grows by at
    //      most 3p per stage, at most 35p over the 11 stages a sweep can have. ----
    uint32_t s = R;
    if (R & 1u) {
        const uint32_t m = Np >> 1;
        for (uint32_t q = tid; q < (T >> 1); q += NTT_THREADS) {
            uint32_t c, j;
            if (rowmajor) {
                j = q & (m - 1);
                c = q >> (R - 1);
            } else {
                c = q & (C - 1);
                j = q >> log_c;
            }
            const uint32_t i0 = lds_index(j, c), i1 = lds_index(j + m, c);
            const frx u = tile_mem.get(i0), v = tile_mem.get(i1);
            tile_mem.put(i0, x_norm_u<FrX>(x_add_lazy<FrX>(u, v)));      // < 4p
            tile_mem.put(i1, x_norm_u<FrX>(x_sub_lazy<FrX, 3>(u, v)));   // < 5p
        }
        __syncthreads();
        s = R - 1;
    }
    for (; s >= 2; s -= 2) {
        const uint32_t h = 1u << (s - 2);
        const bool first = s == R;
        for (uint32_t q = tid; q < (T >> 2); q += NTT_THREADS) {
            uint32_t c, g;
            if (rowmajor) {
                g = q & ((Np >> 2) - 1);
                c = q >> (R - 2);
            } else {
                c = q & (C - 1);
                g = q >> log_c;
            }
            const uint32_t j = g & (h - 1), blk = g >> (s - 2);
            const uint32_t d0 = (blk << s) + j;
            const uint32_t i0 = lds_index(d0, c), i1 = lds_index(d0 + h, c), i2 = lds_index(d0 + 2 * h, c),
                           i3 = lds_index(d0 + 3 * h, c);
            const frx x0 = tile_mem.get(i0), x1 = tile_mem.get(i1), x2 = tile_mem.get(i2), x3 = tile_mem.get(i3);
            frx z0, z1, z2, z3;
            if (first) {
                // stages R-1 and R-2 of block 0: the only twiddle that is not 1 is w_4
                const frx y0 = x_add_lazy<FrX>(x0, x2), y2 = x_sub_lazy<FrX, 3>(x0, x2);
                const frx y1 = x_norm_u<FrX>(x_add_lazy<FrX>(x1, x3)), y3 = x_sub_lazy<FrX, 3>(x1, x3);
                const frx t3 = x_mul<FrX>(y3, tw(Np >> 2));
                z0 = x_add_lazy<FrX>(y0, y1);        // < 8p
                z1 = x_sub_lazy<FrX, 5>(y0, y1);     // < 9p
                z2 = x_add_lazy<FrX>(y2, t3);
                z3 = x_sub_lazy<FrX, 3>(y2, t3);
            } else {
                const uint32_t ea = brev(blk, R - s) << (s - 2);
                const frx ts = tw(2 * ea);
                const frx t2 = x_mul<FrX>(x2, ts), t3 = x_mul<FrX>(x3, ts);
                const frx y0 = x_add_lazy<FrX>(x0, t2), y2 = x_sub_lazy<FrX, 3>(x0, t2);
                const frx y1 = x_add_lazy<FrX>(x1, t3), y3 = x_sub_lazy<FrX, 3>(x1, t3);
                const frx u1 = x_mul<FrX>(y1, tw(ea)), u3 = x_mul<FrX>(y3, tw(ea + (Np >> 2)));
                z0 = x_add_lazy<FrX>(y0, u1);
                z1 = x_sub_lazy<FrX, 3>(y0, u1);
                z2 = x_add_lazy<FrX>(y2, u3);
                z3 = x_sub_lazy<FrX, 3>(y2, u3);     // < input + 6p
            }
            tile_mem.put(i0, x_norm_u<FrX>(z0));
            tile_mem.put(i1, x_norm_u<FrX>(z1));
            tile_mem.put(i2, x_norm_u<FrX>(z2));
            tile_mem.put(i3, x_norm_u<FrX>(z3));
        }
        __syncthreads();
    }

    // ---- store ----
    for (uint32_t e = tid; e < T; e += NTT_THREADS) {
        const uint32_t c = e & (C - 1);
        const uint32_t k = e >> log_c;
        frx x = tile_mem.get(lds_index(brev(k, R), c));
        if (p.post) x = x_mul<FrX>(x, ld_tw(p.post + k));
        size_t addr;
        if (LAST) {
            const uint32_t hi_c = ((k1_base + c) << log_h_rest) + rest;
            const uint32_t K = digits_rev(hi_c, p.log_h, p.n_prev, p.r_prev);
            addr = (size_t)K + ((size_t)k << p.log_h);
            // the transform's output is canonical: a product is below 2p; without one the element is a
            // lazy sum below 64p
            x = p.post ? x_canon2<FrX>(x) : frx_canon64(x);
        } else {
            // step twiddles: for every later sweep q the factor w_N^(k M_p * i_q S_q), i_q read off
            // the lower index (this is also what brings the element back below 2p)
            const uint32_t lo = lo_base + c;
            addr = ((size_t)hi << (R + p.log_s)) + ((size_t)k << p.log_s) + lo;
            const uint32_t separate = p.merged_tab ? p.merged_from : p.n_pairs;
#pragma unroll 1
            for (uint32_t j = 0; j < separate; ++j) {
                const uint32_t iq = (lo >> p.pair_shift[j]) & p.pair_mask[j];
                x = x_mul<FrX>(x, ld_tw(p.pair_tab[j] + (((size_t)iq << R) + k)));
            }
            if (p.merged_tab) x = x_mul<FrX>(x, ld_tw(p.merged_tab + (((size_t)k << p.merged_bits) + (lo & ((1u << p.merged_bits) - 1u)))));
        }
        st(out + addr, x_repack_out<FrX>(x));
    }
}

// ------------------------------------------------------------------------------------------
// Twiddle tables are generated on the device: entry [i][k] of a table is
// c0 * bi^i * bk^k * bik^(i k), stored as a true value of the 29-bit-limb form (48-byte slot).
// Field arithmetic is exact, so the entries are the same whichever way the powers are taken.
struct TableGen {
    fr_t c0, bi, bk, bik, k32;   // Montgomery residues; k32 = 32 (the radix change 2^256 -> 2^261)
    uint32_t log_np;             // k runs over [0, 2^log_np)
    uint32_t count;              // entries
};

__device__ __forceinline__ fr_t fr_pow_u64_dev(fr_t b, uint64_t e) {
    fr_t r = f_one<FrP>();
    while (e) {
        if (e & 1ull) r = fr_mul(r, b);
        b = fr_mul(b, b);
        e >>= 1;
    }
    return r;
}

__global__ __launch_bounds__(256) void ntt_table_kernel(frx_slot* __restrict__ out, TableGen g) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.count) return;
    const uint64_t i = idx >> g.log_np, k = idx & ((1u << g.log_np) - 1u);
    fr_t v = fr_mul(g.c0, fr_pow_u64_dev(g.bk, k));
    if (i) v = fr_mul(fr_mul(v, fr_pow_u64_dev(g.bi, i)), fr_pow_u64_dev(g.bik, i * k));
    const frx x = x_repack_in<FrX>(fr_mul(v, g.k32));
    uint4* q = reinterpret_cast<uint4*>(out + idx);
    q[0] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
    q[1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    q[2] = make_uint4(x.l[8], 0u, 0u, 0u);
}

// merged[k][lo] = prod_j t_j[i_j(lo)][k] over 2 or 3 step-twiddle tables of one sweep (t[2] nullable)
struct MergeArgs {
    const frx_slot* t[3];
    uint32_t shift[3], mask[3];
    uint32_t R, bits;   // k < 2^R, lo < 2^bits
};
__global__ __launch_bounds__(256) void ntt_merge_kernel(frx_slot* __restrict__ out, MergeArgs m) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;   // the grid covers 2^(R + bits) exactly
    const uint32_t k = idx >> m.bits, lo = idx & ((1u << m.bits) - 1u);
    frx x = x_mul<FrX>(ld_tw(m.t[0] + (((size_t)((lo >> m.shift[0]) & m.mask[0]) << m.R) + k)),
                       ld_tw(m.t[1] + (((size_t)((lo >> m.shift[1]) & m.mask[1]) << m.R) + k)));
    if (m.t[2]) x = x_mul<FrX>(x, ld_tw(m.t[2] + (((size_t)((lo >> m.shift[2]) & m.mask[2]) << m.R) + k)));
    x = x_canon2<FrX>(x);
    uint4* q = reinterpret_cast<uint4*>(out + idx);
    q[0] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
    q[1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    q[2] = make_uint4(x.l[8], 0u, 0u, 0u);
}

// a merged table has at most this many entries (48 bytes each)
static constexpr uint32_t NTT_MERGED_MAX_LOG = 24;

struct NttPlan {
    uint32_t log_n = 0, flags = 0, n_pass = 0;
    uint32_t R[NTT_MAX_PASSES] = {0, 0, 0, 0};
    const frx_slot* sub_tw = nullptr;
    const frx_slot* pair[NTT_MAX_PASSES][NTT_MAX_PASSES] = {};   // [p][q], p < q: step twiddles
    // sweep p with two or more later sweeps: the step twiddles of the sweeps from merged_q0[p] on in one table
    const frx_slot* merged[NTT_MAX_PASSES] = {};
    uint32_t merged_q0[NTT_MAX_PASSES] = {0, 0, 0, 0};
    const frx_slot *pre = nullptr, *post = nullptr;   // first sweep's input scale (forward coset), last sweep's output scale (inverse)
    DevBuf<frx_slot> arena;   // owns the tables, unless they were placed in memory the caller provided
    host::fr_h shift;
};

Ctx::~Ctx() {
    plans.clear();
    msm_ws.reset();
    prover_ws.reset();
    ntt_tmp.release();
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (q_ev0) (void)hipEventDestroy(q_ev0);
    if (q_ev1) (void)hipEventDestroy(q_ev1);
    if (stream) (void)hipStreamDestroy(stream);
}

static void plan_geometry(NttPlan* pl, uint32_t log_n, uint32_t flags) {
    pl->log_n = log_n;
    pl->flags = flags;
    // One sweep up to the tile size; two up to 2^21; from there sweeps of 2^8, the last ones shortened -- by
    // two stages at a time where possible: a sweep of an even number of stages runs them all in pairs, and
    // the first pair of a sweep costs one product per four elements instead of four (measured on MI355X,
    // coset transform: 2^22 as 8+8+6 0.56 ms, as 8+7+7 0.627 ms, as 11+11 -- single-column tiles, with the
    // neighbouring tiles kept on one XCD and the twiddles read from L2 -- 0.59 ms; 2^21 as 11+10 0.275 ms
    // against 0.284 ms in three sweeps).
    uint32_t P;
    if (log_n <= NTT_TILE_LOG) {
        P = 1;
        pl->R[0] = log_n;
    } else if (log_n <= 2 * NTT_TILE_LOG - 1) {
        P = 2;
        pl->R[0] = (log_n + 1) / 2;
        pl->R[1] = log_n / 2;
    } else {
        P = (log_n + 7) / 8;
        PLONK_REQUIRE(P <= NTT_MAX_PASSES, "ntt: log_n out of range");
        for (uint32_t p = 0; p < P; ++p) pl->R[p] = 8;
        uint32_t excess = 8 * P - log_n;
        for (uint32_t p = P; excess >= 2 && p-- > 1;) {
            pl->R[p] -= 2;
            excess -= 2;
        }
        for (uint32_t p = P; excess >= 2 && p-- > 1;) {   // a second round for the smallest sizes of a sweep count
            pl->R[p] -= 2;
            excess -= 2;
        }
        if (excess) pl->R[1] -= excess;   // 0 or 1 left
    }
    pl->n_pass = P;
}

// log2 S_q = log2 (N_{q+1} ... N_P)
static uint32_t plan_log_s(const NttPlan* pl, uint32_t q) {
    uint32_t ls = 0;
    for (uint32_t j = q + 1; j < pl->n_pass; ++j) ls += pl->R[j];
    return ls;
}
// Sweep p merges the step twiddles of the later sweeps q >= q0 into one table [k_p][i_q0 .. i_P]: all of
// them when that table is small enough, else all but the next sweep's; 0 = no merged table.
static uint32_t plan_merge_from(const NttPlan* pl, uint32_t p) {
    const uint32_t P = pl->n_pass;
    for (uint32_t q0 = p + 1; q0 + 1 < P; ++q0)
        if (pl->R[p] + plan_log_s(pl, q0 - 1) <= NTT_MERGED_MAX_LOG) return q0;
    return 0;
}

// 48-byte slots the tables of a plan take
static size_t plan_table_slots(const NttPlan* pl) {
    const bool inverse = pl->flags & NTT_INVERSE, coset = pl->flags & NTT_COSET;
    const uint32_t P = pl->n_pass;
    size_t slots = (size_t)1 << (NTT_SUBTW_LOG - 1);
    if (coset && !inverse) slots += (size_t)1 << pl->R[0];
    if (inverse) slots += (size_t)1 << pl->R[P - 1];
    for (uint32_t p = 0; p + 1 < P; ++p)
        for (uint32_t q = p + 1; q < P; ++q) slots += (size_t)1 << (pl->R[p] + pl->R[q]);
    for (uint32_t p = 0; p + 2 < P; ++p) {
        const uint32_t q0 = plan_merge_from(pl, p);
        if (q0) slots += (size_t)1 << (pl->R[p] + plan_log_s(pl, q0 - 1));
    }
    return slots;
}

// Fills the tables of `pl` (geometry set) by kernels on `stream`; they live in `mem` (room for
// plan_table_slots entries) or, when mem is null, in an arena the plan allocates.
static void plan_build_tables(NttPlan* pl, const host::fr_h& shift, frx_slot* mem, hipStream_t stream) {
    using host::Fr;
    const uint32_t log_n = pl->log_n, P = pl->n_pass;
    pl->shift = shift;
    const bool inverse = pl->flags & NTT_INVERSE, coset = pl->flags & NTT_COSET;
    if (!mem) {
        pl->arena.alloc(plan_table_slots(pl));
        mem = pl->arena.p;
    }
    frx_slot* cursor = mem;
    const host::fr_h one = Fr::one();
    const fr_t k32 = to_dev(Fr::from_u64(32));
    auto gen = [&](const host::fr_h& c0, const host::fr_h& bi, const host::fr_h& bk, const host::fr_h& bik, uint32_t log_np,
                   uint32_t log_ni) -> const frx_slot* {
        TableGen g;
        g.c0 = to_dev(c0); g.bi = to_dev(bi); g.bk = to_dev(bk); g.bik = to_dev(bik); g.k32 = k32;
        g.log_np = log_np;
        g.count = 1u << (log_np + log_ni);
        frx_slot* dst = cursor;
        cursor += g.count;
        hipLaunchKernelGGL(ntt_table_kernel, dim3((g.count + 255) / 256), dim3(256), 0, stream, dst, g);
        HIP_CHECK(hipGetLastError());
        return dst;
    };
    // twiddle tables of w = w_N (or its inverse)
    host::fr_h w = host::fr_root_of_unity(log_n);
    if (inverse) w = Fr::inv(w);
    {
        host::fr_h w11 = host::fr_root_of_unity(NTT_SUBTW_LOG);
        if (inverse) w11 = Fr::inv(w11);
        pl->sub_tw = gen(one, one, w11, one, NTT_SUBTW_LOG - 1, 0);
    }
    // index algebra: input index = sum_q i_q S_q, output index = sum_p k_p M_p
    uint32_t log_S[NTT_MAX_PASSES], log_M[NTT_MAX_PASSES];
    {
        uint32_t ls = log_n, lm = 0;
        for (uint32_t p = 0; p < P; ++p) {
            ls -= pl->R[p];
            log_S[p] = ls;
            log_M[p] = lm;
            lm += pl->R[p];
        }
    }
    const host::fr_h n_inv = Fr::inv(Fr::from_u64((uint64_t)1 << log_n));
    const host::fr_h g_inv = coset ? Fr::inv(shift) : one;
    // forward coset: coefficient i is scaled by g^i (fft.rs:129-133), i = sum_q i_q S_q.  The factor of
    // the first digit multiplies the input of the first sweep; that of digit q > 0 rides on the step
    // twiddles applied after sweep q-1.
    pl->pre = nullptr;
    if (coset && !inverse) pl->pre = gen(one, one, Fr::pow_u64(shift, (uint64_t)1 << log_S[0]), one, pl->R[0], 0);
    // inverse: output K is scaled by 1/n (fft.rs:52-56) and, for the coset form, g^-K (fft.rs:147-152),
    // K = sum_p k_p M_p.  The factor of the last digit (and 1/n) multiplies the output of the last
    // sweep; that of digit p < P-1 rides on the step twiddles applied after sweep p.
    pl->post = nullptr;
    if (inverse) pl->post = gen(n_inv, one, Fr::pow_u64(g_inv, (uint64_t)1 << log_M[P - 1]), one, pl->R[P - 1], 0);
    for (uint32_t p = 0; p < NTT_MAX_PASSES; ++p)
        for (uint32_t q = 0; q < NTT_MAX_PASSES; ++q) pl->pair[p][q] = nullptr;
    for (uint32_t p = 0; p + 1 < P; ++p) {
        for (uint32_t q = p + 1; q < P; ++q) {
            // entry [i][k] = step^(i k) gi^i gk^k; w^(M_p S_q) has order N / (M_p S_q) >= N_p N_q
            const host::fr_h step = Fr::pow_u64(w, (uint64_t)1 << (log_M[p] + log_S[q]));
            const bool neighbours = q == p + 1;
            const host::fr_h gi = (coset && !inverse && neighbours) ? Fr::pow_u64(shift, (uint64_t)1 << log_S[q]) : one;
            const host::fr_h gk = (coset && inverse && neighbours) ? Fr::pow_u64(g_inv, (uint64_t)1 << log_M[p]) : one;
            pl->pair[p][q] = gen(one, gi, gk, step, pl->R[p], pl->R[q]);
        }
    }
    for (uint32_t p = 0; p < NTT_MAX_PASSES; ++p) pl->merged[p] = nullptr;
    for (uint32_t p = 0; p + 2 < P; ++p) {
        const uint32_t q0 = plan_merge_from(pl, p);
        pl->merged_q0[p] = q0;
        if (!q0) continue;
        MergeArgs m;
        memset(&m, 0, sizeof(m));
        for (uint32_t q = q0; q < P; ++q) {
            m.t[q - q0] = pl->pair[p][q];
            m.shift[q - q0] = log_S[q];
            m.mask[q - q0] = (1u << pl->R[q]) - 1u;
        }
        m.R = pl->R[p];
        m.bits = log_S[q0 - 1];
        const uint32_t log_count = m.R + m.bits;
        frx_slot* dst = cursor;
        cursor += (size_t)1 << log_count;
        if (log_count >= 8) hipLaunchKernelGGL(ntt_merge_kernel, dim3(1u << (log_count - 8)), dim3(256), 0, stream, dst, m);
        else hipLaunchKernelGGL(ntt_merge_kernel, dim3(1), dim3(1u << log_count), 0, stream, dst, m);
        HIP_CHECK(hipGetLastError());
        pl->merged[p] = dst;
    }
}

static constexpr uint32_t PLAN_SHIFT_SLOTS = 4;   // coset generators cached side by side per (size, direction)
static uint64_t plan_key(uint32_t log_n, uint32_t flags, uint32_t slot) {
    return (uint64_t)log_n | ((uint64_t)(flags & 3u) << 8) | ((uint64_t)slot << 16);
}

static NttPlan* get_plan_shift(Ctx* ctx, uint32_t log_n, uint32_t flags, const host::fr_h& shift) {
    // plans are cached per (log_n, flags) and, for coset transforms, per generator: a few side by side
    // (the sharded prover uses one of its own beside 7), after that the first slot is rebuilt
    const uint32_t slots = (flags & NTT_COSET) ? PLAN_SHIFT_SLOTS : 1;
    uint32_t slot = 0;
    for (; slot < slots; ++slot) {
        auto it = ctx->plans.find(plan_key(log_n, flags, slot));
        if (it == ctx->plans.end()) break;
        if (!(flags & NTT_COSET) || it->second->shift == shift) return it->second.get();
    }
    if (slot == slots) slot = 0;
    HIP_CHECK(hipStreamSynchronize(ctx->stream));  // a replaced plan may still be in use
    std::shared_ptr<NttPlan> pl(new NttPlan());
    plan_geometry(pl.get(), log_n, flags);
    plan_build_tables(pl.get(), shift, nullptr, ctx->stream);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));  // other streams of the context may run the plan next
    ctx->plans[plan_key(log_n, flags, slot)] = pl;
    return pl.get();
}

NttPlan* ntt_get_plan(Ctx* ctx, uint32_t log_n, uint32_t flags) {
    return get_plan_shift(ctx, log_n, flags, host::Fr::from_u64(7));
}

// a tile and its twiddles take more than the 64 KiB of dynamic LDS a kernel may use by default (per device)
static void ntt_raise_lds_limit() {
    const void* variants[2] = {reinterpret_cast<const void*>(ntt_pass_kernel<false>), reinterpret_cast<const void*>(ntt_pass_kernel<true>)};
    for (const void* f : variants) HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024));
}

// `tmp`: the intermediate image of a multi-sweep transform (as large as the strided batch); unused
// when the plan has one sweep.
static void ntt_launch(Ctx* ctx, hipStream_t stream, const NttPlan* pl, fr_t* data, uint32_t batch, size_t batch_stride,
                       fr_t* tmp, size_t in_len = 0) {
    const uint32_t log_n = pl->log_n, P = pl->n_pass;
    if (log_n == 0) return;  // fft.rs:64-66: n == 1 is the identity (1/n = 1 and k^0 = 1 too)
    if (!ctx->ntt_lds_attr_set) {   // per device, and a context belongs to one device
        ntt_raise_lds_limit();
        ctx->ntt_lds_attr_set = true;
    }
    uint32_t log_h = 0;
    uint32_t log_s = log_n;
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t R = pl->R[p];
        log_s -= R;
        const bool last = p + 1 == P;
        NttPassArgs a;
        memset(&a, 0, sizeof(a));
        a.in = (p == 0) ? data : tmp;
        a.out = last ? data : tmp;
        a.pre = p == 0 ? pl->pre : nullptr;
        a.post = last ? pl->post : nullptr;
        a.sub_tw = pl->sub_tw;
        a.n_pairs = P - 1 - p;
        a.merged_tab = pl->merged[p];
        a.merged_from = pl->merged[p] ? pl->merged_q0[p] - (p + 1) : 0;
        a.merged_bits = pl->merged[p] ? plan_log_s(pl, pl->merged_q0[p] - 1) : 0;
        {
            uint32_t ls = log_s;   // log2 S_p; S_q for q > p follows by peeling R[q]
            for (uint32_t q = p + 1; q < P; ++q) {
                ls -= pl->R[q];
                a.pair_tab[q - p - 1] = pl->pair[p][q];
                a.pair_shift[q - p - 1] = ls;
                a.pair_mask[q - p - 1] = (1u << pl->R[q]) - 1u;
            }
        }
        a.log_n = log_n;
        a.R = R;
        a.log_s = log_s;
        a.log_h = log_h;
        a.n_prev = p;
        for (uint32_t q = 0; q < p; ++q) a.r_prev[q] = pl->R[q];
        a.batch_stride = batch_stride;
        a.in_len = (p == 0 && in_len < ((size_t)1 << log_n)) ? in_len : 0;
        uint32_t log_c = NTT_TILE_LOG > R ? NTT_TILE_LOG - R : 0;
        if (last) log_c = p ? std::min(log_c, pl->R[0]) : 0u;
        else log_c = std::min(log_c, log_s);
        a.log_c = log_c;
        const uint32_t tiles = 1u << (log_n - R - log_c);
        const size_t elems = last ? ((size_t)(1u << log_c) * ((1u << R) + 1)) : ((size_t)1 << (R + log_c));
        // an LDS copy of the sweep's twiddles (2^(R-1) entries of 36 bytes) beside a full tile leaves room for one
        // workgroup per CU from R = 9 on; those sweeps read their twiddles from the 48 KiB table in L2 instead
        const bool tw_global = R >= 9;
        a.xcd_remap = (log_c < 2 && tiles >= 32) ? 1u : 0u;
        const size_t lds_bytes = ((elems * 36 + 15) & ~(size_t)15) + (tw_global ? 0 : ((size_t)1 << R) / 2 * 36) + 16;
        const dim3 grid(tiles, batch), block(NTT_THREADS);
        a.is_last = last ? 1u : 0u;
        if (tw_global) hipLaunchKernelGGL(ntt_pass_kernel<true>, grid, block, lds_bytes, stream, a);
        else hipLaunchKernelGGL(ntt_pass_kernel<false>, grid, block, lds_bytes, stream, a);
        HIP_CHECK(hipGetLastError());
        log_h += R;
    }
}

// the context's own intermediate image, grown on demand
static fr_t* ctx_tmp(Ctx* ctx, const NttPlan* pl, uint32_t batch, size_t batch_stride) {
    if (pl->n_pass <= 1) return nullptr;
    ctx->ntt_tmp.ensure((size_t)(batch - 1) * batch_stride + ((size_t)1 << pl->log_n));
    return ctx->ntt_tmp.p;
}

void ntt_run(Ctx* ctx, fr_t* data_dev, uint32_t log_n, uint32_t flags, uint32_t batch, size_t batch_stride, size_t in_len) {
    ntt_run_on(ctx, ctx->stream, data_dev, log_n, flags, batch, batch_stride, in_len);
}

// The same on another stream of the context.  The sweeps share one scratch buffer per context: the
// caller orders this stream against every other transform of the context.
void ntt_run_on(Ctx* ctx, hipStream_t stream, fr_t* data_dev, uint32_t log_n, uint32_t flags, uint32_t batch,
                size_t batch_stride, size_t in_len) {
    NttPlan* pl = ntt_get_plan(ctx, log_n, flags);
    ntt_launch(ctx, stream, pl, data_dev, batch, batch_stride, ctx_tmp(ctx, pl, batch, batch_stride), in_len);
}

void ntt_run_shift(Ctx* ctx, fr_t* data_dev, uint32_t log_n, uint32_t flags, const host::fr_h& shift, uint32_t batch,
                   size_t batch_stride) {
    ntt_run_shift_on(ctx, ctx->stream, data_dev, log_n, flags, shift, batch, batch_stride);
}

void ntt_run_shift_on(Ctx* ctx, hipStream_t stream, fr_t* data_dev, uint32_t log_n, uint32_t flags, const host::fr_h& shift,
                      uint32_t batch, size_t batch_stride) {
    NttPlan* pl = get_plan_shift(ctx, log_n, flags, shift);
    ntt_launch(ctx, stream, pl, data_dev, batch, batch_stride, ctx_tmp(ctx, pl, batch, batch_stride));
}

// ------------------------------------------------------------------------------------------
// A transform that allocates nothing, for callers that cannot call hipMalloc (HIP graph capture,
// serving loops, harnesses that forbid it).  Its intermediate image and its twiddle tables live in a
// static device array; every GPU the library runs on holds its own copy at its own address.
static constexpr uint32_t NOALLOC_MAX_LOG_N = 18;
static constexpr size_t NOALLOC_IMAGE_BYTES = ((size_t)1 << NOALLOC_MAX_LOG_N) * 48;   // intermediate image
static constexpr size_t NOALLOC_TABLE_BYTES = (((size_t)1 << NOALLOC_MAX_LOG_N) + 8192) * sizeof(frx_slot);
static constexpr size_t NOALLOC_SCRATCH_BYTES = NOALLOC_IMAGE_BYTES + NOALLOC_TABLE_BYTES;   // 24.4 MiB
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

static constexpr int NOALLOC_MAX_DEVICES = 64;
struct NoAllocState {
    unsigned char* ptr = nullptr;   // this device's address of `scratch`
    bool plan_valid = false;
    NttPlan plan;                   // tables inside the scratch array (arena stays empty)
    // Calls on one GPU share the array whatever context (stream) or host thread they come from: the mutex
    // serialises the host side, and every call's stream first waits for the event the previous call recorded
    // behind its last kernel, so a call never reads tables that are still being written on another stream nor
    // overwrites an image another stream is still transforming.
    std::mutex mu;
    hipEvent_t last = nullptr;
    bool last_recorded = false;
};
static NoAllocState g_noalloc[NOALLOC_MAX_DEVICES];
static std::mutex g_noalloc_init_mu;

// Set-up, once per device with that device current (plonk_ctx_create calls it): the array's name is
// a host-side stand-in; only the address looked up here may be handed to kernels.
void ntt_noalloc_init(int device) {
    PLONK_REQUIRE(device >= 0 && device < NOALLOC_MAX_DEVICES, "ntt: device index out of range");
    std::lock_guard<std::mutex> lock(g_noalloc_init_mu);
    NoAllocState& st = g_noalloc[device];
    if (st.ptr) return;
    // the sweeps' dynamic-LDS limit is raised here, per device, so that the entry point itself makes no
    // attribute call (a caller may have its stream under graph capture by then)
    ntt_raise_lds_limit();
    HIP_CHECK(hipEventCreateWithFlags(&st.last, hipEventDisableTiming));
    void* p = nullptr;
    HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    st.ptr = static_cast<unsigned char*>(p);
}

uint32_t ntt_noalloc_max_log_n() { return NOALLOC_MAX_LOG_N; }

// passes over HBM a transform of 2^log_n elements takes
uint32_t ntt_sweeps(uint32_t log_n) {
    NttPlan pl;
    plan_geometry(&pl, log_n, 0);
    return pl.n_pass;
}

// One vector of 2^log_n elements, in place, on ctx->stream.  The static workspace keeps the tables
// of the last (log_n, flags, shift) it was used with; another combination regenerates them by a
// few kernels on the same stream.  Nothing in the workspace has to start at zero: the tables and the
// intermediate image are written in full before they are read.
void ntt_run_noalloc(Ctx* ctx, fr_t* data_dev, uint32_t log_n, uint32_t flags, const host::fr_h& shift) {
    PLONK_REQUIRE(log_n <= NOALLOC_MAX_LOG_N, "ntt (no-alloc): size above the 2^18 elements the static workspace holds");
    NoAllocState& st = g_noalloc[ctx->device];
    PLONK_REQUIRE(st.ptr != nullptr, "ntt (no-alloc): static workspace of this device was not set up");
    std::lock_guard<std::mutex> lock(st.mu);
    if (st.last_recorded) HIP_CHECK(hipStreamWaitEvent(ctx->stream, st.last, 0));   // behind the previous call, whatever its stream
    const bool same = st.plan_valid && st.plan.log_n == log_n && st.plan.flags == flags &&
                      (!(flags & NTT_COSET) || st.plan.shift == shift);
    if (!same) {
        st.plan_valid = false;
        plan_geometry(&st.plan, log_n, flags);
        PLONK_REQUIRE(plan_table_slots(&st.plan) * sizeof(frx_slot) <= NOALLOC_TABLE_BYTES &&
                          ((size_t)sizeof(fr_t) << log_n) <= NOALLOC_IMAGE_BYTES,
                      "ntt (no-alloc): transform does not fit the static workspace");
        plan_build_tables(&st.plan, shift, reinterpret_cast<frx_slot*>(st.ptr + NOALLOC_IMAGE_BYTES), ctx->stream);
        st.plan_valid = true;
    }
    ntt_launch(ctx, ctx->stream, &st.plan, data_dev, 1, 0, reinterpret_cast<fr_t*>(st.ptr));
    HIP_CHECK(hipEventRecord(st.last, ctx->stream));
    st.last_recorded = true;
}

}  // namespace plonk
