/* CPU restatement in plain C of the reference's hot loops.  TEST INFRASTRUCTURE ONLY: the
 * checker for sizes where the Python oracle is too slow, and the timed CPU baseline.  Nothing
 * in the product links or loads this file.
 *
 * Follows, loop for loop:
 *   ref_fft_in_place  backend_new/src/fft.rs:62-102   (bit reversal, then log n radix-2 sweeps,
 *                                                      twiddle by running product, no table)
 *   ref_fft / ref_ifft / ref_coset_fft / ref_coset_ifft   fft.rs:23-59, 121-157
 *   ref_quotient      backend_new/src/plonk/prover.rs:400-457 (one pow and one inverse per point)
 *   ref_msm           backend_new/src/curve.rs:169-173: scalars to canonical form, then the
 *                     bucket method of ark-ec 0.4.2 `msm_bigint` (not in the reference tree):
 *                     unsigned c-bit windows, c = 3 below 32 points else ln(n)+2, Jacobian
 *                     accumulators with mixed additions, single thread.
 * Field elements are 4 x u64 (Fr) / 6 x u64 (Fq) little-endian Montgomery limbs, the in-memory
 * form of ark-ff that the reference's Fr / G1Affine wrap.  Pinned against the Python oracle
 * (oracle/*.py) by tests/test_oracle.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fr;
typedef struct { uint64_t l[6]; } fq;

static const uint64_t FR_P[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const uint64_t FR_INV = 0xfffffffeffffffffull;
static const uint64_t FR_ONE[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};
static const uint64_t FR_R2[4] = {0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full, 0x0748d9d99f59ff11ull};
static const uint64_t FQ_P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const uint64_t FQ_INV = 0x89f3fffcfffcfffdull;
static const uint64_t FQ_ONE[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                                   0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};

#define DEF_FIELD(T, N, P, INV, pre)                                                              \
    static inline int pre##_geq_p(const T* a) {                                                   \
        for (int i = N - 1; i >= 0; --i) {                                                        \
            if (a->l[i] > P[i]) return 1;                                                         \
            if (a->l[i] < P[i]) return 0;                                                         \
        }                                                                                         \
        return 1;                                                                                 \
    }                                                                                             \
    static inline void pre##_sub_p(T* a) {                                                        \
        u128 b = 0;                                                                               \
        for (int i = 0; i < N; ++i) {                                                             \
            u128 d = (u128)a->l[i] - P[i] - b;                                                    \
            a->l[i] = (uint64_t)d;                                                                \
            b = (d >> 64) & 1;                                                                    \
        }                                                                                         \
    }                                                                                             \
    static inline void pre##_add(T* r, const T* a, const T* b) {                                  \
        u128 c = 0;                                                                               \
        for (int i = 0; i < N; ++i) {                                                             \
            c += (u128)a->l[i] + b->l[i];                                                         \
            r->l[i] = (uint64_t)c;                                                                \
            c >>= 64;                                                                             \
        }                                                                                         \
        if (pre##_geq_p(r)) pre##_sub_p(r);                                                       \
    }                                                                                             \
    static inline void pre##_sub(T* r, const T* a, const T* b) {                                  \
        u128 bw = 0;                                                                              \
        for (int i = 0; i < N; ++i) {                                                             \
            u128 d = (u128)a->l[i] - b->l[i] - bw;                                                \
            r->l[i] = (uint64_t)d;                                                                \
            bw = (d >> 64) & 1;                                                                   \
        }                                                                                         \
        if (bw) {                                                                                 \
            u128 c = 0;                                                                           \
            for (int i = 0; i < N; ++i) {                                                         \
                c += (u128)r->l[i] + P[i];                                                        \
                r->l[i] = (uint64_t)c;                                                            \
                c >>= 64;                                                                         \
            }                                                                                     \
        }                                                                                         \
    }                                                                                             \
    static inline void pre##_mul(T* r, const T* a, const T* b) {                                  \
        uint64_t t[N + 2];                                                                        \
        memset(t, 0, sizeof(t));                                                                  \
        for (int i = 0; i < N; ++i) {                                                             \
            u128 c = 0;                                                                           \
            for (int j = 0; j < N; ++j) {                                                         \
                c += (u128)a->l[j] * b->l[i] + t[j];                                              \
                t[j] = (uint64_t)c;                                                               \
                c >>= 64;                                                                         \
            }                                                                                     \
            c += t[N];                                                                            \
            t[N] = (uint64_t)c;                                                                   \
            t[N + 1] = (uint64_t)(c >> 64);                                                       \
            uint64_t m = t[0] * INV;                                                              \
            c = (u128)m * P[0] + t[0];                                                            \
            c >>= 64;                                                                             \
            for (int j = 1; j < N; ++j) {                                                         \
                c += (u128)m * P[j] + t[j];                                                       \
                t[j - 1] = (uint64_t)c;                                                           \
                c >>= 64;                                                                         \
            }                                                                                     \
            c += t[N];                                                                            \
            t[N - 1] = (uint64_t)c;                                                               \
            t[N] = t[N + 1] + (uint64_t)(c >> 64);                                                \
        }                                                                                         \
        T o;                                                                                      \
        for (int i = 0; i < N; ++i) o.l[i] = t[i];                                                \
        if (t[N] || pre##_geq_p(&o)) pre##_sub_p(&o);                                             \
        *r = o;                                                                                   \
    }                                                                                             \
    static inline int pre##_is_zero(const T* a) {                                                 \
        uint64_t o = 0;                                                                           \
        for (int i = 0; i < N; ++i) o |= a->l[i];                                                 \
        return o == 0;                                                                            \
    }

DEF_FIELD(fr, 4, FR_P, FR_INV, fr)
DEF_FIELD(fq, 6, FQ_P, FQ_INV, fq)

static void fr_one(fr* r) { memcpy(r->l, FR_ONE, 32); }
static void fr_from_u64(fr* r, uint64_t v) {
    fr a = {{v, 0, 0, 0}}, r2;
    memcpy(r2.l, FR_R2, 32);
    fr_mul(r, &a, &r2);
}
/* field.rs:65 -- pow with a u64 exponent */
static void fr_pow(fr* r, const fr* a, uint64_t e) {
    fr acc, b = *a;
    fr_one(&acc);
    while (e) {
        if (e & 1) fr_mul(&acc, &acc, &b);
        fr_mul(&b, &b, &b);
        e >>= 1;
    }
    *r = acc;
}
/* field.rs:60 -- inverse (a^(r-2)) */
static void fr_inv(fr* r, const fr* a) {
    uint64_t e[4] = {FR_P[0] - 2, FR_P[1], FR_P[2], FR_P[3]};
    fr acc;
    fr_one(&acc);
    for (int i = 255; i >= 0; --i) {
        fr_mul(&acc, &acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fr_mul(&acc, &acc, a);
    }
    *r = acc;
}
/* field.rs:167-182 */
static void fr_root_of_unity(fr* r, unsigned log_n) {
    static const uint64_t T[4] = {0xfffe5bfeffffffffull, 0x09a1d80553bda402ull, 0x299d7d483339d808ull, 0x0000000073eda753ull};
    fr seven, acc;
    fr_from_u64(&seven, 7);
    fr_one(&acc);
    for (int i = 255; i >= 0; --i) {
        fr_mul(&acc, &acc, &acc);
        if ((T[i / 64] >> (i % 64)) & 1) fr_mul(&acc, &acc, &seven);
    }
    for (unsigned i = log_n; i < 32; ++i) fr_mul(&acc, &acc, &acc);
    *r = acc;
}

/* fft.rs:62-102 */
static void ref_fft_in_place(fr* v, size_t n, const fr* omega) {
    if (n == 1) return;
    unsigned log_n = 0;
    while (((size_t)1 << log_n) < n) ++log_n;
    for (size_t i = 0; i < n; ++i) {
        size_t j = 0, x = i;
        for (unsigned b = 0; b < log_n; ++b) { j = (j << 1) | (x & 1); x >>= 1; }
        if (i < j) { fr t = v[i]; v[i] = v[j]; v[j] = t; }
    }
    for (size_t m = 1; m < n; m *= 2) {
        fr omega_m;
        fr_pow(&omega_m, omega, n / (2 * m));
        for (size_t k = 0; k < n; k += 2 * m) {
            fr w;
            fr_one(&w);
            for (size_t j = 0; j < m; ++j) {
                fr t, u = v[k + j];
                fr_mul(&t, &w, &v[k + j + m]);
                fr_add(&v[k + j], &u, &t);
                fr_sub(&v[k + j + m], &u, &t);
                fr_mul(&w, &w, &omega_m);
            }
        }
    }
}

/* flags: 1 = inverse, 2 = coset (shift 7 unless `shift` given).  In place over n = 2^log_n. */
void ref_ntt(uint64_t* data, unsigned log_n, unsigned flags, const uint64_t* shift) {
    fr* v = (fr*)data;
    size_t n = (size_t)1 << log_n;
    fr omega, g;
    fr_root_of_unity(&omega, log_n);
    if (shift) memcpy(g.l, shift, 32); else fr_from_u64(&g, 7);
    if (!(flags & 1)) {
        if (flags & 2) { /* fft.rs:129-133 */
            fr kp;
            fr_one(&kp);
            for (size_t i = 0; i < n; ++i) { fr_mul(&v[i], &v[i], &kp); fr_mul(&kp, &kp, &g); }
        }
        ref_fft_in_place(v, n, &omega);
    } else {
        fr oi, ni, nn; /* fft.rs:49-56 */
        fr_inv(&oi, &omega);
        ref_fft_in_place(v, n, &oi);
        fr_from_u64(&nn, (uint64_t)n);
        fr_inv(&ni, &nn);
        for (size_t i = 0; i < n; ++i) fr_mul(&v[i], &v[i], &ni);
        if (flags & 2) { /* fft.rs:147-152 */
            fr gi, kp;
            fr_inv(&gi, &g);
            fr_one(&kp);
            for (size_t i = 0; i < n; ++i) { fr_mul(&v[i], &v[i], &kp); fr_mul(&kp, &kp, &gi); }
        }
    }
}

/* prover.rs:400-457.  ext: 15 arrays of n4 = 4n coset evaluations in the order
 * a b c z z_omega q_m q_l q_r q_o q_c s1 s2 s3 l1 pi; points [first, first+count) only. */
void ref_quotient(const uint64_t* ext, uint64_t n, const uint64_t* alpha_, const uint64_t* beta_, const uint64_t* gamma_,
                  const uint64_t* k1_, const uint64_t* k2_, uint64_t first, uint64_t count, uint64_t* t_out) {
    size_t n4 = 4 * n;
    const fr* E = (const fr*)ext;
    const fr *A = E, *B = E + n4, *C = E + 2 * n4, *Z = E + 3 * n4, *ZW = E + 4 * n4, *QM = E + 5 * n4, *QL = E + 6 * n4,
             *QR = E + 7 * n4, *QO = E + 8 * n4, *QC = E + 9 * n4, *S1 = E + 10 * n4, *S2 = E + 11 * n4, *S3 = E + 12 * n4,
             *L1 = E + 13 * n4, *PI = E + 14 * n4;
    fr alpha, beta, gamma, k1, k2, one, x, w4, g;
    memcpy(alpha.l, alpha_, 32); memcpy(beta.l, beta_, 32); memcpy(gamma.l, gamma_, 32);
    memcpy(k1.l, k1_, 32); memcpy(k2.l, k2_, 32);
    fr_one(&one);
    unsigned log4 = 0;
    while (((size_t)1 << log4) < n4) ++log4;
    fr_root_of_unity(&w4, log4);
    fr_from_u64(&g, 7);
    fr_pow(&x, &w4, first);
    fr_mul(&x, &x, &g);
    fr* T = (fr*)t_out;
    for (size_t i = first; i < first + count; ++i) {
        fr zh, gate, t, u, num, den, perm, l1c, numer, bx;
        fr_pow(&zh, &x, n); /* :403 */
        fr_sub(&zh, &zh, &one);
        fr_mul(&gate, &QM[i], &A[i]); fr_mul(&gate, &gate, &B[i]);
        fr_mul(&t, &QL[i], &A[i]); fr_add(&gate, &gate, &t);
        fr_mul(&t, &QR[i], &B[i]); fr_add(&gate, &gate, &t);
        fr_mul(&t, &QO[i], &C[i]); fr_add(&gate, &gate, &t);
        fr_add(&gate, &gate, &QC[i]); fr_add(&gate, &gate, &PI[i]);
        fr_mul(&bx, &beta, &x);
        fr_add(&t, &A[i], &bx); fr_add(&t, &t, &gamma); fr_mul(&num, &Z[i], &t);
        fr_mul(&u, &beta, &k1); fr_mul(&u, &u, &x); fr_add(&t, &B[i], &u); fr_add(&t, &t, &gamma); fr_mul(&num, &num, &t);
        fr_mul(&u, &beta, &k2); fr_mul(&u, &u, &x); fr_add(&t, &C[i], &u); fr_add(&t, &t, &gamma); fr_mul(&num, &num, &t);
        fr_mul(&u, &beta, &S1[i]); fr_add(&t, &A[i], &u); fr_add(&t, &t, &gamma); fr_mul(&den, &ZW[i], &t);
        fr_mul(&u, &beta, &S2[i]); fr_add(&t, &B[i], &u); fr_add(&t, &t, &gamma); fr_mul(&den, &den, &t);
        fr_mul(&u, &beta, &S3[i]); fr_add(&t, &C[i], &u); fr_add(&t, &t, &gamma); fr_mul(&den, &den, &t);
        fr_sub(&perm, &num, &den);
        fr_sub(&t, &Z[i], &one); fr_mul(&l1c, &L1[i], &t);
        fr_mul(&t, &alpha, &perm); fr_add(&numer, &gate, &t);
        fr_mul(&t, &alpha, &alpha); fr_mul(&t, &t, &l1c); fr_add(&numer, &numer, &t);
        fr_inv(&t, &zh); /* :447 */
        fr_mul(&T[i - first], &numer, &t);
        fr_mul(&x, &x, &w4);
    }
}

/* ---- G1 (Jacobian), y^2 = x^3 + 4 ---- */
typedef struct { fq x, y, z; } jac;
typedef struct { fq x, y; } aff; /* (0,0) marks the identity */

static void jac_set_id(jac* p) { memset(p, 0, sizeof(*p)); memcpy(p->x.l, FQ_ONE, 48); memcpy(p->y.l, FQ_ONE, 48); }
static int aff_is_id(const aff* a) { return fq_is_zero(&a->x) && fq_is_zero(&a->y); }

static void jac_double(jac* r, const jac* p) {
    if (fq_is_zero(&p->z) || fq_is_zero(&p->y)) { jac_set_id(r); return; }
    fq A, B, C, D, E, F, t, X3, Y3, Z3;
    fq_mul(&A, &p->x, &p->x); fq_mul(&B, &p->y, &p->y); fq_mul(&C, &B, &B);
    fq_add(&t, &p->x, &B); fq_mul(&t, &t, &t); fq_sub(&t, &t, &A); fq_sub(&t, &t, &C); fq_add(&D, &t, &t);
    fq_add(&E, &A, &A); fq_add(&E, &E, &A); fq_mul(&F, &E, &E);
    fq_sub(&X3, &F, &D); fq_sub(&X3, &X3, &D);
    fq_sub(&t, &D, &X3); fq_mul(&Y3, &E, &t);
    fq_add(&t, &C, &C); fq_add(&t, &t, &t); fq_add(&t, &t, &t); fq_sub(&Y3, &Y3, &t);
    fq_mul(&Z3, &p->y, &p->z); fq_add(&Z3, &Z3, &Z3);
    r->x = X3; r->y = Y3; r->z = Z3;
}

static void jac_add(jac* r, const jac* p, const jac* q) {
    if (fq_is_zero(&p->z)) { *r = *q; return; }
    if (fq_is_zero(&q->z)) { *r = *p; return; }
    fq Z1Z1, Z2Z2, U1, U2, S1, S2, H, R, HH, HHH, V, t, X3, Y3, Z3;
    fq_mul(&Z1Z1, &p->z, &p->z); fq_mul(&Z2Z2, &q->z, &q->z);
    fq_mul(&U1, &p->x, &Z2Z2); fq_mul(&U2, &q->x, &Z1Z1);
    fq_mul(&S1, &p->y, &q->z); fq_mul(&S1, &S1, &Z2Z2);
    fq_mul(&S2, &q->y, &p->z); fq_mul(&S2, &S2, &Z1Z1);
    fq_sub(&H, &U2, &U1); fq_sub(&R, &S2, &S1);
    if (fq_is_zero(&H)) {
        if (fq_is_zero(&R)) jac_double(r, p); else jac_set_id(r);
        return;
    }
    fq_mul(&HH, &H, &H); fq_mul(&HHH, &H, &HH); fq_mul(&V, &U1, &HH);
    fq_mul(&X3, &R, &R); fq_sub(&X3, &X3, &HHH); fq_sub(&X3, &X3, &V); fq_sub(&X3, &X3, &V);
    fq_sub(&t, &V, &X3); fq_mul(&Y3, &R, &t); fq_mul(&t, &S1, &HHH); fq_sub(&Y3, &Y3, &t);
    fq_mul(&Z3, &p->z, &q->z); fq_mul(&Z3, &Z3, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}

static void jac_add_mixed(jac* r, const jac* p, const aff* q) {
    if (aff_is_id(q)) { *r = *p; return; }
    jac qq;
    qq.x = q->x; qq.y = q->y; memcpy(qq.z.l, FQ_ONE, 48);
    if (fq_is_zero(&p->z)) { *r = qq; return; }
    fq Z1Z1, U2, S2, H, R, HH, HHH, V, t, X3, Y3, Z3;
    fq_mul(&Z1Z1, &p->z, &p->z);
    fq_mul(&U2, &q->x, &Z1Z1);
    fq_mul(&S2, &q->y, &p->z); fq_mul(&S2, &S2, &Z1Z1);
    fq_sub(&H, &U2, &p->x); fq_sub(&R, &S2, &p->y);
    if (fq_is_zero(&H)) {
        if (fq_is_zero(&R)) jac_double(r, p); else jac_set_id(r);
        return;
    }
    fq_mul(&HH, &H, &H); fq_mul(&HHH, &H, &HH); fq_mul(&V, &p->x, &HH);
    fq_mul(&X3, &R, &R); fq_sub(&X3, &X3, &HHH); fq_sub(&X3, &X3, &V); fq_sub(&X3, &X3, &V);
    fq_sub(&t, &V, &X3); fq_mul(&Y3, &R, &t); fq_mul(&t, &p->y, &HHH); fq_sub(&Y3, &Y3, &t);
    fq_mul(&Z3, &p->z, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}

/* curve.rs:169-173.  points: n x (x,y) Montgomery Fq, (0,0) = identity; scalars: n Montgomery Fr.
 * out: Jacobian X,Y,Z Montgomery (18 u64). */
void ref_msm(const uint64_t* points, const uint64_t* scalars, uint64_t n, uint64_t* out) {
    const aff* P = (const aff*)points;
    jac acc;
    jac_set_id(&acc);
    if (n == 0) { memcpy(out, &acc, sizeof(acc)); return; }
    /* into_bigint(): Montgomery -> canonical */
    fr* s = (fr*)malloc(n * sizeof(fr));
    fr one_raw = {{1, 0, 0, 0}};
    for (size_t i = 0; i < n; ++i) fr_mul(&s[i], (const fr*)scalars + i, &one_raw);
    unsigned c = n < 32 ? 3 : (unsigned)(log((double)n)) + 2;
    size_t nb = ((size_t)1 << c) - 1;
    jac* buckets = (jac*)malloc(nb * sizeof(jac));
    unsigned nwin = (255 + c - 1) / c;
    jac* wsum = (jac*)malloc(nwin * sizeof(jac));
    for (unsigned w = 0; w < nwin; ++w) {
        unsigned w0 = w * c;
        for (size_t b = 0; b < nb; ++b) jac_set_id(&buckets[b]);
        for (size_t i = 0; i < n; ++i) {
            unsigned limb = w0 / 64, off = w0 % 64;
            uint64_t d = s[i].l[limb] >> off;
            if (off + c > 64 && limb + 1 < 4) d |= s[i].l[limb + 1] << (64 - off);
            d &= ((uint64_t)1 << c) - 1;
            if (d) jac_add_mixed(&buckets[d - 1], &buckets[d - 1], &P[i]);
        }
        jac running, total;
        jac_set_id(&running); jac_set_id(&total);
        for (size_t b = nb; b-- > 0;) {
            jac_add(&running, &running, &buckets[b]);
            jac_add(&total, &total, &running);
        }
        wsum[w] = total;
    }
    acc = wsum[nwin - 1];
    for (int w = (int)nwin - 2; w >= 0; --w) {
        for (unsigned k = 0; k < c; ++k) jac_double(&acc, &acc);
        jac_add(&acc, &acc, &wsum[w]);
    }
    memcpy(out, &acc, sizeof(acc));
    free(s); free(buckets); free(wsum);
}
