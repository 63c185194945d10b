// The synthetic 2^k-gate benchmark circuit, generated on the host in native code (the Python
// generator of plonk_amd/circuits.py takes minutes at 2^24 gates).  Gate i even: a*b = c (q_M = 1,
// q_O = -1), odd: a+b = c (q_L = q_R = 1, q_O = -1); a[i+1] = c[i] by copy constraint; a[0], every
// b[i] and then the eleven blinding scalars come from splitmix64(0x504C4F4E4B000000 + k), four
// outputs per scalar as little-endian limbs with the top limb masked to 62 bits; the SRS trapdoor
// from splitmix64(0x5352530000000000 + k).  sigma in closed form: the copy classes {a[i+1], c[i]}
// are the 2-cycles a[i+1] <-> c[i] that Permutation::compute (backend_new/src/plonk/
// permutation.rs:55-140) produces for them; wire A is encoded as w^i, B as k1 w^i, C as k2 w^i
// (permutation.rs:289-296), k1 / k2 = first k >= 6 / 7 with k^n != 1 (permutation.rs:307-326).
// Input generation only: nothing here is on the timed path.  No counterpart in the reference.
#include "common.hpp"

namespace plonk {

namespace {
struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    host::fr_h fr_rand() {   // canonical value below 2^254 < r, returned in Montgomery form
        host::fr_h v;
        for (int i = 0; i < 4; ++i) v.l[i] = next();
        v.l[3] &= (1ull << 62) - 1;
        return host::Fr::to_mont(v);
    }
};

host::fr_h coset_generator(uint64_t n, uint64_t index) {
    uint64_t k = 5 + index;
    while (host::Fr::pow_u64(host::Fr::from_u64(k), n) == host::Fr::one()) ++k;
    return host::Fr::from_u64(k);
}

void put(plonk_fr* dst, const host::fr_h& v) { memcpy(dst->l, v.l, 32); }
}  // namespace

void synthetic_circuit(uint32_t log_n, plonk_fr* a, plonk_fr* b, plonk_fr* c, plonk_fr* const sel[5], plonk_fr* const sig[3],
                       plonk_fr blinding[11], plonk_fr* tau, plonk_fr* k1_out, plonk_fr* k2_out) {
    using host::Fr;
    using host::fr_h;
    PLONK_REQUIRE(log_n >= 2 && log_n <= 26, "synthetic circuit: 2^2 .. 2^26 gates");
    const size_t n = (size_t)1 << log_n;
    SplitMix64 rng{0x504C4F4E4B000000ull + log_n};
    const fr_h one = Fr::one(), zero = Fr::zero(), neg1 = Fr::neg(one);
    const fr_h k1 = coset_generator(n, 1), k2 = coset_generator(n, 2);
    const fr_h omega = host::fr_root_of_unity(log_n);
    fr_h cur = rng.fr_rand();
    fr_h w = one;        // w^i
    fr_h w_prev = one;   // w^(i-1)
    for (size_t i = 0; i < n; ++i) {
        const fr_h bi = rng.fr_rand();
        put(a + i, cur);
        put(b + i, bi);
        const bool is_mul = (i & 1) == 0;
        cur = is_mul ? Fr::mul(cur, bi) : Fr::add(cur, bi);
        put(c + i, cur);
        put(sel[0] + i, is_mul ? one : zero);
        put(sel[1] + i, is_mul ? zero : one);
        put(sel[2] + i, is_mul ? zero : one);
        put(sel[3] + i, neg1);
        put(sel[4] + i, zero);
        const fr_h w_next = Fr::mul(w, omega);
        // sigma(a[i]) = c[i-1] (i >= 1), sigma(b[i]) = b[i], sigma(c[i]) = a[i+1] (i <= n-2)
        put(sig[0] + i, i == 0 ? one : Fr::mul(k2, w_prev));
        put(sig[1] + i, Fr::mul(k1, w));
        put(sig[2] + i, i + 1 < n ? w_next : Fr::mul(k2, w));
        w_prev = w;
        w = w_next;
    }
    for (int j = 0; j < 11; ++j) put(blinding + j, rng.fr_rand());
    SplitMix64 srs_rng{0x5352530000000000ull + log_n};
    put(tau, srs_rng.fr_rand());
    put(k1_out, k1);
    put(k2_out, k2);
}

}  // namespace plonk
