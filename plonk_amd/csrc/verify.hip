// Host-side verifier and the CompactProof JSON wire format (no device work).
//
// verify_3wire restates Verifier::verify (backend_new/src/plonk/verifier.rs:42-314): transcript
// replay, Z_H(zeta), L1(zeta), PI(zeta), the linearisation commitment [r]_1, [F]_1, [E]_1 and the
// single pairing check e(W_zeta + u W_zeta_omega, [tau]_2) == e(zeta W_zeta + u zeta omega
// W_zeta_omega + F - E, [1]_2).  O(1) work per proof -- a few hundred G1 operations and one
// two-pairing product -- which the reference also keeps on the CPU.
// proof_to_json / proof_from_json are serde_json's image of CompactProof (trace.rs:255-332) with the
// element encodings of curve.rs:418-450 and field.rs:294-320.
#include <string>
#include <vector>

#include "common.hpp"
#include "pairing.hpp"
#include "transcript.hpp"

namespace plonk {

using host::Fr;
using host::fr_h;
using host::G1AffineH;
using host::G1XyzzH;

static G1AffineH g1_in(const plonk_g1& p) {
    G1AffineH a;
    memcpy(a.x.l, p.x, 48);
    memcpy(a.y.l, p.y, 48);
    a.inf = p.infinity != 0;
    if (a.inf) {
        a.x = host::Fq::zero();
        a.y = host::Fq::zero();
    }
    return a;
}
static fr_h fr_in_host(const plonk_fr& x) {
    fr_h h;
    memcpy(h.l, x.l, 32);
    return h;
}
static host::G2AffineH g2_in(const plonk_g2& p) {
    host::G2AffineH a;
    memcpy(a.x.c0.l, p.x_c0, 48);
    memcpy(a.x.c1.l, p.x_c1, 48);
    memcpy(a.y.c0.l, p.y_c0, 48);
    memcpy(a.y.c1.l, p.y_c1, 48);
    a.inf = p.infinity != 0;
    return a;
}
void g2_out(const host::G2AffineH& a, plonk_g2* p) {
    memset(p, 0, sizeof(*p));
    if (a.inf) {
        p->infinity = 1;
        return;
    }
    memcpy(p->x_c0, a.x.c0.l, 48);
    memcpy(p->x_c1, a.x.c1.l, 48);
    memcpy(p->y_c0, a.y.c0.l, 48);
    memcpy(p->y_c1, a.y.c1.l, 48);
}

static bool fq_reduced(const host::fq_h& a) { return !host::Fq::geq_p(a); }

// a G1 input the verifier can compute with: reduced coordinates, on the curve, in the order-r subgroup
// (what G1Affine::from_compressed_bytes guarantees for the points of a deserialised proof, curve.rs:90-93)
static bool g1_valid(const G1AffineH& a) {
    if (a.inf) return true;
    if (!fq_reduced(a.x) || !fq_reduced(a.y) || !host::g1_on_curve(a)) return false;
    const fr_h r_minus_1 = Fr::from_mont(Fr::neg(Fr::one()));
    G1XyzzH t = host::g1_mul_limbs(host::g1_from_affine(a), r_minus_1.l, 4);
    return host::g1_is_inf(host::g1_add(t, host::g1_from_affine(a)));
}

static G1XyzzH g1_scale(const G1AffineH& p, const fr_h& k_mont) {
    const fr_h k = Fr::from_mont(k_mont);
    return host::g1_mul_limbs(host::g1_from_affine(p), k.l, 4);
}

int verify_3wire(const plonk_vk* vk, const uint64_t* pi_positions, size_t n_positions, const plonk_g2* g2_tau_in,
                 const plonk_proof* pr, const plonk_fr* pub, size_t n_pub, bool* accepted) {
    *accepted = false;
    const uint64_t n = vk->n;
    if (n == 0 || (n & (n - 1))) return PLONK_E_INVALID;
    const fr_h omega = fr_in_host(vk->omega), k1 = fr_in_host(vk->k1), k2 = fr_in_host(vk->k2);
    if (Fr::geq_p(omega) || Fr::geq_p(k1) || Fr::geq_p(k2)) return PLONK_E_INVALID;
    const plonk_g1* vk_pts[8] = {&vk->q_m_comm, &vk->q_l_comm, &vk->q_r_comm, &vk->q_o_comm, &vk->q_c_comm,
                                 &vk->s_sigma1_comm, &vk->s_sigma2_comm, &vk->s_sigma3_comm};
    const plonk_g1* pr_pts[9] = {&pr->a_comm, &pr->b_comm, &pr->c_comm, &pr->z_comm, &pr->t_lo_comm,
                                 &pr->t_mid_comm, &pr->t_hi_comm, &pr->w_zeta_comm, &pr->w_zeta_omega_comm};
    G1AffineH V[8], C[9];
    for (int i = 0; i < 8; ++i) {
        V[i] = g1_in(*vk_pts[i]);
        if (!g1_valid(V[i])) return PLONK_E_INVALID;
    }
    for (int i = 0; i < 9; ++i) {
        C[i] = g1_in(*pr_pts[i]);
        if (!g1_valid(C[i])) return PLONK_E_INVALID;
    }
    const plonk_fr* pr_ev[6] = {&pr->a_eval, &pr->b_eval, &pr->c_eval, &pr->s_sigma1_eval, &pr->s_sigma2_eval, &pr->z_omega_eval};
    fr_h ev[6];
    for (int i = 0; i < 6; ++i) {
        ev[i] = fr_in_host(*pr_ev[i]);
        if (Fr::geq_p(ev[i])) return PLONK_E_INVALID;
    }
    std::vector<fr_h> pi(n_pub);
    for (size_t i = 0; i < n_pub; ++i) {
        pi[i] = fr_in_host(pub[i]);
        if (Fr::geq_p(pi[i])) return PLONK_E_INVALID;
    }
    for (size_t i = 0; i < n_positions; ++i)
        if (pi_positions[i] >= n) return PLONK_E_INVALID;
    const host::G2AffineH g2_tau = g2_in(*g2_tau_in);
    if (!g2_tau.inf && (!fq_reduced(g2_tau.x.c0) || !fq_reduced(g2_tau.x.c1) || !fq_reduced(g2_tau.y.c0) ||
                        !fq_reduced(g2_tau.y.c1) || !host::g2_on_curve(g2_tau) || !host::g2_in_subgroup(g2_tau)))
        return PLONK_E_INVALID;   // ark's G2Affine cannot hold a point outside the order-r subgroup
    const G1AffineH &a_comm = C[0], &b_comm = C[1], &c_comm = C[2], &z_comm = C[3], &t_lo = C[4], &t_mid = C[5], &t_hi = C[6],
                    &w_zeta = C[7], &w_zeta_omega = C[8];
    const fr_h &a_bar = ev[0], &b_bar = ev[1], &c_bar = ev[2], &s1_bar = ev[3], &s2_bar = ev[4], &zw_bar = ev[5];

    // step 1: transcript (verifier.rs:51-92)
    host::Transcript tr("PLONK-v1");
    for (size_t i = 0; i < n_pub; ++i) tr.absorb_fr("pi_" + std::to_string(i), pi[i]);
    tr.absorb_g1("a_comm", a_comm); tr.absorb_g1("b_comm", b_comm); tr.absorb_g1("c_comm", c_comm);
    const fr_h beta = tr.squeeze_challenge("beta"), gamma = tr.squeeze_challenge("gamma");
    tr.absorb_g1("z_comm", z_comm);
    const fr_h alpha = tr.squeeze_challenge("alpha");
    tr.absorb_g1("t_lo_comm", t_lo); tr.absorb_g1("t_mid_comm", t_mid); tr.absorb_g1("t_hi_comm", t_hi);
    const fr_h zeta = tr.squeeze_challenge("zeta");
    tr.absorb_fr("a_zeta", a_bar); tr.absorb_fr("b_zeta", b_bar); tr.absorb_fr("c_zeta", c_bar);
    tr.absorb_fr("s_sigma1_zeta", s1_bar); tr.absorb_fr("s_sigma2_zeta", s2_bar); tr.absorb_fr("z_zeta_omega", zw_bar);
    const fr_h v = tr.squeeze_challenge("v");
    tr.absorb_g1("w_zeta_comm", w_zeta); tr.absorb_g1("w_zeta_omega_comm", w_zeta_omega);
    const fr_h u = tr.squeeze_challenge("u");

    // steps 2-4: Z_H(zeta), L1(zeta), PI(zeta) (:106-143)
    const fr_h one = Fr::one(), n_fr = Fr::from_u64(n);
    const fr_h zeta_n = Fr::pow_u64(zeta, n);
    const fr_h zh_zeta = Fr::sub(zeta_n, one);
    const fr_h l1_zeta = zeta == one ? one : Fr::mul(zh_zeta, Fr::inv(Fr::mul(n_fr, Fr::sub(zeta, one))));
    fr_h pi_zeta = Fr::zero();
    for (size_t i = 0; i < n_pub; ++i) {
        if (i >= n_positions) continue;   // :127-131
        const fr_h wp = Fr::pow_u64(omega, pi_positions[i]);
        if (zeta == wp) continue;         // :134
        const fr_h lp = Fr::mul(Fr::mul(wp, zh_zeta), Fr::inv(Fr::mul(n_fr, Fr::sub(zeta, wp))));
        pi_zeta = Fr::sub(pi_zeta, Fr::mul(pi[i], lp));
    }

    // step 5: [r]_1 (:160-202)
    const fr_h bz = Fr::mul(beta, zeta);
    fr_h pc1 = Fr::add(Fr::add(a_bar, bz), gamma);
    pc1 = Fr::mul(pc1, Fr::add(Fr::add(b_bar, Fr::mul(k1, bz)), gamma));
    pc1 = Fr::mul(pc1, Fr::add(Fr::add(c_bar, Fr::mul(k2, bz)), gamma));
    const fr_h f1 = Fr::add(Fr::add(a_bar, Fr::mul(beta, s1_bar)), gamma);
    const fr_h f2 = Fr::add(Fr::add(b_bar, Fr::mul(beta, s2_bar)), gamma);
    const fr_h pc2 = Fr::neg(Fr::mul(Fr::mul(Fr::mul(f1, f2), beta), zw_bar));
    const fr_h alpha2 = Fr::mul(alpha, alpha);
    const fr_h zeta_2n = Fr::mul(zeta_n, zeta_n);
    G1XyzzH r_comm = host::g1_identity();
    auto acc = [&](G1XyzzH& dst, const G1AffineH& p, const fr_h& k) { dst = host::g1_add(dst, g1_scale(p, k)); };
    acc(r_comm, V[0], Fr::mul(a_bar, b_bar));
    acc(r_comm, V[1], a_bar);
    acc(r_comm, V[2], b_bar);
    acc(r_comm, V[3], c_bar);
    r_comm = host::g1_add(r_comm, host::g1_from_affine(V[4]));
    acc(r_comm, z_comm, Fr::add(Fr::mul(alpha, pc1), Fr::mul(alpha2, l1_zeta)));
    acc(r_comm, V[7], Fr::mul(alpha, pc2));
    const fr_h t_coeff = Fr::neg(zh_zeta);
    acc(r_comm, t_lo, t_coeff);
    acc(r_comm, t_mid, Fr::mul(t_coeff, zeta_n));
    acc(r_comm, t_hi, Fr::mul(t_coeff, zeta_2n));

    // steps 6-8: [F]_1, r_0, [E]_1 (:244-287)
    const fr_h v2 = Fr::mul(v, v), v3 = Fr::mul(v2, v), v4 = Fr::mul(v3, v), v5 = Fr::mul(v4, v);
    G1XyzzH f_comm = r_comm;
    acc(f_comm, a_comm, v);
    acc(f_comm, b_comm, v2);
    acc(f_comm, c_comm, v3);
    acc(f_comm, V[5], v4);
    acc(f_comm, V[6], v5);
    acc(f_comm, z_comm, u);
    const fr_h perm_term = Fr::mul(Fr::mul(Fr::mul(Fr::mul(alpha, zw_bar), f1), f2), Fr::add(c_bar, gamma));
    const fr_h r0 = Fr::sub(Fr::sub(pi_zeta, Fr::mul(alpha2, l1_zeta)), perm_term);
    fr_h e_eval = Fr::neg(r0);
    e_eval = Fr::add(e_eval, Fr::mul(v, a_bar));
    e_eval = Fr::add(e_eval, Fr::mul(v2, b_bar));
    e_eval = Fr::add(e_eval, Fr::mul(v3, c_bar));
    e_eval = Fr::add(e_eval, Fr::mul(v4, s1_bar));
    e_eval = Fr::add(e_eval, Fr::mul(v5, s2_bar));
    e_eval = Fr::add(e_eval, Fr::mul(u, zw_bar));
    const G1XyzzH e_comm = g1_scale(host::g1_generator(), e_eval);

    // step 9: the pairing check (:290-314)
    G1XyzzH w_comb = host::g1_from_affine(w_zeta);
    acc(w_comb, w_zeta_omega, u);
    G1XyzzH right = f_comm;
    acc(right, w_zeta, zeta);
    acc(right, w_zeta_omega, Fr::mul(u, Fr::mul(zeta, omega)));
    right = host::g1_add(right, host::g1_neg(e_comm));
    *accepted = host::pairing_check(host::g1_to_affine(w_comb), g2_tau, host::g1_to_affine(right), host::g2_generator());
    return PLONK_OK;
}

// ---- CompactProof JSON ----
static void hex_append(std::string& s, const uint8_t* b, size_t n) {
    static const char* d = "0123456789abcdef";
    for (size_t i = 0; i < n; ++i) {
        s.push_back(d[b[i] >> 4]);
        s.push_back(d[b[i] & 15]);
    }
}

static void json_g1(std::string& s, const plonk_g1& p) {   // curve.rs:418-434
    const G1AffineH a = g1_in(p);
    uint8_t comp[48];
    host::g1_compress(a, comp);
    s += "{\"compressed\":\"0x";
    hex_append(s, comp, 48);
    if (a.inf) {
        s += "\",\"x\":\"infinity\",\"y\":\"infinity\"}";   // curve.rs:112-113,124-125
        return;
    }
    const host::fq_h x = host::Fq::from_mont(a.x), y = host::Fq::from_mont(a.y);
    uint8_t be[48];
    auto big_endian = [&](const host::fq_h& v) {
        for (int i = 0; i < 6; ++i)
            for (int b = 0; b < 8; ++b) be[47 - (i * 8 + b)] = (uint8_t)(v.l[i] >> (8 * b));
    };
    s += "\",\"x\":\"0x";
    big_endian(x);
    hex_append(s, be, 48);
    s += "\",\"y\":\"0x";
    big_endian(y);
    hex_append(s, be, 48);
    s += "\"}";
}

static void json_fr(std::string& s, const plonk_fr& x) {   // field.rs:294-308, :128-144
    uint8_t le[32], be[32];
    host::fr_to_bytes_le(fr_in_host(x), le);
    for (int i = 0; i < 32; ++i) be[i] = le[31 - i];
    std::string full = "0x";
    hex_append(full, be, 32);
    s += "{\"full\":\"" + full + "\",\"short\":\"" + full.substr(0, 10) + "..." + full.substr(full.size() - 8) + "\"}";
}

std::string proof_to_json(const plonk_proof* p) {
    std::string s;
    s.reserve(4096);
    s += "{\"round1\":{\"a\":"; json_g1(s, p->a_comm);
    s += ",\"b\":"; json_g1(s, p->b_comm);
    s += ",\"c\":"; json_g1(s, p->c_comm);
    s += "},\"round2\":{\"z\":"; json_g1(s, p->z_comm);
    s += "},\"round3\":{\"t_lo\":"; json_g1(s, p->t_lo_comm);
    s += ",\"t_mid\":"; json_g1(s, p->t_mid_comm);
    s += ",\"t_hi\":"; json_g1(s, p->t_hi_comm);
    s += "},\"round4\":{\"a_zeta\":"; json_fr(s, p->a_eval);
    s += ",\"b_zeta\":"; json_fr(s, p->b_eval);
    s += ",\"c_zeta\":"; json_fr(s, p->c_eval);
    s += ",\"s_sigma1_zeta\":"; json_fr(s, p->s_sigma1_eval);
    s += ",\"s_sigma2_zeta\":"; json_fr(s, p->s_sigma2_eval);
    s += ",\"z_omega_zeta\":"; json_fr(s, p->z_omega_eval);
    s += "},\"round5\":{\"w_zeta\":"; json_g1(s, p->w_zeta_comm);
    s += ",\"w_zeta_omega\":"; json_g1(s, p->w_zeta_omega_comm);
    s += "}}";
    return s;
}

// A reader for the subset of JSON the format needs: nested objects whose leaves are strings (other
// value kinds are skipped).  Collects "path.to.key" -> string value.
namespace {
struct JsonReader {
    const char* p;
    const char* end;
    std::vector<std::pair<std::string, std::string>> leaves;
    std::vector<std::string> objects;   // path of every object met, in order (a key given twice shows up twice)
    bool ok = true;
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool string(std::string* out) {
        if (p >= end || *p != '"') return false;
        ++p;
        out->clear();
        while (p < end && *p != '"') {
            if (*p == '\\') {   // the values of this format hold no escapes; accept the simple ones
                if (++p >= end) return false;
                switch (*p) {
                    case '"': out->push_back('"'); break;
                    case '\\': out->push_back('\\'); break;
                    case '/': out->push_back('/'); break;
                    case 'n': out->push_back('\n'); break;
                    case 't': out->push_back('\t'); break;
                    case 'r': out->push_back('\r'); break;
                    case 'b': out->push_back('\b'); break;
                    case 'f': out->push_back('\f'); break;
                    default: return false;
                }
                ++p;
            } else {
                out->push_back(*p++);
            }
        }
        if (p >= end) return false;
        ++p;
        return true;
    }
    bool value(const std::string& path, int depth) {
        if (depth > 16) return false;
        ws();
        if (p >= end) return false;
        if (*p == '{') {
            objects.push_back(path);
            ++p;
            ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                ws();
                std::string key;
                if (!string(&key)) return false;
                ws();
                if (p >= end || *p != ':') return false;
                ++p;
                if (!value(path.empty() ? key : path + "." + key, depth + 1)) return false;
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                return false;
            }
        }
        if (*p == '[') {
            ++p;
            ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (size_t i = 0;; ++i) {
                if (!value(path + "[" + std::to_string(i) + "]", depth + 1)) return false;
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return false;
            }
        }
        if (*p == '"') {
            std::string v;
            if (!string(&v)) return false;
            leaves.emplace_back(path, v);
            return true;
        }
        // number / true / false / null: skip the token
        const char* start = p;
        while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n' && *p != '\t' && *p != '\r') ++p;
        return p > start;
    }
    const std::string* find(const std::string& path) const {
        for (const auto& kv : leaves)
            if (kv.first == path) return &kv.second;
        return nullptr;
    }
    // how often a field the format knows was given: serde's derived Deserialize refuses a known field given twice
    // ("duplicate field"); unknown fields it skips, however often they come
    size_t count_leaf(const std::string& path) const {
        size_t c = 0;
        for (const auto& kv : leaves) c += kv.first == path;
        return c;
    }
    size_t count_object(const std::string& path) const {
        size_t c = 0;
        for (const auto& o : objects) c += o == path;
        return c;
    }
};

bool unhex(const std::string& in, uint8_t* out, size_t n) {
    size_t off = (in.size() >= 2 && in[0] == '0' && in[1] == 'x') ? 2 : 0;   // strip_prefix("0x")
    if (in.size() - off != 2 * n) return false;
    auto nib = [](char c) -> int {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    };
    for (size_t i = 0; i < n; ++i) {
        const int hi = nib(in[off + 2 * i]), lo = nib(in[off + 2 * i + 1]);
        if (hi < 0 || lo < 0) return false;
        out[i] = (uint8_t)(hi * 16 + lo);
    }
    return true;
}
}  // namespace

// What serde's Deserialize of CompactProof reads: `compressed` of every point (curve.rs:436-450)
// and `full` of every scalar (field.rs:310-320); everything else is ignored.
bool proof_from_json(const char* json, size_t len, plonk_proof* out) {
    JsonReader rd{json, json + len, {}, {}, true};
    if (!rd.value("", 0)) return false;
    rd.ws();
    if (rd.p != rd.end) return false;
    static const char* g1_paths[9] = {"round1.a", "round1.b", "round1.c", "round2.z", "round3.t_lo", "round3.t_mid",
                                      "round3.t_hi", "round5.w_zeta", "round5.w_zeta_omega"};
    static const char* fr_paths[6] = {"round4.a_zeta", "round4.b_zeta", "round4.c_zeta", "round4.s_sigma1_zeta",
                                      "round4.s_sigma2_zeta", "round4.z_omega_zeta"};
    plonk_proof pr;
    memset(&pr, 0, sizeof(pr));
    plonk_g1* g[9] = {&pr.a_comm, &pr.b_comm, &pr.c_comm, &pr.z_comm, &pr.t_lo_comm, &pr.t_mid_comm, &pr.t_hi_comm,
                      &pr.w_zeta_comm, &pr.w_zeta_omega_comm};
    plonk_fr* f[6] = {&pr.a_eval, &pr.b_eval, &pr.c_eval, &pr.s_sigma1_eval, &pr.s_sigma2_eval, &pr.z_omega_eval};
    for (const char* round : {"round1", "round2", "round3", "round4", "round5"})
        if (rd.count_object(round) > 1) return false;
    for (int k = 0; k < 9; ++k)
        if (rd.count_object(g1_paths[k]) > 1 || rd.count_leaf(std::string(g1_paths[k]) + ".compressed") > 1) return false;
    for (int k = 0; k < 6; ++k)
        if (rd.count_object(fr_paths[k]) > 1 || rd.count_leaf(std::string(fr_paths[k]) + ".full") > 1) return false;
    for (int k = 0; k < 9; ++k) {
        const std::string* v = rd.find(std::string(g1_paths[k]) + ".compressed");
        uint8_t comp[48];
        if (!v || !unhex(*v, comp, 48)) return false;
        G1AffineH a;
        if (!host::g1_decompress(comp, &a)) return false;
        memset(g[k], 0, sizeof(plonk_g1));
        if (a.inf) {
            g[k]->infinity = 1;
        } else {
            memcpy(g[k]->x, a.x.l, 48);
            memcpy(g[k]->y, a.y.l, 48);
        }
    }
    for (int k = 0; k < 6; ++k) {
        const std::string* v = rd.find(std::string(fr_paths[k]) + ".full");
        uint8_t be[32], le[32];
        if (!v || !unhex(*v, be, 32)) return false;
        for (int i = 0; i < 32; ++i) le[i] = be[31 - i];
        fr_h h;
        if (!host::fr_from_bytes_le(le, &h)) return false;
        memcpy(f[k]->l, h.l, 32);
    }
    *out = pr;
    return true;
}

}  // namespace plonk
