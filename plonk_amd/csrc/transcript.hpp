// SHA-256 and the Fiat-Shamir transcript of backend_new/src/transcript.rs:39-173, byte for byte:
// a streaming SHA-256 state that is finalised at every squeeze and re-seeded with the digest.
// Host only: a proof needs about twenty small hashes.
#pragma once
#include <stdint.h>
#include <string.h>
#include <string>

#include "host_field.hpp"

namespace plonk {
namespace host {

class Sha256 {
  public:
    Sha256() { reset(); }
    void reset() {
        static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                                       0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
        memcpy(h_, iv, sizeof(iv));
        len_ = 0;
        fill_ = 0;
    }
    void update(const void* data, size_t n) {
        const uint8_t* p = static_cast<const uint8_t*>(data);
        len_ += n;
        while (n) {
            size_t take = 64 - fill_;
            if (take > n) take = n;
            memcpy(buf_ + fill_, p, take);
            fill_ += take;
            p += take;
            n -= take;
            if (fill_ == 64) {
                block(buf_);
                fill_ = 0;
            }
        }
    }
    void finalize(uint8_t out[32]) {
        uint64_t bits = len_ * 8;
        uint8_t pad = 0x80;
        update(&pad, 1);
        uint8_t z = 0;
        while (fill_ != 56) update(&z, 1);
        uint8_t lb[8];
        for (int i = 0; i < 8; ++i) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(lb, 8);
        for (int i = 0; i < 8; ++i) {
            out[4 * i] = (uint8_t)(h_[i] >> 24);
            out[4 * i + 1] = (uint8_t)(h_[i] >> 16);
            out[4 * i + 2] = (uint8_t)(h_[i] >> 8);
            out[4 * i + 3] = (uint8_t)h_[i];
        }
    }

  private:
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
            0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
            0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
            0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
            0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
            0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
            0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
            0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; ++i)
            w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h_[0], b = h_[1], c = h_[2], d = h_[3], e = h_[4], f = h_[5], g = h_[6], hh = h_[7];
        for (int i = 0; i < 64; ++i) {
            uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
            uint32_t ch = (e & f) ^ (~e & g);
            uint32_t t1 = hh + S1 + ch + K[i] + w[i];
            uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
            uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
            uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h_[0] += a; h_[1] += b; h_[2] += c; h_[3] += d; h_[4] += e; h_[5] += f; h_[6] += g; h_[7] += hh;
    }
    uint32_t h_[8];
    uint64_t len_;
    uint8_t buf_[64];
    size_t fill_;
};

class Transcript {
  public:
    // transcript.rs:41-43
    explicit Transcript(const char* domain_separator) { h_.update(domain_separator, strlen(domain_separator)); }
    // transcript.rs:57-61: label || u64_le(len) || data
    void absorb_bytes(const std::string& label, const uint8_t* data, size_t n) {
        h_.update(label.data(), label.size());
        uint8_t le[8];
        for (int i = 0; i < 8; ++i) le[i] = (uint8_t)((uint64_t)n >> (8 * i));
        h_.update(le, 8);
        h_.update(data, n);
    }
    // transcript.rs:71-74
    void absorb_fr(const std::string& label, const fr_h& x_mont) {
        uint8_t b[32];
        fr_to_bytes_le(x_mont, b);
        absorb_bytes(label, b, 32);
    }
    // transcript.rs:77-80: label || 48 compressed bytes, no length prefix
    void absorb_g1(const std::string& label, const G1AffineH& p) {
        uint8_t b[48];
        g1_compress(p, b);
        h_.update(label.data(), label.size());
        h_.update(b, 48);
    }
    // transcript.rs:110-131 + hash_to_fr :157-173
    fr_h squeeze_challenge(const std::string& label) {
        h_.update(label.data(), label.size());
        uint8_t d[32];
        h_.finalize(d);
        h_.reset();
        h_.update(d, 32);
        fr_h out;
        while (!fr_from_bytes_le(d, &out)) {
            Sha256 again;
            again.update(d, 32);
            again.finalize(d);
        }
        return out;
    }

  private:
    Sha256 h_;
};

}  // namespace host
}  // namespace plonk
