// Host-side input generator of the reference's tests and benches:
// MockEmbedder::hash_to_vector (src/embed.rs:124-155 of the reference).
//
// The reference hashes with std's DefaultHasher, which today is SipHash-1-3
// with zero keys (std documents the algorithm as unspecified). One hasher
// lives across all dimensions: for i in 0..dim it absorbs the text (its bytes
// plus the 0xFF terminator of str::hash) and i as 8 little-endian bytes, and
// reads finish() WITHOUT resetting, so value i depends on everything absorbed
// before. value = (hash as f32 / 2^64) * 2 - 1, then optional L2 normalisation
// with a sequential f32 sum.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "trag.h"
#include "trag_error.h"

namespace {

inline uint64_t rotl(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }

struct Sip13 {
    uint64_t v0 = 0x736f6d6570736575ull, v1 = 0x646f72616e646f6dull, v2 = 0x6c7967656e657261ull,
             v3 = 0x7465646279746573ull;  // keys (0, 0)
    uint64_t tail = 0;                    // pending bytes, little-endian
    unsigned ntail = 0;
    uint64_t length = 0;

    static inline void round(uint64_t& a, uint64_t& b, uint64_t& c, uint64_t& d) {
        a += b; b = rotl(b, 13); b ^= a; a = rotl(a, 32);
        c += d; d = rotl(d, 16); d ^= c;
        a += d; d = rotl(d, 21); d ^= a;
        c += b; b = rotl(b, 17); b ^= c; c = rotl(c, 32);
    }
    inline void word(uint64_t m) {
        v3 ^= m;
        round(v0, v1, v2, v3);
        v0 ^= m;
    }
    void write(const unsigned char* p, size_t n) {
        length += n;
        for (size_t i = 0; i < n; ++i) {
            tail |= (uint64_t)p[i] << (8 * ntail);
            if (++ntail == 8) {
                word(tail);
                tail = 0;
                ntail = 0;
            }
        }
    }
    uint64_t finish() const {
        uint64_t a = v0, b = v1, c = v2, d = v3;
        const uint64_t last = ((length & 0xff) << 56) | tail;
        d ^= last;
        round(a, b, c, d);
        a ^= last;
        c ^= 0xff;
        round(a, b, c, d);
        round(a, b, c, d);
        round(a, b, c, d);
        return a ^ b ^ c ^ d;
    }
};

}  // namespace

extern "C" trag_status trag_mock_embed(const char* text, uint64_t len, uint32_t dim, int normalize,
                                       float* out) {
    if ((!text && len) || (!out && dim)) return trag::set_error(TRAG_INVALID_ARG, "trag_mock_embed: null argument");
    Sip13 h;
    const unsigned char term = 0xFF;
    for (uint32_t i = 0; i < dim; ++i) {
        h.write(reinterpret_cast<const unsigned char*>(text), (size_t)len);
        h.write(&term, 1);
        unsigned char idx[8];
        const uint64_t v = i;
        for (int b = 0; b < 8; ++b) idx[b] = (unsigned char)(v >> (8 * b));
        h.write(idx, 8);
        const float f = (float)h.finish();
        out[i] = (f / 18446744073709551616.0f) * 2.0f - 1.0f;  // u64::MAX as f32 == 2^64
    }
    if (normalize) {
        float sum = 0.0f;
        for (uint32_t i = 0; i < dim; ++i) {
            const float p = out[i] * out[i];
            sum = sum + p;
        }
        const float norm = std::sqrt(sum);
        if (norm > 0.0f)
            for (uint32_t i = 0; i < dim; ++i) out[i] /= norm;
    }
    return TRAG_OK;
}
