// Edwards25519 / SHA-512 device helpers shared by the native verifier (vx_justification.hip) and the EdDSA witness
// generators (vx_ed_air.hip): GF(2^255-19) as 8 x 32-bit limbs with schoolbook products and the 2^256 = 38 fold, extended
// twisted-Edwards points (a = -1), RFC 8032 decoding / encoding, FIPS 180-4 SHA-512, reduction mod the group order.
// Written for obviousness, not speed: a few hundred signatures per proof.
#pragma once
#include <stdint.h>

#include "ed25519_constants.h"

namespace ed {
struct U256 {
    uint32_t w[8];
};
static __device__ const uint32_t FE_P[8] = ED_P_INIT;
// group order L = 2^252 + 27742317777372353535851937790883648493
static __device__ const uint32_t SC_L[8] = ED_L_INIT;
// d = -121665/121666, sqrt(-1), base point (x, y)
static __device__ const uint32_t ED_D[8] = ED_D_INIT;
static __device__ const uint32_t ED_D2[8] = ED_D2_INIT;
static __device__ const uint32_t ED_SQRTM1[8] = ED_SQRTM1_INIT;
static __device__ const uint32_t ED_BX[8] = ED_BX_INIT;
static __device__ const uint32_t ED_BY[8] = ED_BY_INIT;

static __device__ bool u256_geq(const uint32_t* a, const uint32_t* b) {
    for (int i = 7; i >= 0; --i) {
        if (a[i] > b[i]) return true;
        if (a[i] < b[i]) return false;
    }
    return true;
}
static __device__ uint32_t u256_add(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint64_t c = 0;
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t)a[i] + b[i];
        r[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)c;
}
static __device__ uint32_t u256_sub(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    int64_t c = 0;
    for (int i = 0; i < 8; ++i) {
        c += (int64_t)a[i] - b[i];
        r[i] = (uint32_t)c;
        c >>= 32;
    }
    return (uint32_t)(c & 1);
}
// canonical representative in [0, p)
static __device__ void fe_canon(U256& a) {
    while (u256_geq(a.w, FE_P)) u256_sub(a.w, a.w, FE_P);
}
static __device__ U256 fe_add(const U256& a, const U256& b) {
    U256 r;
    uint32_t c = u256_add(r.w, a.w, b.w);
    while (c) {  // 2^256 = 38 (mod p)
        uint32_t t[8] = {38, 0, 0, 0, 0, 0, 0, 0};
        c = u256_add(r.w, r.w, t);
    }
    return r;
}
static __device__ U256 fe_sub(const U256& a, const U256& b) {
    U256 bb = b;
    fe_canon(bb);
    U256 np;  // a + (p - b), p - b in (0, p]
    u256_sub(np.w, FE_P, bb.w);
    return fe_add(a, np);
}
static __device__ U256 fe_mul(const U256& a, const U256& b) {
    uint32_t t[16];
    for (int i = 0; i < 16; ++i) t[i] = 0;
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
        for (int j = 0; j < 8; ++j) {
            c += (uint64_t)a.w[i] * b.w[j] + t[i + j];
            t[i + j] = (uint32_t)c;
            c >>= 32;
        }
        t[i + 8] = (uint32_t)c;
    }
    // fold the high half: hi * 2^256 = hi * 38
    U256 r;
    uint64_t c = 0;
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t)t[i] + (uint64_t)t[i + 8] * 38;
        r.w[i] = (uint32_t)c;
        c >>= 32;
    }
    while (c) {  // c < 39
        uint64_t cc = c * 38;
        c = 0;
        for (int i = 0; i < 8; ++i) {
            cc += r.w[i];
            r.w[i] = (uint32_t)cc;
            cc >>= 32;
        }
        c = cc;
    }
    return r;
}
static __device__ U256 fe_pow(const U256& a, const uint32_t* e) {  // e: 256-bit exponent
    U256 r;
    for (int i = 0; i < 8; ++i) r.w[i] = i == 0;
    for (int bit = 255; bit >= 0; --bit) {
        r = fe_mul(r, r);
        if ((e[bit >> 5] >> (bit & 31)) & 1) r = fe_mul(r, a);
    }
    return r;
}
static __device__ bool fe_eq(U256 a, U256 b) {
    fe_canon(a);
    fe_canon(b);
    for (int i = 0; i < 8; ++i)
        if (a.w[i] != b.w[i]) return false;
    return true;
}
static __device__ U256 fe_from(const uint32_t* c) {
    U256 r;
    for (int i = 0; i < 8; ++i) r.w[i] = c[i];
    return r;
}
struct Pt {
    U256 X, Y, Z, T;
};
static __device__ Pt pt_add(const Pt& p, const Pt& q) {  // extended twisted Edwards, a = -1 (add-2008-hwcd-3)
    U256 A = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    U256 B = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    U256 C = fe_mul(fe_mul(p.T, q.T), fe_from(ED_D2));
    U256 D = fe_mul(p.Z, q.Z);
    D = fe_add(D, D);
    U256 E = fe_sub(B, A), F = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    return {fe_mul(E, F), fe_mul(G, H), fe_mul(F, G), fe_mul(E, H)};
}
static __device__ Pt pt_identity() {
    Pt r;
    for (int i = 0; i < 8; ++i) r.X.w[i] = r.T.w[i] = 0, r.Y.w[i] = r.Z.w[i] = i == 0;
    return r;
}
static __device__ Pt pt_scalar_mul(const Pt& p, const uint32_t* k) {  // variable time, 256-bit scalar
    Pt acc = pt_identity();
    for (int bit = 255; bit >= 0; --bit) {
        acc = pt_add(acc, acc);
        if ((k[bit >> 5] >> (bit & 31)) & 1) acc = pt_add(acc, p);
    }
    return acc;
}
// a*P + b*Q in one pass (Shamir): one doubling and ONE addition per bit, the addend (identity, P, Q or P+Q) picked by
// data selects so that the lanes of a wave never diverge -- with a branch per bit every wave would execute all cases.
// The addition law is complete on this curve, so adding the identity is fine.
static __device__ Pt pt_double_scalar_mul(const Pt& p, const uint32_t* a, const Pt& q, const uint32_t* b) {
    const Pt pq = pt_add(p, q), id = pt_identity();
    Pt acc = id;
    for (int bit = 255; bit >= 0; --bit) {
        acc = pt_add(acc, acc);
        const bool ba = (a[bit >> 5] >> (bit & 31)) & 1, bb = (b[bit >> 5] >> (bit & 31)) & 1;
        Pt t;
        const U256* src[4][4] = {{&id.X, &id.Y, &id.Z, &id.T}, {&p.X, &p.Y, &p.Z, &p.T}, {&q.X, &q.Y, &q.Z, &q.T}, {&pq.X, &pq.Y, &pq.Z, &pq.T}};
        U256* dst[4] = {&t.X, &t.Y, &t.Z, &t.T};
        for (int c = 0; c < 4; ++c)
            for (int w = 0; w < 8; ++w) {
                const uint32_t lo = ba ? src[1][c]->w[w] : src[0][c]->w[w], hi = ba ? src[3][c]->w[w] : src[2][c]->w[w];
                dst[c]->w[w] = bb ? hi : lo;
            }
        acc = pt_add(acc, t);
    }
    return acc;
}
// RFC 8032 5.1.3 decoding; false when the encoding is not a curve point
static __device__ bool pt_decode(const uint8_t* s, Pt* out) {
    U256 y;
    for (int i = 0; i < 8; ++i) y.w[i] = (uint32_t)s[4 * i] | ((uint32_t)s[4 * i + 1] << 8) | ((uint32_t)s[4 * i + 2] << 16) | ((uint32_t)s[4 * i + 3] << 24);
    const uint32_t sign = y.w[7] >> 31;
    y.w[7] &= 0x7FFFFFFF;
    if (u256_geq(y.w, FE_P)) return false;
    U256 one;
    for (int i = 0; i < 8; ++i) one.w[i] = i == 0;
    const U256 y2 = fe_mul(y, y);
    const U256 u = fe_sub(y2, one), v = fe_add(fe_mul(y2, fe_from(ED_D)), one);
    // x = u v^3 (u v^7)^((p-5)/8)
    const U256 v3 = fe_mul(fe_mul(v, v), v), v7 = fe_mul(fe_mul(v3, v3), v);
    const uint32_t e58[8] = ED_EXP_P58_INIT;
    U256 x = fe_mul(fe_mul(u, v3), fe_pow(fe_mul(u, v7), e58));
    const U256 vx2 = fe_mul(v, fe_mul(x, x));
    U256 zero;
    for (int i = 0; i < 8; ++i) zero.w[i] = 0;
    if (!fe_eq(vx2, u)) {
        if (!fe_eq(vx2, fe_sub(zero, u))) return false;
        x = fe_mul(x, fe_from(ED_SQRTM1));
    }
    fe_canon(x);
    bool xz = true;
    for (int i = 0; i < 8; ++i) xz &= x.w[i] == 0;
    if (xz && sign) return false;
    if ((x.w[0] & 1) != sign) x = fe_sub(zero, x);
    *out = {x, y, one, fe_mul(x, y)};
    return true;
}
static __device__ void pt_encode(const Pt& p, uint8_t* out) {
    const uint32_t em2[8] = ED_EXP_PM2_INIT;
    const U256 zi = fe_pow(p.Z, em2);
    U256 x = fe_mul(p.X, zi), y = fe_mul(p.Y, zi);
    fe_canon(x);
    fe_canon(y);
    y.w[7] |= (x.w[0] & 1) << 31;
    for (int i = 0; i < 8; ++i)
        for (int b = 0; b < 4; ++b) out[4 * i + b] = (uint8_t)(y.w[i] >> (8 * b));
}

// ---- SHA-512 (FIPS 180-4)
static __device__ const uint64_t K512[80] = SHA512_K_INIT;
__device__ __forceinline__ uint64_t r64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
static __device__ void sha512(const uint8_t* msg, size_t len, uint8_t* out64) {  // len < 240
    uint64_t h[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                     0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    uint8_t buf[256];
    const size_t total = (len + 17 <= 128) ? 128 : 256;
    for (size_t i = 0; i < total; ++i) buf[i] = i < len ? msg[i] : 0;
    buf[len] = 0x80;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; ++i) buf[total - 1 - i] = (uint8_t)(bits >> (8 * i));
    for (size_t off = 0; off < total; off += 128) {
        uint64_t w[80];
        for (int i = 0; i < 16; ++i) {
            uint64_t v = 0;
            for (int b = 0; b < 8; ++b) v = (v << 8) | buf[off + 8 * i + b];
            w[i] = v;
        }
        for (int i = 16; i < 80; ++i) {
            const uint64_t s0 = r64(w[i - 15], 1) ^ r64(w[i - 15], 8) ^ (w[i - 15] >> 7);
            const uint64_t s1 = r64(w[i - 2], 19) ^ r64(w[i - 2], 61) ^ (w[i - 2] >> 6);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 80; ++i) {
            const uint64_t S1 = r64(e, 14) ^ r64(e, 18) ^ r64(e, 41), ch = (e & f) ^ (~e & g);
            const uint64_t t1 = hh + S1 + ch + K512[i] + w[i];
            const uint64_t S0 = r64(a, 28) ^ r64(a, 34) ^ r64(a, 39), mj = (a & b) ^ (a & c) ^ (b & c);
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + S0 + mj;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    for (int i = 0; i < 8; ++i)
        for (int b = 0; b < 8; ++b) out64[8 * i + b] = (uint8_t)(h[i] >> (56 - 8 * b));
}
// 512-bit little-endian integer mod L, bit-serial (r = 2r + bit; conditional subtract)
static __device__ void sc_reduce512(const uint8_t* in64, uint32_t* out) {
    uint32_t r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int bit = 511; bit >= 0; --bit) {
        uint32_t carry = (in64[bit >> 3] >> (bit & 7)) & 1;
        for (int i = 0; i < 8; ++i) {
            const uint32_t nc = r[i] >> 31;
            r[i] = (r[i] << 1) | carry;
            carry = nc;
        }
        if (u256_geq(r, SC_L)) u256_sub(r, r, SC_L);  // r < L < 2^253 before doubling: no carry out
    }
    for (int i = 0; i < 8; ++i) out[i] = r[i];
}

}  // namespace ed
