// Poseidon-Goldilocks permutation (width 12, x^7, 4 + 22 + 4 rounds) for gfx950.
// Replaces plonky2::hash::poseidon::Poseidon::poseidon (v0.2.0), the hash behind
// every Merkle cap and the Fiat-Shamir challenger of the reference's prover
// (AlgebraicHasher bound at circuits/header_range.rs:28-29).
//
// One permutation per lane, state in 24 VGPRs.  The MDS layer never forms
// 128-bit products: each state word is split into 32-bit halves, the circulant
// row sums of the halves stay below 2^42 and are computed by shifts and adds
// only (poseidon_mds_half), and the two sums are recombined with
// 2^64 = 2^32 - 1 (mod p).
#pragma once
#include "gl.cuh"
#include "poseidon_constants.h"

static __constant__ uint64_t POSEIDON_RC[360] = VX_POSEIDON_RC_INIT;

// x^7; intermediate products stay non-canonical (the multiplier and the MDS split accept any
// 64-bit representative), so no compare/select is spent between the four multiplications
__device__ __forceinline__ uint64_t poseidon_sbox(uint64_t x) {
    const uint64_t x2 = gl_mul_nc(x, x), x3 = gl_mul_nc(x2, x), x4 = gl_mul_nc(x2, x2);
    return gl_mul_nc(x3, x4);
}

// Circulant MDS on one 32-bit half of the state, in the "frequency domain" of the factor 4 of
// 12 = 3 x 4: y = x (*) d (cyclic convolution with the reversed first row).  With w = z^3 the
// product splits into three length-4 real FFTs (twiddles +-1, +-i: exact in integers), a 3x3
// block product per frequency (plain / (-i)-twisted / negacyclic for w = 1, -i, -1) and three
// inverse FFTs.  For this matrix every frequency-domain coefficient is +-2^k:
//   w = 1 : 64*(1, 2, 1)      w = -i : 2*((2,-1), (-4,1), (16,1))      w = -1 : 4*(-1, -8, 2)
// so the whole layer is shifts and adds on int64 -- no integer multiplier, which is quarter-rate
// on CDNA.  The factors 64 / 2 / 4 and the inverse FFT's 1/4 are folded:
//   y[3q+c] = 16*A0[c] + (-1)^q A2[c] + {R, -I, -R, I}[q].
__device__ __forceinline__ void poseidon_mds_half(const uint32_t* x, int64_t* y) {
    int64_t u0[3], u2[3], ur[3], ui[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int64_t x0 = x[a], x1 = x[3 + a], x2 = x[6 + a], x3 = x[9 + a];
        const int64_t A = x0 + x2, B = x1 + x3;
        u0[a] = A + B;
        u2[a] = A - B;
        ur[a] = x0 - x2;
        ui[a] = x3 - x1;  // U1 = (x0 - x2) - i (x1 - x3)
    }
    // w = 1: A0[c] = sum_a U0[a] * e0[(c - a) mod 3], e0 = (1, 2, 1)
    const int64_t s0 = u0[0] + u0[1] + u0[2];
    const int64_t a0[3] = {s0 + u0[2], s0 + u0[0], s0 + u0[1]};
    // w = -1: A2[c] = sum_a (+-) U2[a] * e2[(c - a) mod 3], e2 = (-1, -8, 2), wrapped terms negated
    const int64_t a2[3] = {-u2[0] - 2 * u2[1] + 8 * u2[2], -8 * u2[0] - u2[1] - 2 * u2[2], 2 * u2[0] - 8 * u2[1] - u2[2]};
    // w = -i: e1 = ((2,-1), (-4,1), (16,1)); wrapped terms are multiplied by -i: (re, im) -> (im, -re)
    int64_t p0r[3], p0i[3], p1r[3], p1i[3], p2r[3], p2i[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        p0r[a] = 2 * ur[a] + ui[a];   // U1[a] * (2 - i)
        p0i[a] = 2 * ui[a] - ur[a];
        p1r[a] = -4 * ur[a] - ui[a];  // U1[a] * (-4 + i)
        p1i[a] = ur[a] - 4 * ui[a];
        p2r[a] = 16 * ur[a] - ui[a];  // U1[a] * (16 + i)
        p2i[a] = 16 * ui[a] + ur[a];
    }
    // c = 0: a=0 e[0]; a=1 e[2] wrap; a=2 e[1] wrap
    const int64_t r0 = p0r[0] + p2i[1] + p1i[2], i0 = p0i[0] - p2r[1] - p1r[2];
    // c = 1: a=0 e[1]; a=1 e[0]; a=2 e[2] wrap
    const int64_t r1 = p1r[0] + p0r[1] + p2i[2], i1 = p1i[0] + p0i[1] - p2r[2];
    // c = 2: a=0 e[2]; a=1 e[1]; a=2 e[0]
    const int64_t r2 = p2r[0] + p1r[1] + p0r[2], i2 = p2i[0] + p1i[1] + p0i[2];
    const int64_t rr[3] = {r0, r1, r2}, ii[3] = {i0, i1, i2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int64_t t = 16 * a0[c], pp = t + a2[c], mm = t - a2[c];
        y[c] = pp + rr[c];
        y[3 + c] = mm - ii[c];
        y[6 + c] = pp - rr[c];
        y[9 + c] = mm + ii[c];
    }
}

// MDS layer + the NEXT round's constant add.  Each output is al + ah 2^32 with al, ah < 2^43 (row sums of the
// 32-bit halves, plus the constant's halves when RC):  ah 2^32 = hi(ah) 2^64 + lo(ah) 2^32, and 2^64 = eps, so
//   w = hi(ah) * eps + al          one v_mad_u64_u32, < 2^44, cannot wrap
//   y = w + lo(ah) 2^32            a 32-bit add into the high word; its carry is worth 2^64 = eps
// The result is left NON-canonical (any value in [0, 2^64)): the s-box multiplier and the next MDS split accept
// that; poseidon_permute canonicalises once at the end.  (v_mad_u64_u32 issues at the rate of any other VOP3
// instruction on gfx950, tools/isa_rate.hip, so a mad that replaces an add + compare + select is a clear win.)
template <bool RC>
__device__ __forceinline__ void poseidon_mds(uint64_t* s, int rc_next) {
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
    int64_t yl[12], yh[12];
    poseidon_mds_half(lo, yl);
    poseidon_mds_half(hi, yh);
    yl[0] += (int64_t)lo[0] * VX_POSEIDON_MDS_DIAG0;
    yh[0] += (int64_t)hi[0] * VX_POSEIDON_MDS_DIAG0;
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        uint64_t al = (uint64_t)yl[r], ah = (uint64_t)yh[r];
        if (RC) {
            const uint64_t c = POSEIDON_RC[rc_next + r];
            al += c & GL_EPS;
            ah += c >> 32;
        }
        const uint64_t w = (uint64_t)(uint32_t)(ah >> 32) * GL_EPS + al;
        uint32_t yhi;
        const bool carry = __builtin_add_overflow((uint32_t)(w >> 32), (uint32_t)ah, &yhi);
        const uint64_t y = ((uint64_t)yhi << 32) | (uint32_t)w;
        s[r] = y + (carry ? (uint64_t)GL_EPS : 0);  // y wrapped to < 2^44: no second carry
    }
}

__device__ __forceinline__ void poseidon_permute(uint64_t* s) {
    int rc = 12;  // constants of round k+1 are added by the MDS layer of round k
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl_add_nc(s[i], POSEIDON_RC[i]);
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox(s[i]);
        poseidon_mds<true>(s, rc);
        rc += 12;
    }
#pragma unroll 1
    for (int r = 0; r < 22; ++r) {
        s[0] = poseidon_sbox(s[0]);
        poseidon_mds<true>(s, rc);
        rc += 12;
    }
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox(s[i]);
        poseidon_mds<true>(s, rc);
        rc += 12;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox(s[i]);
    poseidon_mds<false>(s, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl_canon(s[i]);
}
