// Poseidon-Goldilocks permutation (width 12, x^7, 4 + 22 + 4 rounds) for gfx950.
// Replaces plonky2::hash::poseidon::Poseidon::poseidon (v0.2.0), the hash behind
// every Merkle cap and the Fiat-Shamir challenger of the reference's prover
// (AlgebraicHasher bound at circuits/header_range.rs:28-29).
//
// One permutation per lane, state in 24 VGPRs.  The MDS layer never forms
// 128-bit products: each state word is split into its low 52 and high 12 bits,
// the circulant row sums of the two parts (< 2^61 and < 2^21) are computed by
// shifts and adds only (poseidon_mds_part: 64-bit ops for the low part, plain
// 32-bit ops for the high part), and the two sums are recombined with
// 2^64 = 2^32 - 1 (mod p).
#pragma once
#include <utility>
#include "gl.cuh"
#include "poseidon_constants.h"

// folded form: partial rounds (4..25) carry a single constant, for element 0 (poseidon_constants.h)
static __constant__ uint64_t POSEIDON_RC[360] = VX_POSEIDON_RC_FOLDED_INIT;

// x^7; intermediate products stay non-canonical (the multiplier and the MDS split accept any
// 64-bit representative), so no compare/select is spent between the four multiplications
__device__ __forceinline__ uint64_t poseidon_sbox(uint64_t x) {
    const uint64_t x2 = gl_mul_nc(x, x), x3 = gl_mul_nc(x2, x), x4 = gl_mul_nc(x2, x2);
    return gl_mul_nc(x3, x4);
}

// Circulant MDS on one part (low 52 / high 12 bits) of the state, in the "frequency domain" of the factor 4 of
// 12 = 3 x 4: y = x (*) d (cyclic convolution with the reversed first row).  With w = z^3 the
// product splits into three length-4 real FFTs (twiddles +-1, +-i: exact in integers), a 3x3
// block product per frequency (plain / (-i)-twisted / negacyclic for w = 1, -i, -1) and three
// inverse FFTs.  For this matrix every frequency-domain coefficient is +-2^k:
//   w = 1 : 64*(1, 2, 1)      w = -i : 2*((2,-1), (-4,1), (16,1))      w = -1 : 4*(-1, -8, 2)
// so the whole layer is shifts and adds (83 of them per part instead of 144 multiply-adds).  The factors 64 / 2 / 4 and the inverse FFT's 1/4 are folded:
//   y[3q+c] = 16*A0[c] + (-1)^q A2[c] + {R, -I, -R, I}[q].
// T = accumulator type: int64_t (inputs < 2^52: every intermediate stays below 2^61) or int32_t (inputs < 2^12).
template <class T, class TIn>
__device__ __forceinline__ void poseidon_mds_part(const TIn* x, T* y) {
    T u0[3], u2[3], ur[3], ui[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const T x0 = (T)x[a], x1 = (T)x[3 + a], x2 = (T)x[6 + a], x3 = (T)x[9 + a];
        const T A = x0 + x2, B = x1 + x3;
        u0[a] = A + B;
        u2[a] = A - B;
        ur[a] = x0 - x2;
        ui[a] = x3 - x1;  // U1 = (x0 - x2) - i (x1 - x3)
    }
    // w = 1: A0[c] = sum_a U0[a] * e0[(c - a) mod 3], e0 = (1, 2, 1)
    const T s0 = u0[0] + u0[1] + u0[2];
    const T a0[3] = {s0 + u0[2], s0 + u0[0], s0 + u0[1]};
    // w = -1: A2[c] = sum_a (+-) U2[a] * e2[(c - a) mod 3], e2 = (-1, -8, 2), wrapped terms negated
    const T a2[3] = {-u2[0] - 2 * u2[1] + 8 * u2[2], -8 * u2[0] - u2[1] - 2 * u2[2], 2 * u2[0] - 8 * u2[1] - u2[2]};
    // w = -i: e1 = ((2,-1), (-4,1), (16,1)); wrapped terms are multiplied by -i: (re, im) -> (im, -re)
    T p0r[3], p0i[3], p1r[3], p1i[3], p2r[3], p2i[3];
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
    const T r0 = p0r[0] + p2i[1] + p1i[2], i0 = p0i[0] - p2r[1] - p1r[2];
    // c = 1: a=0 e[1]; a=1 e[0]; a=2 e[2] wrap
    const T r1 = p1r[0] + p0r[1] + p2i[2], i1 = p1i[0] + p0i[1] - p2r[2];
    // c = 2: a=0 e[2]; a=1 e[1]; a=2 e[0]
    const T r2 = p2r[0] + p1r[1] + p0r[2], i2 = p2i[0] + p1i[1] + p0i[2];
    const T rr[3] = {r0, r1, r2}, ii[3] = {i0, i1, i2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const T t = 16 * a0[c], pp = t + a2[c], mm = t - a2[c];
        y[c] = pp + rr[c];
        y[3 + c] = mm - ii[c];
        y[6 + c] = pp - rr[c];
        y[9 + c] = mm + ii[c];
    }
}

// MDS layer + the NEXT round's constant add.  State word = xl + xh 2^52 (xl < 2^52, xh < 2^12); each output is
// yl + yh 2^52 with yl < 2^61, yh < 2^21 (row sums of the parts, plus the constant's parts when RC):
//   yh 2^52 = (yh >> 12) 2^64 + (yh & 0xfff) 2^52,  and 2^64 = eps, so
//   w = (yh >> 12) * eps + yl        one v_mad_u64_u32, < 2^62, cannot wrap
//   y = w + ((yh << 20) mod 2^32) 2^32   a 32-bit add into the high word; its carry is worth 2^64 = eps
// The result is left NON-canonical (any value in [0, 2^64)): the s-box multiplier and the next MDS split accept
// that; poseidon_permute canonicalises once at the end.  (v_mad_u64_u32 issues at the rate of any other VOP3
// instruction on gfx950 and plain 32-bit add/sub/shift at almost twice that rate, tools/isa_rate.hip: the
// 52/12 split puts half of the layer on the cheap instructions and needs no zero-extension moves.)
// (Tried: a 55/9 split with the high part as packed 16-bit dot products, v_dot2_u32_u16, 84 instructions instead of
// ~140: the permutation went from 66.3 k to 70.6 k cycles -- the dot instructions do not issue at full rate.)
// RC: 0 = no constants, 1 = the next round's constant for element 0 only (a partial round follows), 12 = all twelve
#ifndef VX_POSEIDON_FP64
#define VX_POSEIDON_FP64 0
#endif
#if VX_POSEIDON_FP64
// (Tried, off: tools/ab_poseidon_fp64.sh, profiles/r03_ab_poseidon_fp64.txt -- same outputs, v_add_f64 / v_fma_f64 issue at 4.2 cycles as
// hoped, but the layer went from 1241 to 1298 cycles and the permutation from 61.7 k to 63.8 k: the 24 conversions and the longer
// dependent chains cost more than the 48 saved instructions.)
// The low part in EXACT double-precision arithmetic.  The layer's network grows its inputs by at most 2^8 (every intermediate is a
// linear form whose absolute coefficients sum to <= 256; + 8 x[0] for the diagonal), so with a 43 / 21 split the low part's
// intermediates stay below 2^52 -- integers a double holds exactly -- and the high part's below 2^30.  One v_add_f64 / v_fma_f64
// does what the 64-bit integer form needs a v_lshl_add_u64 (sums) or a two-instruction borrow chain (differences) plus a shift for:
// 78 instructions for the low part instead of 139.  In: (2^52 + lo43) is assembled as the bit pattern of a double (the mantissa IS the
// integer) and 2^52 subtracted; out: 2^52 (+ the round constant's low part) is added, which leaves the integer in the mantissa again.
struct PoseidonRcSplit {
    uint64_t d[360];  // bit pattern of the double 2^52 + (c mod 2^43)
    uint32_t h[360];  // c >> 43
};
constexpr PoseidonRcSplit poseidon_rc_split() {
    constexpr uint64_t rc[360] = VX_POSEIDON_RC_FOLDED_INIT;
    PoseidonRcSplit t{};
    for (int i = 0; i < 360; ++i) {
        t.d[i] = 0x4330000000000000ULL + (rc[i] & ((1ULL << 43) - 1));
        t.h[i] = (uint32_t)(rc[i] >> 43);
    }
    return t;
}
static __constant__ PoseidonRcSplit POSEIDON_RCT = poseidon_rc_split();
template <int RC>
__device__ __forceinline__ void poseidon_mds(uint64_t* s, int rc_next) {
    constexpr double TWO52 = 4503599627370496.0;
    double xl[12];
    uint32_t xh[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const uint32_t lo = (uint32_t)s[i], hi = (uint32_t)(s[i] >> 32);
        xl[i] = __longlong_as_double((long long)(((uint64_t)(0x43300000u | (hi & 0x7FFu)) << 32) | lo)) - TWO52;
        xh[i] = hi >> 11;
    }
    double yl[12];
    int32_t yh[12];
    poseidon_mds_part<double>(xl, yl);
    poseidon_mds_part<int32_t>(xh, yh);
    yl[0] += xl[0] * (double)VX_POSEIDON_MDS_DIAG0;
    yh[0] += (int32_t)(xh[0] * VX_POSEIDON_MDS_DIAG0);
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        double magic = TWO52;
        uint32_t ah = (uint32_t)yh[r];
        if (RC == 12 || (RC == 1 && r == 0)) {
            magic = __longlong_as_double((long long)POSEIDON_RCT.d[rc_next + r]);  // 2^52 + (c mod 2^43), a scalar operand
            ah += POSEIDON_RCT.h[rc_next + r];                                      // c >> 43
        }
        // 0 <= yl + (c & M43) < 2^52: the sum lands in [2^52, 2^53) and its low 52 mantissa bits are the integer
        const uint64_t al = (uint64_t)__double_as_longlong(yl[r] + magic) & ((1ULL << 52) - 1);
        const uint64_t w = (uint64_t)(ah >> 21) * GL_EPS + al;  // (ah >> 21) 2^64 = (ah >> 21) eps; < 2^53
        uint32_t yhi;
        const bool carry = __builtin_add_overflow((uint32_t)(w >> 32), ah << 11, &yhi);  // + (ah mod 2^21) 2^43
        const uint64_t y = ((uint64_t)yhi << 32) | (uint32_t)w;
        s[r] = y + (carry ? (uint64_t)GL_EPS : 0);  // y wrapped to < 2^53: no second carry
    }
}
#else
template <int RC>
__device__ __forceinline__ void poseidon_mds(uint64_t* s, int rc_next) {
    constexpr uint64_t M52 = (1ULL << 52) - 1;
    uint64_t xl[12];
    uint32_t xh[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        xl[i] = s[i] & M52;
        xh[i] = (uint32_t)(s[i] >> 52);
    }
    int64_t yl[12];
    int32_t yh[12];
    poseidon_mds_part<int64_t>(xl, yl);
    poseidon_mds_part<int32_t>(xh, yh);
    yl[0] += (int64_t)(xl[0] * VX_POSEIDON_MDS_DIAG0);
    yh[0] += (int32_t)(xh[0] * VX_POSEIDON_MDS_DIAG0);
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        uint64_t al = (uint64_t)yl[r];
        uint32_t ah = (uint32_t)yh[r];
        if (RC == 12 || (RC == 1 && r == 0)) {
            const uint64_t c = POSEIDON_RC[rc_next + r];
            al += c & M52;
            ah += (uint32_t)(c >> 52);
        }
        const uint64_t w = (uint64_t)(ah >> 12) * GL_EPS + al;
        uint32_t yhi;
        const bool carry = __builtin_add_overflow((uint32_t)(w >> 32), ah << 20, &yhi);
        const uint64_t y = ((uint64_t)yhi << 32) | (uint32_t)w;
        s[r] = y + (carry ? (uint64_t)GL_EPS : 0);  // y wrapped to < 2^62: no second carry
    }
}

#endif

// ---- partial rounds on a SPLIT state (VX_POSEIDON_SPLIT_PARTIAL, default; tools/ab_poseidon_split.sh: permutation 62.3 k -> 58.8 k
// cycles per wave, 2.60 -> 2.75 G permutations/s, header_range_256 8.59 -> 8.84 proofs/s, same outputs): in a partial round only element 0 meets the s-box, so only
// element 0 has to be a 64-bit word; the other eleven go from one linear layer straight into the next and stay in the layer's
// own representation, low part + high part 2^52.  Between two layers they are merely NORMALISED -- the bits of the low part
// above 2^52 move to the high part, the bits of the high part above 2^12 come back as multiples of 2^64 = 2^32 - 1 -- six
// instructions (one of them a multiply-add) instead of the nine of a full recombination and a fresh split.
//   in : xl[i] < 2^52 + 2^42, xh[i] < 2^12  (i >= 1);   element 0 enters as the 64-bit word s0 (already through the s-box)
//   out: the same for i >= 1;  returns the next s0 = y[0] + the round constant of element 0
// Bounds: the layer grows its inputs by at most 264 (poseidon_mds_part + the diagonal): yl < 2^61, yh < 2^21; t = yl >> 52 < 2^9,
// yh + t < 2^21 + 2^9, u = (yh + t) >> 12 < 2^9 + 1, u (2^32 - 1) < 2^42.
#ifndef VX_POSEIDON_SPLIT_PARTIAL
#define VX_POSEIDON_SPLIT_PARTIAL 1
#endif
__device__ __forceinline__ uint64_t poseidon_mds_split(uint64_t s0, uint64_t* xl, uint32_t* xh, int rc_next) {
    constexpr uint64_t M52 = (1ULL << 52) - 1;
    xl[0] = s0 & M52;
    xh[0] = (uint32_t)(s0 >> 52);
    int64_t yl[12];
    int32_t yh[12];
    poseidon_mds_part<int64_t>(xl, yl);
    poseidon_mds_part<int32_t>(xh, yh);
    yl[0] += (int64_t)(xl[0] * VX_POSEIDON_MDS_DIAG0);
    yh[0] += (int32_t)(xh[0] * VX_POSEIDON_MDS_DIAG0);
#pragma unroll
    for (int r = 1; r < 12; ++r) {
        const uint64_t al = (uint64_t)yl[r];
        const uint32_t hh = (uint32_t)yh[r] + (uint32_t)(al >> 52);
        xh[r] = hh & 0xFFFu;
        xl[r] = (uint64_t)(hh >> 12) * GL_EPS + (al & M52);
    }
    const uint64_t c = POSEIDON_RC[rc_next];
    const uint64_t al = (uint64_t)yl[0] + (c & M52);
    const uint32_t ah = (uint32_t)yh[0] + (uint32_t)(c >> 52);
    const uint64_t w = (uint64_t)(ah >> 12) * GL_EPS + al;
    uint32_t yhi;
    const bool carry = __builtin_add_overflow((uint32_t)(w >> 32), ah << 20, &yhi);
    const uint64_t y = ((uint64_t)yhi << 32) | (uint32_t)w;
    return y + (carry ? (uint64_t)GL_EPS : 0);
}
// the layer that ends the partial rounds: split state in, twelve 64-bit words out, all twelve constants of the next (full) round added
__device__ __forceinline__ void poseidon_mds_split_out(uint64_t s0, uint64_t* xl, uint32_t* xh, uint64_t* s, int rc_next) {
    constexpr uint64_t M52 = (1ULL << 52) - 1;
    xl[0] = s0 & M52;
    xh[0] = (uint32_t)(s0 >> 52);
    int64_t yl[12];
    int32_t yh[12];
    poseidon_mds_part<int64_t>(xl, yl);
    poseidon_mds_part<int32_t>(xh, yh);
    yl[0] += (int64_t)(xl[0] * VX_POSEIDON_MDS_DIAG0);
    yh[0] += (int32_t)(xh[0] * VX_POSEIDON_MDS_DIAG0);
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        const uint64_t c = POSEIDON_RC[rc_next + r];
        const uint64_t al = (uint64_t)yl[r] + (c & M52);
        const uint32_t ah = (uint32_t)yh[r] + (uint32_t)(c >> 52);
        const uint64_t w = (uint64_t)(ah >> 12) * GL_EPS + al;
        uint32_t yhi;
        const bool carry = __builtin_add_overflow((uint32_t)(w >> 32), ah << 20, &yhi);
        const uint64_t y = ((uint64_t)yhi << 32) | (uint32_t)w;
        s[r] = y + (carry ? (uint64_t)GL_EPS : 0);
    }
}

__device__ __forceinline__ void poseidon_permute(uint64_t* s) {
    int rc = 12;  // constants of round k+1 are added by the MDS layer of round k
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl_add_nc(s[i], POSEIDON_RC[i]);
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox(s[i]);
        poseidon_mds<12>(s, rc);
        rc += 12;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox(s[i]);
    poseidon_mds<1>(s, rc);  // round 4 is partial
    rc += 12;
#if VX_POSEIDON_SPLIT_PARTIAL && !VX_POSEIDON_FP64
    {
        constexpr uint64_t M52 = (1ULL << 52) - 1;
        uint64_t xl[12];
        uint32_t xh[12];
#pragma unroll
        for (int i = 1; i < 12; ++i) {
            xl[i] = s[i] & M52;
            xh[i] = (uint32_t)(s[i] >> 52);
        }
        uint64_t s0 = s[0];
#pragma unroll 1
        for (int r = 0; r < 21; ++r) {
            s0 = poseidon_mds_split(poseidon_sbox(s0), xl, xh, rc);
            rc += 12;
        }
        poseidon_mds_split_out(poseidon_sbox(s0), xl, xh, s, rc);  // round 26 is full again
        rc += 12;
    }
#else
#pragma unroll 1
    for (int r = 0; r < 21; ++r) {
        s[0] = poseidon_sbox(s[0]);
        poseidon_mds<1>(s, rc);
        rc += 12;
    }
    s[0] = poseidon_sbox(s[0]);
    poseidon_mds<12>(s, rc);  // round 26 is full again (its constants absorbed what the partial rounds pushed forward)
    rc += 12;
#endif
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox(s[i]);
        poseidon_mds<12>(s, rc);
        rc += 12;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox(s[i]);
    poseidon_mds<0>(s, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl_canon(s[i]);
}

// ---- cooperative permutation: one state element per lane, 16-lane groups (lanes 12..15 idle) -------------------
// For small trees (few leaves, upper Merkle levels, FRI tails) the one-lane-per-permutation kernels leave the chip
// empty and each permutation is a 15 k-instruction dependency chain (47 us for a lone wave).  Here the twelve
// s-boxes of a full round run in twelve lanes and the MDS row of lane r is a dot product over the group's state,
// exchanged through LDS: a shorter chain per permutation.  `g` = the group's 12 LDS slots.  A group never spans two
// waves and LDS operations of one wave execute in order, so the exchange needs no block barrier -- only that the
// compiler keeps the write before the reads and the reads before the next write (wave-scope fences).
__device__ __forceinline__ uint64_t poseidon_mds_coop(uint64_t s, int l, uint64_t* g) {
    constexpr uint32_t C[12] = VX_POSEIDON_MDS_CIRC_INIT;
    if (l < 12) g[l] = s;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint64_t al = 0, ah = 0;  // sum C_i lo32(x_i), sum C_i hi32(x_i): both < 2^41
    const int r = l < 12 ? l : 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        int j = r + i;
        if (j >= 12) j -= 12;
        const uint64_t x = g[j];
        al += (uint64_t)(uint32_t)x * C[i];
        ah += (x >> 32) * C[i];
    }
    if (l == 0) {
        const uint64_t x0 = g[0];
        al += (uint64_t)(uint32_t)x0 * VX_POSEIDON_MDS_DIAG0;
        ah += (x0 >> 32) * VX_POSEIDON_MDS_DIAG0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // all reads issued before the next layer overwrites the slots
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // y = al + ah 2^32 = al + hi32(ah) 2^64 + lo32(ah) 2^32, 2^64 = eps
    const uint64_t w = (ah >> 32) * GL_EPS + al;  // < 2^42
    uint32_t yhi;
    const bool carry = __builtin_add_overflow((uint32_t)(w >> 32), (uint32_t)ah, &yhi);
    const uint64_t y = ((uint64_t)yhi << 32) | (uint32_t)w;
    return y + (carry ? (uint64_t)GL_EPS : 0);
}
// ---- the same layer with wavefront shuffles instead of LDS (north_star's "wavefront shuffles for the Poseidon MDS"): the
// 15 other lanes of the 16-lane row come in through DPP row rotations (v_mov_b32 ... row_ror:n, two per 64-bit element).
// Each lane multiplies what rotation n brings by a coefficient fixed for the kernel: C[(q - l) mod 12] when the source lane
// q holds a state element, 0 when it is one of the four idle lanes -- found by rotating the lane numbers themselves, so the
// code does not depend on the direction row_ror turns.  Default since the A/B in profiles/README.md (3-9 % on small trees, caps
// identical); -DVX_POSEIDON_COOP_DPP=0 brings the LDS exchange back.
#ifndef VX_POSEIDON_COOP_DPP
#define VX_POSEIDON_COOP_DPP 1
#endif
template <int N>
__device__ __forceinline__ uint32_t dpp_ror(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x120 + N, 0xf, 0xf, false);
}
struct CoopCoef {
    uint32_t c[16];
};
static __device__ const uint32_t POSEIDON_MDS_CIRC_DEV[12] = VX_POSEIDON_MDS_CIRC_INIT;
template <int N>
__device__ __forceinline__ void coop_coef_one(CoopCoef& k, int l) {
    const int q = (int)dpp_ror<N>((uint32_t)l);
    int d = q - l;
    if (d < 0) d += 12;
    k.c[N] = (q < 12 && l < 12) ? POSEIDON_MDS_CIRC_DEV[d < 12 ? d : 0] : 0;
}
template <int... Ns>
__device__ __forceinline__ void coop_coef_all(CoopCoef& k, int l, std::integer_sequence<int, Ns...>) {
    (coop_coef_one<Ns + 1>(k, l), ...);
}
template <int N>
__device__ __forceinline__ void coop_mac_one(const CoopCoef& k, uint64_t s, uint64_t& al, uint64_t& ah) {
    const uint32_t lo = dpp_ror<N>((uint32_t)s), hi = dpp_ror<N>((uint32_t)(s >> 32));
    al += (uint64_t)lo * k.c[N];
    ah += (uint64_t)hi * k.c[N];
}
template <int... Ns>
__device__ __forceinline__ void coop_mac_all(const CoopCoef& k, uint64_t s, uint64_t& al, uint64_t& ah, std::integer_sequence<int, Ns...>) {
    (coop_mac_one<Ns + 1>(k, s, al, ah), ...);
}
__device__ __forceinline__ uint64_t poseidon_mds_coop_dpp(uint64_t s, int l, const CoopCoef& k) {
    uint64_t al = (uint64_t)(uint32_t)s * k.c[0], ah = (s >> 32) * k.c[0];
    coop_mac_all(k, s, al, ah, std::make_integer_sequence<int, 15>{});
    if (l == 0) {
        al += (uint64_t)(uint32_t)s * VX_POSEIDON_MDS_DIAG0;
        ah += (s >> 32) * VX_POSEIDON_MDS_DIAG0;
    }
    const uint64_t w = (ah >> 32) * GL_EPS + al;
    uint32_t yhi;
    const bool carry = __builtin_add_overflow((uint32_t)(w >> 32), (uint32_t)ah, &yhi);
    const uint64_t y = ((uint64_t)yhi << 32) | (uint32_t)w;
    return y + (carry ? (uint64_t)GL_EPS : 0);
}
// s = this lane's state element (l < 12); returns the permuted element, canonical
__device__ __forceinline__ uint64_t poseidon_permute_coop(uint64_t s, int l, uint64_t* g) {
    const int lc = l < 12 ? l : 0;
#if VX_POSEIDON_COOP_DPP
    CoopCoef k;
    k.c[0] = l < 12 ? POSEIDON_MDS_CIRC_DEV[0] : 0;
    coop_coef_all(k, l, std::make_integer_sequence<int, 15>{});
    if (l >= 12) s = 0;
#pragma unroll 1
    for (int r = 0; r < 30; ++r) {
        s = gl_add_nc(s, POSEIDON_RC[12 * r + lc]);
        const bool full = r < 4 || r >= 26;
        if (full || l == 0) s = poseidon_sbox(s);
        s = poseidon_mds_coop_dpp(s, l, k);
    }
    return gl_canon(s);
#endif
#pragma unroll 1
    for (int r = 0; r < 30; ++r) {
        s = gl_add_nc(s, POSEIDON_RC[12 * r + lc]);  // folded table: partial rounds carry a constant for element 0 only
        const bool full = r < 4 || r >= 26;
        if (full || l == 0) s = poseidon_sbox(s);
        s = poseidon_mds_coop(s, l, g);
    }
    return gl_canon(s);
}
