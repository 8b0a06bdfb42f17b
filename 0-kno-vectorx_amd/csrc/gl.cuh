// Goldilocks field (p = 2^64 - 2^32 + 1) and quadratic extension F_p[X]/(X^2-7)
// device arithmetic for gfx950.  K1 of SURVEY.md section 2.2: what plonky2_field
// (v0.2.0, Cargo.lock:4871-4873; used by the reference at header.rs:47,
// subchain_verification.rs:448) provides on the CPU.
//
// CDNA4 has no 64x64->128 multiply: the product is built from 32-bit
// v_mul_lo/v_mul_hi/v_mad_u64_u32 (the compiler's expansion of __umul64hi) and
// reduced with 2^64 = 2^32 - 1, 2^96 = -1 (mod p).  Every function takes and
// returns CANONICAL values (< p) unless its name says otherwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL

struct gl2 {
    uint64_t a, b;  // a + b X
};

__device__ __forceinline__ uint64_t gl_canon(uint64_t x) { return x >= GL_P ? x - GL_P : x; }

// a + b: the sum exceeds p exactly when a + b overflows 64 bits or (a + b) + eps does, so the two
// carry flags decide (no 64-bit compares).
__device__ __forceinline__ uint64_t gl_add(uint64_t a, uint64_t b) {
    uint64_t s, t;
    const bool c1 = __builtin_add_overflow(a, b, &s);
    const bool c2 = __builtin_add_overflow(s, (uint64_t)GL_EPS, &t);  // s - p (mod 2^64)
    return (c1 | c2) ? t : s;
}
__device__ __forceinline__ uint64_t gl_sub(uint64_t a, uint64_t b) {
    uint64_t d;
    const bool bw = __builtin_sub_overflow(a, b, &d);
    return d - (bw ? (uint64_t)GL_EPS : 0);  // + p (mod 2^64)
}
// a + b where b is CANONICAL and a is any 64-bit representative; result in [0, 2^64), not canonical.
// (a + b < 2^64 + p, so after one wrap the value is below p and the + eps cannot wrap again.)
__device__ __forceinline__ uint64_t gl_add_nc(uint64_t a, uint64_t b) {
    uint64_t s;
    const bool c = __builtin_add_overflow(a, b, &s);
    return s + (c ? (uint64_t)GL_EPS : 0);
}
__device__ __forceinline__ uint64_t gl_neg(uint64_t a) { return a ? GL_P - a : 0; }
__device__ __forceinline__ uint64_t gl_dbl(uint64_t a) { return gl_add(a, a); }

// (hi, lo) 128-bit -> [0, 2^64), NOT necessarily canonical (may be in [p, 2^64)).
// value = lo + hl*2^64 + hh*2^96 = lo - hh + hl*(2^32-1)  (mod p).
#ifndef GL_REDUCE_C
// Hand-scheduled: 8 vector instructions instead of the 12 the compiler needs for the two conditional +-eps
// corrections (tools/isa_rate.hip: v_mad_u64_u32 issues at nearly the rate of any VOP3 add, so instruction COUNT
// is what the integer kernels pay for).  A = lo - hh wraps with borrow b; T = A + hl*eps wraps with carry c
// (the mad's own carry-out); V = T + (c-b)*2^64, and 2^64 = eps, so R = T + d*eps with d = c-b in {-1,0,1},
// applied as T - d (v_mad_i64_i32 by -1) and d added to the high word.  |d| = 1 cannot wrap again: c=1,b=0
// means T <= 2^64 - 2^33; c=0,b=1 means T >= 2^64 - 2^32 + 1.  gfx90a+ needs 2 wait states between a VALU
// write of VCC / an SGPR and a VALU read of it; inside an asm block that is ours to honour (s_nop 1).
__device__ __forceinline__ uint64_t gl_reduce128_nc(uint64_t hi, uint64_t lo) {
    const uint32_t hh = (uint32_t)(hi >> 32), hl = (uint32_t)hi;
    uint32_t al, ah, mb, mc;
    asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %4, vcc\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32 %2, vcc, %3, %3, vcc"
        : "=&v"(al), "=&v"(ah), "=&v"(mb)
        : "v"((uint32_t)lo), "v"((uint32_t)(lo >> 32)), "v"(hh)
        : "vcc");
    const uint64_t a = ((uint64_t)ah << 32) | al;
    uint64_t t, sc;
    asm("v_mad_u64_u32 %0, %1, %3, -1, %4\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, %1"
        : "=&v"(t), "=&s"(sc), "=&v"(mc)
        : "v"(hl), "v"(a));
    const uint32_t d = mb - mc;  // c - b
    uint64_t r;
    asm("v_mad_i64_i32 %0, vcc, %1, -1, %2" : "=v"(r) : "v"(d), "v"(t) : "vcc");
    return r + ((uint64_t)d << 32);
}
#else
__device__ __forceinline__ uint64_t gl_reduce128_nc(uint64_t hi, uint64_t lo) {
    const uint32_t hh = (uint32_t)(hi >> 32), hl = (uint32_t)hi;
    uint64_t t0 = lo - hh;
    if (lo < hh) t0 -= GL_EPS;                    // borrow: + p (mod 2^64)
    const uint64_t t1 = ((uint64_t)hl << 32) - hl;  // hl * (2^32 - 1); LLVM emits it as a v_mad_u64_u32 by -1
    uint64_t t2 = t0 + t1;
    if (t2 < t1) t2 += GL_EPS;                    // carry: - p (mod 2^64)
    return t2;
}
#endif
__device__ __forceinline__ uint64_t gl_reduce128(uint64_t hi, uint64_t lo) { return gl_canon(gl_reduce128_nc(hi, lo)); }
// 64 x 64 -> 128 from four 32 x 32 + 64 multiply-adds (v_mad_u64_u32); the compiler's own
// expansion of a * b and __umul64hi(a, b) computes the partial products twice.
#ifndef GL_MUL_C
// rows: t0 = a0 b0; t1 = a0 b1 + hi(t0) (cannot overflow); t2 = a1 b0 + t1 with the mad's carry-out k (65 bits);
// lo = (lo(t0), lo(t2)); hi = a1 b1 + hi(t2) + k 2^32.  Taking k from the instruction saves the extra
// 64-bit add and two register-pair moves of the C form below.
__device__ __forceinline__ void gl_mul128(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t0 = (uint64_t)a0 * b0;
    const uint64_t t1 = (uint64_t)a0 * b1 + (t0 >> 32);
    uint64_t t2, sk;
    uint32_t k;
    asm("v_mad_u64_u32 %0, %1, %3, %4, %5\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %2, 0, 1, %1"
        : "=&v"(t2), "=&s"(sk), "=&v"(k)
        : "v"(a1), "v"(b0), "v"(t1));
    hi = (uint64_t)a1 * b1 + (((uint64_t)k << 32) | (t2 >> 32));
    lo = (t2 << 32) | (uint32_t)t0;
}
#else
__device__ __forceinline__ void gl_mul128(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t0 = (uint64_t)a0 * b0;
    const uint64_t t1 = (uint64_t)a0 * b1 + (t0 >> 32);
    const uint64_t t2 = (uint64_t)a1 * b0 + (uint32_t)t1;
    hi = (uint64_t)a1 * b1 + (t1 >> 32) + (t2 >> 32);
    lo = (t2 << 32) | (uint32_t)t0;
}
#endif
// ---- lazy dot products: sum_k c_k * p_k kept as a 160-bit integer, reduced once --------------------------------
// (the random-linear-combination accumulators of the quotient kernel: one 64x64 multiply + a 5-word carry chain
// per term instead of multiply + reduce + modular add)
struct gl_acc {
    uint64_t lo, hi;
    uint32_t top;
};
__device__ __forceinline__ void gl_acc_zero(gl_acc& a) { a.lo = a.hi = 0, a.top = 0; }
// a += c * p (any 64-bit representatives); at most 2^32 terms
__device__ __forceinline__ void gl_mac(gl_acc& a, uint64_t c, uint64_t p) {
    uint64_t ph, pl;
    gl_mul128(c, p, ph, pl);
    const bool c1 = __builtin_add_overflow(a.lo, pl, &a.lo);
    const bool c2 = __builtin_add_overflow(a.hi, ph, &a.hi);
    const bool c3 = __builtin_add_overflow(a.hi, (uint64_t)c1, &a.hi);
    a.top += (uint32_t)c2 + (uint32_t)c3;
}
// lo + hi 2^64 + top 2^128 (mod p), canonical.  2^128 = eps^2 = 2^64 - 2^33 + 1 = -2^32 (mod p).
__device__ __forceinline__ uint64_t gl_reduce128(uint64_t hi, uint64_t lo);
__device__ __forceinline__ uint64_t gl_acc_reduce(const gl_acc& a) {
    const uint64_t r = gl_reduce128(a.hi, a.lo);
    return gl_sub(r, (uint64_t)a.top << 32);  // top < 2^31: top 2^32 < p
}

// inputs: any 64-bit representatives; output in [0, 2^64) (non-canonical)
__device__ __forceinline__ uint64_t gl_mul_nc(uint64_t a, uint64_t b) {
    uint64_t hi, lo;
    gl_mul128(a, b, hi, lo);
    return gl_reduce128_nc(hi, lo);
}
__device__ __forceinline__ uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_canon(gl_mul_nc(a, b)); }
__device__ __forceinline__ uint64_t gl_sqr(uint64_t a) { return gl_mul(a, a); }
// multiply by a small constant c < 2^32
__device__ __forceinline__ uint64_t gl_mul_small(uint64_t a, uint32_t c) {
    uint64_t lo = (a & GL_EPS) * c;  // < 2^64
    uint64_t hi = (a >> 32) * c;     // < 2^64, weight 2^32
    // a*c = lo + hi*2^32 ; hi*2^32 = (hi_lo << 32) + hi_hi * 2^64
    uint64_t hi_lo = hi & GL_EPS, hi_hi = hi >> 32;
    uint64_t r = gl_canon(lo);
    r = gl_add(r, gl_canon(hi_lo << 32));
    r = gl_add(r, hi_hi * GL_EPS);  // hi_hi < 2^32 so product < p
    return r;
}
__device__ __forceinline__ uint64_t gl_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, a);
        a = gl_sqr(a);
        e >>= 1;
    }
    return r;
}
// a^(p-2) by an addition chain (p - 2 = (2^32 - 2) 2^32 + (2^32 - 1)): 65 squarings + 10 multiplications instead of the
// 64 + 63 of square-and-multiply -- the exponent is all ones.  Intermediates stay non-canonical.
__device__ __forceinline__ uint64_t gl_sqr_n(uint64_t x, int n) {
#pragma unroll 1
    for (int i = 0; i < n; ++i) x = gl_mul_nc(x, x);
    return x;
}
// (not inlined: a kernel that inverts at several places -- the FRI combination does 8 times per lane -- would otherwise
// carry eight copies of the chain and spill)
__device__ __noinline__ static uint64_t gl_inv(uint64_t a) {
    const uint64_t x2 = gl_mul_nc(gl_mul_nc(a, a), a);       // a^(2^2 - 1)
    const uint64_t x4 = gl_mul_nc(gl_sqr_n(x2, 2), x2);      // 2^4 - 1
    const uint64_t x6 = gl_mul_nc(gl_sqr_n(x4, 2), x2);      // 2^6 - 1
    const uint64_t x8 = gl_mul_nc(gl_sqr_n(x4, 4), x4);      // 2^8 - 1
    const uint64_t x16 = gl_mul_nc(gl_sqr_n(x8, 8), x8);     // 2^16 - 1
    const uint64_t x24 = gl_mul_nc(gl_sqr_n(x16, 8), x8);    // 2^24 - 1
    const uint64_t x30 = gl_mul_nc(gl_sqr_n(x24, 6), x6);    // 2^30 - 1
    const uint64_t x31 = gl_mul_nc(gl_mul_nc(x30, x30), a);  // 2^31 - 1
    const uint64_t t2 = gl_mul_nc(x31, x31);                 // a^(2^32 - 2)
    return gl_mul(gl_sqr_n(t2, 32), gl_mul_nc(t2, a));       // (a^(2^32-2))^(2^32) * a^(2^32-1)
}

__device__ __forceinline__ gl2 gl2_add(gl2 x, gl2 y) { return {gl_add(x.a, y.a), gl_add(x.b, y.b)}; }
__device__ __forceinline__ gl2 gl2_sub(gl2 x, gl2 y) { return {gl_sub(x.a, y.a), gl_sub(x.b, y.b)}; }
__device__ __forceinline__ gl2 gl2_mul(gl2 x, gl2 y) {
    uint64_t bb = gl_mul(x.b, y.b);
    return {gl_add(gl_mul(x.a, y.a), gl_mul_small(bb, 7)), gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a))};
}
__device__ __forceinline__ gl2 gl2_scale(gl2 x, uint64_t s) { return {gl_mul(x.a, s), gl_mul(x.b, s)}; }
__device__ __forceinline__ gl2 gl2_inv(gl2 x) {
    uint64_t n = gl_sub(gl_sqr(x.a), gl_mul_small(gl_sqr(x.b), 7));
    uint64_t ni = gl_inv(n);
    return {gl_mul(x.a, ni), gl_mul(gl_neg(x.b), ni)};
}

__device__ __forceinline__ uint32_t brev32(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }
