// Lazy Goldilocks arithmetic for the radix-16 NTT rounds: a value is the INTEGER  V = lo + hi 2^32 + k 2^64  with a small
// signed third word k, congruent mod p = 2^64 - 2^32 + 1 to the field element it stands for.  An addition or subtraction is
// a three-word carry chain (v_add_co / v_addc_co / v_addc_co) with no reduction at all -- a canonical 64-bit modular add
// costs six VALU instructions on gfx950, a subtract five -- and only the shift-twiddles (V 2^S, the powers of w_16 = 2^12)
// and the end of a round fold the third word back with  2^64 = 2^32 - 1,  2^96 = -1  (mod p).
//
// Plain C++ (no device intrinsics besides the two below, which have portable fallbacks) so that tests/ can compile the very
// same code for the host and compare it with big-integer arithmetic (tests/test_gl96_host.py).
#pragma once
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define GL96_FN __device__ __forceinline__
#elif defined(__HIPCC__)
#define GL96_FN __host__ __device__ inline
#else
#define GL96_FN static inline
#endif

namespace gl96 {

struct X {
    uint32_t lo, hi;
    int32_t k;
};

GL96_FN uint32_t addc(uint32_t a, uint32_t b, uint32_t cin, uint32_t* cout) {
#if defined(__clang__)
    return __builtin_addc(a, b, cin, cout);
#else
    const uint64_t s = (uint64_t)a + b + cin;
    *cout = (uint32_t)(s >> 32);
    return (uint32_t)s;
#endif
}
GL96_FN uint32_t subc(uint32_t a, uint32_t b, uint32_t bin, uint32_t* bout) {
#if defined(__clang__)
    return __builtin_subc(a, b, bin, bout);
#else
    const uint64_t d = (uint64_t)a - b - bin;
    *bout = (uint32_t)(d >> 63);
    return (uint32_t)d;
#endif
}
GL96_FN uint32_t alignbit(uint32_t hi, uint32_t lo, int sh) {  // ((hi:lo) >> sh) & 0xFFFFFFFF, 0 < sh < 32
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
}

GL96_FN X from64(uint64_t v) { return X{(uint32_t)v, (uint32_t)(v >> 32), 0}; }

GL96_FN X add(const X& a, const X& b) {
    X r;
    uint32_t c0, c1;
    r.lo = addc(a.lo, b.lo, 0u, &c0);
    r.hi = addc(a.hi, b.hi, c0, &c1);
    r.k = (int32_t)((uint32_t)a.k + (uint32_t)b.k + c1);
    return r;
}
GL96_FN X sub(const X& a, const X& b) {
    X r;
    uint32_t b0, b1;
    r.lo = subc(a.lo, b.lo, 0u, &b0);
    r.hi = subc(a.hi, b.hi, b0, &b1);
    r.k = (int32_t)((uint32_t)a.k - (uint32_t)b.k - b1);
    return r;
}

// (w0, w1, w2) - sext96(s):  three-word subtraction of a sign-extended 32-bit value
GL96_FN X sub_sext(uint32_t w0, uint32_t w1, uint32_t w2, int32_t s) {
    const uint32_t ss = (uint32_t)(s >> 31);
    X r;
    uint32_t b0, b1;
    r.lo = subc(w0, (uint32_t)s, 0u, &b0);
    r.hi = subc(w1, ss, b0, &b1);
    r.k = (int32_t)(w2 - ss - b1);
    return r;
}

// V 2^S (mod p) for 0 < S < 96, S not a multiple of 32; |a.k| 2^(S mod 32) must stay below 2^30 (the rounds keep |k| <= 7,
// and S mod 32 = 28 occurs only where |k| <= 6).  The result has |k| <= 2.
template <int S>
GL96_FN X shl(const X& a) {
    static_assert(S > 0 && S < 96 && (S & 31) != 0, "shift");
    constexpr int t = S & 31, q = S >> 5;
#ifdef GL96_TRACK_SHIFT
    GL96_TRACK_SHIFT(a.k, t);
#endif
    const uint32_t w0 = a.lo << t, w1 = alignbit(a.hi, a.lo, 32 - t);
    const int32_t w2 = (int32_t)((a.hi >> (32 - t)) + ((uint32_t)a.k << t));  // signed: the third word of V 2^t
    if (q == 0) {
        // w0 + w1 2^32 + w2 2^64,  w2 2^64 = w2 2^32 - w2
        X r = sub_sext(w0, w1, 0u, w2);
        uint32_t c;
        r.hi = addc(r.hi, (uint32_t)w2, 0u, &c);
        r.k = (int32_t)((uint32_t)r.k + (uint32_t)(w2 >> 31) + c);
        return r;
    } else if (q == 1) {
        // (V 2^t) 2^32 = w0 2^32 + w1 2^64 + w2 2^96 = w0 2^32 + w1 (2^32 - 1) - w2
        const uint64_t prod = (uint64_t)w1 * 0xFFFFFFFFu, base = (uint64_t)w0 << 32;
        uint64_t m;
        const uint32_t cm = __builtin_add_overflow(prod, base, &m) ? 1u : 0u;
        return sub_sext((uint32_t)m, (uint32_t)(m >> 32), cm, w2);
    } else {
        // (V 2^t) 2^64 = w0 2^64 + w1 2^96 + w2 2^128 = w0 (2^32 - 1) - w1 - w2 2^32
        const uint64_t m = (uint64_t)w0 * 0xFFFFFFFFu;
        X r;
        uint32_t b0, b1;
        r.lo = subc((uint32_t)m, w1, 0u, &b0);
        r.hi = subc((uint32_t)(m >> 32), (uint32_t)w2, b0, &b1);
        r.k = (int32_t)(0u - (uint32_t)(w2 >> 31) - b1);
        return r;
    }
}

// ---- back to 64 bits.  fold_fast is exact whenever neither word wraps; fold_ok says whether that held (the caller
// takes fold_exact for the whole round otherwise -- the halves of v must come within |k| of 0 or 2^32 for that).
constexpr uint32_t FOLD_K = 64;  // bound on |k| at the end of a round (asserted by the host test)
GL96_FN uint64_t fold_fast(const X& a) {  // lo - k + (hi + k) 2^32
    return ((uint64_t)(a.hi + (uint32_t)a.k) << 32) | (uint32_t)(a.lo - (uint32_t)a.k);
}
// running minimum of the distance of both halves from the wrap points (start from 0xFFFFFFFF)
GL96_FN uint32_t fold_margin(uint32_t m, const X& a) {
    const uint32_t zl = a.lo + FOLD_K, zh = a.hi + FOLD_K;
    const uint32_t z = zl < zh ? zl : zh;
    return m < z ? m : z;
}
GL96_FN bool fold_ok(uint32_t m) { return m >= 2 * FOLD_K; }
// exact, any |k| < 2^24: canonical result
GL96_FN uint64_t fold_exact(const X& a) {
    constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
    // W = V + Q p >= 0 with Q = 2^24:  lo + hi 2^32 + (k + Q) 2^64 - Q 2^32 + Q
    const uint32_t Q = 1u << 24;
    const unsigned __int128 W = ((unsigned __int128)(uint32_t)(a.k + (int32_t)Q) << 64) + (((uint64_t)a.hi << 32) | a.lo) - ((uint64_t)Q << 32) + Q;
    const uint64_t wl = (uint64_t)W, wh = (uint64_t)(W >> 64);  // wh < 2^26
    // wl + wh 2^64 = wl + wh (2^32 - 1)
    unsigned __int128 t = (unsigned __int128)wl + (unsigned __int128)wh * 0xFFFFFFFFu;
    uint64_t r = (uint64_t)t;
    const uint64_t top = (uint64_t)(t >> 64);  // 0 or 1
    unsigned __int128 u = (unsigned __int128)r + top * 0xFFFFFFFFULL;
    r = (uint64_t)u;
    if ((uint64_t)(u >> 64)) r += 0xFFFFFFFFULL;
    return r >= P ? r - P : r;
}

// ---- the radix-16 round on 16 lazy values (decimation in frequency / in time), twiddles w_16^K = +-2^(12 K') as shifts.
// FWD: w_16 = 2^156 (forward transform), INV: w_16^-1 = 2^36; exponents mod 192, 2^96 = -1.
template <int K, int INV>
struct W16 {
    static constexpr int E = ((INV ? 192 - 156 : 156) * K) % 192;
    static constexpr int SH = E % 96;
    static constexpr bool NEG = E >= 96;
};
template <int K, int INV>
GL96_FN void bfly_dif(X& u, X& v) {  // (u + v, (u - v) w_16^K)
    const X s = add(u, v);
    const X d = W16<K, INV>::NEG ? sub(v, u) : sub(u, v);
    u = s;
    if constexpr (W16<K, INV>::SH == 0) v = d;
    else v = shl<W16<K, INV>::SH>(d);
}
template <int K, int INV>
GL96_FN void bfly_dit(X& u, X& v) {  // (u + v w, u - v w)
    X t;
    if constexpr (W16<K, INV>::SH == 0) t = v;
    else t = shl<W16<K, INV>::SH>(v);
    const X a = add(u, t), b = sub(u, t);
    u = W16<K, INV>::NEG ? b : a;
    v = W16<K, INV>::NEG ? a : b;
}
// stage S (span 2^S in the field index a = e >> G) of a DFT over the top Q bits of the register index e
template <int Q, int INV, int S, int DIT, int E>
GL96_FN void stage_pair(X* x) {
    constexpr int G = 4 - Q, a = E >> G;
    if constexpr (((a >> S) & 1) == 0 && a < (1 << Q)) {
        constexpr int h = 1 << S, j = a & (h - 1), K = j * (8 >> S);
        if constexpr (DIT) bfly_dit<K, INV>(x[E], x[E + (h << G)]);
        else bfly_dif<K, INV>(x[E], x[E + (h << G)]);
    }
}
template <int Q, int INV, int S, int DIT>
GL96_FN void stage(X* x) {
    stage_pair<Q, INV, S, DIT, 0>(x), stage_pair<Q, INV, S, DIT, 1>(x), stage_pair<Q, INV, S, DIT, 2>(x), stage_pair<Q, INV, S, DIT, 3>(x);
    stage_pair<Q, INV, S, DIT, 4>(x), stage_pair<Q, INV, S, DIT, 5>(x), stage_pair<Q, INV, S, DIT, 6>(x), stage_pair<Q, INV, S, DIT, 7>(x);
    stage_pair<Q, INV, S, DIT, 8>(x), stage_pair<Q, INV, S, DIT, 9>(x), stage_pair<Q, INV, S, DIT, 10>(x), stage_pair<Q, INV, S, DIT, 11>(x);
    stage_pair<Q, INV, S, DIT, 12>(x), stage_pair<Q, INV, S, DIT, 13>(x), stage_pair<Q, INV, S, DIT, 14>(x), stage_pair<Q, INV, S, DIT, 15>(x);
}
template <int Q, int INV>
GL96_FN void dif_round(X* x) {
    if constexpr (Q >= 4) stage<Q, INV, 3, 0>(x);
    if constexpr (Q >= 3) stage<Q, INV, 2, 0>(x);
    if constexpr (Q >= 2) stage<Q, INV, 1, 0>(x);
    stage<Q, INV, 0, 0>(x);
}
template <int Q, int INV>
GL96_FN void dit_round(X* x) {
    stage<Q, INV, 0, 1>(x);
    if constexpr (Q >= 2) stage<Q, INV, 1, 1>(x);
    if constexpr (Q >= 3) stage<Q, INV, 2, 1>(x);
    if constexpr (Q >= 4) stage<Q, INV, 3, 1>(x);
}

}  // namespace gl96
