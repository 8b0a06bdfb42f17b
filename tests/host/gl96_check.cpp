// Host check of 0-kno-vectorx_amd/csrc/gl96.h (the lazy arithmetic of the NTT rounds) against 128-bit integer arithmetic:
// shifts, folds, and whole radix-16 rounds (both directions, every Q) on random and corner-case inputs.  Prints "ok <max |k|
// seen at a shift> <max |k| seen at a fold>" or the first mismatch.  Built and run by tests/test_gl96_host.py (g++).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static int g_max_k_shift = 0, g_max_k_fold = 0, g_track = 0;
static long long g_shift_headroom = 1LL << 40;  // min over shifts of 2^31 - (|k| + 1) 2^t
#define GL96_TRACK_SHIFT(k, t)                                                        \
    if (g_track) {                                                                    \
        const int ak = (k) < 0 ? -(k) : (k);                                          \
        if (ak > g_max_k_shift) g_max_k_shift = ak;                                   \
        const long long room = (1LL << 31) - ((long long)(ak + 1) << (t));            \
        if (room < g_shift_headroom) g_shift_headroom = room;                         \
    }
#include "../../0-kno-vectorx_amd/csrc/gl96.h"

typedef unsigned __int128 u128;
static const uint64_t P = 0xFFFFFFFF00000001ULL;
static uint64_t mulmod(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % P); }
static uint64_t powmod(uint64_t b, uint64_t e) {
    uint64_t r = 1;
    for (; e; e >>= 1, b = mulmod(b, b))
        if (e & 1) r = mulmod(r, b);
    return r;
}
static uint64_t val(const gl96::X& a) {  // V mod p
    __int128 v = (__int128)(((u128)a.hi << 32) | a.lo) + ((__int128)a.k << 64);
    v %= (__int128)P;
    if (v < 0) v += P;
    return (uint64_t)v;
}
static uint64_t rng_s = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd() {
    uint64_t z = (rng_s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL, z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static const uint64_t SPECIAL[] = {0, 1, 2, P - 1, P - 2, 0xFFFFFFFFULL, 0x100000000ULL, 0x100000001ULL, P - 0x100000000ULL, P - 0xFFFFFFFFULL,
                                   1ULL << 63, (1ULL << 63) - 1, 0xFFFFFFFF00000000ULL, 0xFFFFFFFE00000002ULL, 0x1FFFFFFFFULL, 0x8000000080000000ULL,
                                   0x7FFFFFFF7FFFFFFFULL, 0xFFFFFFFEFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFF00000002ULL, 0x00000000FFFFFFFEULL, 0xFFFFFFFE00000000ULL};
static const int NSPECIAL = sizeof(SPECIAL) / sizeof(SPECIAL[0]);
static uint64_t pick(int mode) {  // mode 0 random, 1 special (any 64-bit representative), 2 mixed
    if (mode == 0 || (mode == 2 && (rnd() & 1))) return rnd();
    return SPECIAL[rnd() % NSPECIAL];
}

template <int S>
static int check_shl() {
    const uint64_t two_s = powmod(2, S);
    for (int it = 0; it < 200000; ++it) {
        gl96::X a = gl96::from64(pick(it % 3));
        a.k = (int)(rnd() % 15) - 7;
        if ((S & 31) == 28 && a.k > 6) a.k = 6;
        if ((S & 31) == 28 && a.k < -6) a.k = -6;
        const gl96::X r = gl96::shl<S>(a);
        if (val(r) != mulmod(val(a), two_s) || r.k > 2 || r.k < -2) {
            printf("shl<%d> mismatch: lo %08x hi %08x k %d -> k %d\n", S, a.lo, a.hi, a.k, r.k);
            return 1;
        }
    }
    return 0;
}

static void ref_bfly(uint64_t& u, uint64_t& v, int K, int inv, int dit) {
    const uint64_t w = powmod(powmod(2, inv ? 36 : 156), K);
    if (dit) {
        const uint64_t t = mulmod(v % P, w), a = (uint64_t)(((u128)(u % P) + t) % P), b = (uint64_t)(((u128)(u % P) + P - t) % P);
        u = a, v = b;
    } else {
        const uint64_t s = (uint64_t)(((u128)(u % P) + v % P) % P), d = mulmod((uint64_t)(((u128)(u % P) + P - v % P) % P), w);
        u = s, v = d;
    }
}
static void ref_round(uint64_t* x, int Q, int inv, int dit) {
    const int G = 4 - Q;
    for (int si = 0; si < Q; ++si) {
        const int S = dit ? si : Q - 1 - si;
        for (int E = 0; E < 16; ++E) {
            const int a = E >> G;
            if (((a >> S) & 1) == 0) {
                const int h = 1 << S, j = a & (h - 1), K = j * (8 >> S);
                ref_bfly(x[E], x[E + (h << G)], K, inv, dit);
            }
        }
    }
}
template <int Q, int INV, int DIT>
static int check_round() {
    for (int it = 0; it < 60000; ++it) {
        uint64_t in[16], ref[16];
        gl96::X x[16];
        const int mode = it % 4;
        for (int e = 0; e < 16; ++e) {
            in[e] = mode == 3 ? SPECIAL[(it / 4 + e * (it % 7 + 1)) % NSPECIAL] : pick(mode);
            if (it == 3) in[e] = 0xFFFFFFFFFFFFFFFFULL;
            if (it == 7) in[e] = (e & 1) ? 0 : 0xFFFFFFFFFFFFFFFFULL;
            if (it == 11) in[e] = (e & 8) ? 0xFFFFFFFFFFFFFFFFULL : 0;
            ref[e] = in[e], x[e] = gl96::from64(in[e]);
        }
        g_track = 1;
        if (DIT) gl96::dit_round<Q, INV>(x);
        else gl96::dif_round<Q, INV>(x);
        g_track = 0;
        ref_round(ref, Q, INV, DIT);
        uint32_t margin = 0xFFFFFFFFu;
        for (int e = 0; e < 16; ++e) {
            const int ak = x[e].k < 0 ? -x[e].k : x[e].k;
            if (ak > g_max_k_fold) g_max_k_fold = ak;
            margin = gl96::fold_margin(margin, x[e]);
        }
        for (int e = 0; e < 16; ++e) {
            if (val(x[e]) != ref[e] % P || gl96::fold_exact(x[e]) != ref[e] % P) {
                printf("round<Q=%d,INV=%d,DIT=%d> mismatch at e=%d it=%d\n", Q, INV, DIT, e, it);
                return 1;
            }
            if (gl96::fold_ok(margin) && gl96::fold_fast(x[e]) % P != ref[e] % P) {
                printf("fold_fast accepted a wrapping value: round<Q=%d,INV=%d,DIT=%d> e=%d it=%d\n", Q, INV, DIT, e, it);
                return 1;
            }
        }
    }
    return 0;
}
static int check_fold() {
    for (int it = 0; it < 2000000; ++it) {
        gl96::X a = gl96::from64(pick(it % 3));
        a.k = (int)(rnd() % (2 * gl96::FOLD_K + 1)) - (int)gl96::FOLD_K;
        if (it % 5 == 0) a.lo = (uint32_t)(rnd() % 200) - 100;   // near the wrap points
        if (it % 7 == 0) a.hi = (uint32_t)(rnd() % 200) - 100;
        const uint64_t want = val(a);
        if (gl96::fold_exact(a) != want) return printf("fold_exact mismatch\n"), 1;
        if (gl96::fold_ok(gl96::fold_margin(0xFFFFFFFFu, a)) && gl96::fold_fast(a) % P != want) return printf("fold_fast mismatch lo %08x hi %08x k %d\n", a.lo, a.hi, a.k), 1;
    }
    return 0;
}

int main() {
    int bad = 0;
    bad |= check_shl<12>() | check_shl<24>() | check_shl<36>() | check_shl<48>() | check_shl<60>() | check_shl<72>() | check_shl<84>();
    bad |= check_fold();
    bad |= check_round<4, 0, 0>() | check_round<4, 1, 0>() | check_round<4, 0, 1>() | check_round<4, 1, 1>();
    bad |= check_round<3, 0, 0>() | check_round<3, 1, 0>() | check_round<3, 0, 1>() | check_round<3, 1, 1>();
    bad |= check_round<2, 0, 0>() | check_round<2, 1, 0>() | check_round<2, 0, 1>() | check_round<2, 1, 1>();
    bad |= check_round<1, 0, 0>() | check_round<1, 1, 0>() | check_round<1, 0, 1>() | check_round<1, 1, 1>();
    if (g_shift_headroom < 0) return printf("a shift overflowed its third word (headroom %lld)\n", g_shift_headroom), 1;
    if (g_max_k_fold > (int)gl96::FOLD_K) return printf("|k| %d at a fold exceeds FOLD_K\n", g_max_k_fold), 1;
    if (bad) return 1;
    printf("ok %d %d %lld\n", g_max_k_shift, g_max_k_fold, g_shift_headroom);
    return 0;
}
