// Sha512Air (AIR ids 11 / 14 / 13 for 2^16 / 2^15 / 2^10 rows): H_i = SHA-512(R_i || A_i || M) for every enabled slot -- the hash half
// of the conditional EdDSA verifications of verify_simple_justification (/root/reference
// circuits/builder/justification.rs:229-243 -> curta's EdDSA gadget, starkyx v1.0.0, not vendored; native mirror
// circuits/input/mod.rs:241-247).  FIPS 180-4, one round per row; 64-bit words are split in 32-bit halves wherever
// arithmetic happens.  A slot takes 160 rows: block 1 = R || A || M || 80 00.. (80 rows), block 2 = the constant length
// block (80 rows, continues from block 1).  Everything positional is a periodic column of full period.  Only words an
// XOR reads are bit columns (a, b, c, e, f, g, new a, new e, window positions 0, 1, 14); Sigma0 / Sigma1 / Ch / Maj are
// degree-3 polynomials of those bits (no cells); the schedule's sigma0 + sigma1 is one value (two halves) defined by an
// unconditional polynomial identity, because the schedule equation itself carries a selector.  R || A arrives over the bus from EdAir (air_ed.cuh) at rows 0, 2, 4, 6 of block 1 (8
// little-endian 16-bit limbs per tuple, cut from the bits of window positions 0 / 1); the digest goes back from rows 74..79
// of block 2 as three 32-bit feed-forward halves per tuple (held in FFV from row 74 on; EdAir's byte cells bound them), all
// under the slot's flag SGF.
// Public inputs: message words 8..14 of block 1 as (lo, hi) halves, bus_on.  Constraint ORDER is protocol:
// oracle/sha512_air.py restates it independently.
#pragma once
#include <vector>

#include "air.cuh"
#include "air_ed.cuh"
#include "ed25519_constants.h"

namespace s5 {
constexpr int A_ = 0, B_ = 64, C_ = 128, E_ = 192, F_ = 256, G_ = 320, DV = 384, HV = 386, NA0 = 388, NE0 = 452;
constexpr int W0B = 516, W1B = 580, W14B = 644, WV0 = 708, WV15 = 732;
constexpr int SV = 734;  // sigma0(W[1]) + sigma1(W[14]) as (lo, hi) halves (values below 2^33)
constexpr int CE0 = 736, CA0 = 742, CW0 = 748, FFV0 = 752, FFC0 = 768, HIN0 = 784, SGF = 800, COLS = 801;
constexpr int SLOT_ROWS = 160, SEND0 = 80 + 74, MSG_LEN = 53;
enum { P_B1, P_INB, P_SCHED, P_KLO, P_KHI, P_LAST, P_CONT, P_HSET, P_RCV, P_T0, P_FFK, P_SGK, P_SD0, N_PERIODIC = 18 };
VX_HD constexpr int WV(int p) { return p == 15 ? WV15 : WV0 + 2 * (p - 2); }
VX_HD constexpr int st_bits(int wd) { return wd == 0 ? A_ : wd == 1 ? B_ : wd == 2 ? C_ : wd == 4 ? E_ : wd == 5 ? F_ : wd == 6 ? G_ : -1; }
#define S5_IV_INIT {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL, \
                    0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL}
static __device__ const uint64_t IV[8] = S5_IV_INIT;
static const uint64_t IV_H[8] = S5_IV_INIT;
static const uint64_t K_H[80] = SHA512_K_INIT;
#if defined(__HIP_DEVICE_COMPILE__)
VX_HD uint64_t iv(int i) { return IV[i]; }
#else
VX_HD uint64_t iv(int i) { return IV_H[i]; }
#endif
VX_HD constexpr uint64_t pad2(int p) { return p == 15 ? 8 * (64 + MSG_LEN) : 0; }  // block 2 of a 117-byte message
}  // namespace s5

template <int LOGN, int ID_>
struct Sha512AirT {
    static constexpr int ID = ID_, COLS = s5::COLS, PUB = 15, PERIODIC = s5::N_PERIODIC, PERIOD_LOG = LOGN, QUOT_ROWS_PER_LANE = 1, AUX = 4, CHAL = 4, AUXPUB = 1, EXACT_LOG = 1;
    static constexpr int plog(int) { return LOGN; }
    static constexpr size_t max_slots() { return ((size_t)1 << LOGN) / s5::SLOT_ROWS; }

    static void periodic_values(std::vector<uint64_t>& v) {
        using namespace s5;
        const size_t n = (size_t)1 << LOGN;
        v.assign((size_t)N_PERIODIC * n, 0);
        auto P = [&](int q, size_t row) -> uint64_t& { return v[(size_t)q * n + row]; };
        for (size_t s = 0; s < max_slots(); ++s) {
            const size_t b = s * SLOT_ROWS;
            P(P_B1, b) = 1;
            for (int blk = 0; blk < 2; ++blk) {
                const size_t o = b + 80 * blk;
                for (int r = 0; r < 80; ++r) {
                    if (r < 79) P(P_INB, o + r) = 1;
                    if (r < 64) P(P_SCHED, o + r) = 1;
                    P(P_KLO, o + r) = K_H[r] & 0xFFFFFFFFULL, P(P_KHI, o + r) = K_H[r] >> 32;
                }
                P(P_LAST, o + 79) = 1, P(P_HSET, o) = 1;
            }
            P(P_CONT, b + 79) = 1;
            for (int j = 0; j < 4; ++j) P(P_RCV, b + 2 * j) = 1, P(P_T0, b + 2 * j) = 4 * s + j;
            for (int j = 0; j < 6; ++j) P(P_SD0 + j, b + SEND0 + j) = 1, P(P_T0, b + SEND0 + j) = 8 * s + j;
            for (size_t r = b + SEND0; r < b + SEND0 + 5; ++r) P(P_FFK, r) = 1;
            for (size_t r = b; r < b + 159; ++r) P(P_SGK, r) = 1;
        }
    }

    template <class F, class Row, class Cn>
    __host__ __device__ static void eval(const Row& loc, const Row& nxt, const F* per, const F* pub, const F* chal, const F* apub, Cn& c) {
        using namespace s5;
        const F inb = per[P_INB], sched_on = per[P_SCHED], last = per[P_LAST];
        const F one = F::from(1), two = F::from(2), four = F::from(4), zero = F::from(0), two32 = F::from(1ULL << 32);
        auto val = [&](const Row& row, int col0, int nb) -> F {
            F acc = row[col0 + nb - 1];
#pragma unroll 1
            for (int i = nb - 2; i >= 0; --i) acc = acc + acc + row[col0 + i];
            return acc;
        };
        // half h (0 = low 32 bits) of window position p / state word wd
        auto window = [&](const Row& row, int p, int h) -> F {
            return p == 0 ? val(row, W0B + 32 * h, 32) : p == 1 ? val(row, W1B + 32 * h, 32) : p == 14 ? val(row, W14B + 32 * h, 32) : row[WV(p) + h];
        };
        auto state_word = [&](const Row& row, int wd, int h) -> F { return wd == 3 ? row[DV + h] : wd == 7 ? row[HV + h] : val(row, st_bits(wd) + 32 * h, 32); };
        // ---- 1. booleans
        {
            const int lo[5] = {0, NA0, CE0, FFC0, SGF}, hi[5] = {DV, WV0, FFV0, HIN0, SGF + 1};
#pragma unroll 1
            for (int q = 0; q < 5; ++q)
#pragma unroll 1
                for (int col = lo[q]; col < hi[q]; ++col) {
                    const F x = loc[col];
                    c.constraint(x * (x - one));
                }
        }
        // ---- 2. SV = sigma0(W[1]) + sigma1(W[14]), per half: XORs as polynomials of the window bits (shifted-out bits are absent)
        {
            auto sig_half = [&](int col0, int r0, int r1, int shift, int h) -> F {
                F acc = zero;
#pragma unroll 1
                for (int i = 31; i >= 0; --i) {
                    const int b = 32 * h + i;
                    const F x = loc[col0 + ((b + r0) & 63)], y = loc[col0 + ((b + r1) & 63)];
                    const F xy = x * y;
                    if (b + shift >= 64) acc = acc + acc + (x + y - two * xy);
                    else {
                        const F z = loc[col0 + b + shift];
                        acc = acc + acc + (x + y + z - two * (xy + (x + y) * z) + four * (xy * z));
                    }
                }
                return acc;
            };
#pragma unroll 1
            for (int h = 0; h < 2; ++h) c.constraint(loc[SV + h] - sig_half(W1B, 1, 8, 7, h) - sig_half(W14B, 19, 61, 6, h));
        }
        // ---- 3. the round (local, every row): Sigma / Ch / Maj are degree-3 polynomials of the state bits
        {
            auto x3half = [&](int col0, int r0, int r1, int r2, int h) -> F {
                F acc = zero;
#pragma unroll 1
                for (int i = 31; i >= 0; --i) {
                    const int b = 32 * h + i;
                    const F x = loc[col0 + ((b + r0) & 63)], y = loc[col0 + ((b + r1) & 63)], z = loc[col0 + ((b + r2) & 63)];
                    const F xy = x * y;
                    acc = acc + acc + (x + y + z - two * (xy + (x + y) * z) + four * (xy * z));
                }
                return acc;
            };
            auto chhalf = [&](int h) -> F {
                F acc = zero;
#pragma unroll 1
                for (int i = 31; i >= 0; --i) {
                    const F e = loc[E_ + 32 * h + i], f = loc[F_ + 32 * h + i], g = loc[G_ + 32 * h + i];
                    acc = acc + acc + (e * f + (one - e) * g);
                }
                return acc;
            };
            auto majhalf = [&](int h) -> F {
                F acc = zero;
#pragma unroll 1
                for (int i = 31; i >= 0; --i) {
                    const F a = loc[A_ + 32 * h + i], b = loc[B_ + 32 * h + i], cc = loc[C_ + 32 * h + i];
                    const F ab = a * b;
                    acc = acc + acc + (ab + (a + b) * cc - two * (ab * cc));
                }
                return acc;
            };
            F cin_e = zero, cin_a = zero;
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                const F t1 = loc[HV + h] + x3half(E_, 14, 18, 41, h) + chhalf(h) + per[P_KLO + h] + val(loc, W0B + 32 * h, 32);
                const F ce = val(loc, CE0 + 3 * h, 3), ca = val(loc, CA0 + 3 * h, 3);
                F rhs_e = loc[DV + h] + t1, rhs_a = t1 + x3half(A_, 28, 34, 39, h) + majhalf(h);
                if (h) rhs_e = rhs_e + cin_e, rhs_a = rhs_a + cin_a;
                c.constraint(val(loc, NE0 + 32 * h, 32) + two32 * ce - rhs_e);
                c.constraint(val(loc, NA0 + 32 * h, 32) + two32 * ca - rhs_a);
                cin_e = ce, cin_a = ca;
            }
        }
        // ---- 4. state shift inside a block
#pragma unroll 1
        for (int i = 0; i < 64; ++i) {
            c.constraint(inb * (nxt[A_ + i] - loc[NA0 + i]));
            c.constraint(inb * (nxt[E_ + i] - loc[NE0 + i]));
            c.constraint(inb * (nxt[B_ + i] - loc[A_ + i]));
            c.constraint(inb * (nxt[C_ + i] - loc[B_ + i]));
            c.constraint(inb * (nxt[F_ + i] - loc[E_ + i]));
            c.constraint(inb * (nxt[G_ + i] - loc[F_ + i]));
        }
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            c.constraint(inb * (nxt[DV + h] - val(loc, C_ + 32 * h, 32)));
            c.constraint(inb * (nxt[HV + h] - val(loc, G_ + 32 * h, 32)));
        }
        // ---- 5. message schedule: window shift, w_(r+16) while r <= 63
#pragma unroll 1
        for (int i = 0; i < 64; ++i) c.constraint(inb * (nxt[W0B + i] - loc[W1B + i]));
#pragma unroll 1
        for (int p = 1; p < 15; ++p)
#pragma unroll 1
            for (int h = 0; h < 2; ++h) c.constraint(inb * (window(nxt, p, h) - window(loc, p + 1, h)));
        {
            F cin = zero;
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                const F cw = val(loc, CW0 + 2 * h, 2);
                F rhs = loc[SV + h] + loc[WV(9) + h] + val(loc, W0B + 32 * h, 32);
                if (h) rhs = rhs + cin;
                c.constraint(sched_on * (nxt[WV15 + h] + two32 * cw - rhs));
                cin = cw;
            }
        }
        // ---- 6. feed-forward at r = 79: FF = H_in + (NA, a, b, c, NE, e, f, g)
        {
            const int s80[8] = {NA0, A_, B_, C_, NE0, E_, F_, G_};
#pragma unroll 1
            for (int wd = 0; wd < 8; ++wd)
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    F rhs = loc[HIN0 + 2 * wd + h] + val(loc, s80[wd] + 32 * h, 32);
                    if (h) rhs = rhs + loc[FFC0 + 2 * wd];
                    c.constraint(last * (loc[FFV0 + 2 * wd + h] + two32 * loc[FFC0 + 2 * wd + h] - rhs));
                }
        }
        // ---- 7. block starts: IV at block 1, block 1's output at block 2; the H_in register
        const F b1 = per[P_B1], cont = per[P_CONT], hset = per[P_HSET];
#pragma unroll 1
        for (int wd = 0; wd < 8; ++wd)
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                const F sl = state_word(loc, wd, h), sn = state_word(nxt, wd, h);
                c.constraint(b1 * (sl - F::from((iv(wd) >> (32 * h)) & 0xFFFFFFFFULL)));
                c.constraint(cont * (sn - loc[FFV0 + 2 * wd + h]));
                c.constraint(hset * (loc[HIN0 + 2 * wd + h] - sl));
                c.constraint(inb * (nxt[HIN0 + 2 * wd + h] - loc[HIN0 + 2 * wd + h]));
            }
        // ---- 8. message words: the public tail of block 1, the constant block 2
#pragma unroll 1
        for (int j = 0; j < 8; ++j)
#pragma unroll 1
            for (int h = 0; h < 2; ++h) c.constraint(b1 * (window(loc, 8 + j, h) - (j < 7 ? pub[2 * j + h] : zero)));
#pragma unroll 1
        for (int p = 0; p < 16; ++p)
#pragma unroll 1
            for (int h = 0; h < 2; ++h) c.constraint(cont * (window(nxt, p, h) - F::from((pad2(p) >> (32 * h)) & 0xFFFFFFFFULL)));
        // ---- 9. the digest is held from row 74 of block 2 on (it is sent from there); the slot flag is kept
#pragma unroll 1
        for (int k = 0; k < 16; ++k) c.constraint(per[P_FFK] * (nxt[FFV0 + k] - loc[FFV0 + k]));
        c.constraint(per[P_SGK] * (nxt[SGF] - loc[SGF]));
        // ---- 10. the bus: receive rows take 8 limbs of the words at window positions 0 and 1 (limb j of a word = bytes 2j, 2j+1
        // of its big-endian byte string, little-endian); send row j gives the feed-forward halves 3j, 3j+1, 3j+2
        {
            const X2<F> beta{chal[0], chal[1]}, gamma{chal[2], chal[3]}, g2 = gamma * gamma, g3 = g2 * gamma, g4 = g2 * g2;
            const F k8 = F::from(256), k16 = F::from(65536), k32 = F::from(1ULL << 32), rcv = per[P_RCV];
            auto limb = [&](int q) -> F {  // q = 0..7: limbs of W0 then W1
                const int col0 = (q < 4 ? W0B : W1B), j = q & 3;
                return val(loc, col0 + 56 - 16 * j, 8) + val(loc, col0 + 48 - 16 * j, 8) * k8;
            };
            F t[3] = {rcv * (limb(0) + limb(1) * k16 + limb(2) * k32), rcv * (limb(3) + limb(4) * k16 + limb(5) * k32), rcv * (limb(6) + limb(7) * k16)};
            F snd = zero;
#pragma unroll 1
            for (int i = 0; i < 3; ++i)
#pragma unroll 1
                for (int j = 0; j < 6; ++j)
                    if (3 * j + i < 16) t[i] = t[i] + per[P_SD0 + j] * loc[FFV0 + 3 * j + i];
#pragma unroll 1
            for (int j = 0; j < 6; ++j) snd = snd + per[P_SD0 + j];
            const F m = loc[SGF] * pub[14] * (snd - rcv);
            const F tag = snd * F::from(edc::TAG_EDH) + rcv * F::from(edc::TAG_EDMSG);
            const X2<F> d = beta + per[P_T0] + gamma * t[0] + g2 * t[1] + g3 * t[2] + g4 * tag;
            const X2<F> h{loc[COLS], loc[COLS + 1]}, z{loc[COLS + 2], loc[COLS + 3]}, zn{nxt[COLS + 2], nxt[COLS + 3]};
            c.constraint_x2(h * d - m);
            c.constraint_x2(zn - z - h + X2<F>{apub[0], apub[1]});
        }
    }
};
using Sha512Air16 = Sha512AirT<16, 11>;
using Sha512Air15 = Sha512AirT<15, 14>;
using Sha512Air10 = Sha512AirT<10, 13>;
