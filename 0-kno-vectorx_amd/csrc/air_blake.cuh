// BlakeChainAir (AIR id 3): BLAKE2b-256 parent-hash chain over a sequence of encoded headers --
// the hash-chain core of verify_subchain (/root/reference
// circuits/builder/subchain_verification.rs:150-177: hash_encoded_header + parent-hash link;
// circuits/builder/header.rs:14-19 curta_blake2b_variable).  The reference proves Blake2b with
// curta's byte-lookup STARK (starkyx v1.0.0, not vendored); this AIR is a from-scratch
// bit-decomposed ARX arithmetisation of RFC 7693, degree <= 3, 16 rows per compression:
//   r = 0 INIT (out-state = initial work vector), r = 1..12 ROUND (row r holds round r-1's eight
//   G evaluations), r = 13 FIN1 (T = H ^ v_lo, V' = v_hi, bits of H), r = 14 FIN2 (bits of h_out = T ^ V'),
//   r = 15 PAD (H = next h_in, digest register D updated).
// Columns: 8 G x 8 words x 64 bits, 64 carries, 32 message-schedule limbs, 64 range-check bits,
// H register (16 limbs; its bits only in free G cells of rows 13/14), digest register D (8 limbs), flags, byte counter, block number,
// zero-padding mask (8 mask bits + running count per row).
// Constraint ORDER is protocol: oracle/blake_air.py restates it independently.
#pragma once
#include <type_traits>

#include "air.cuh"
#include "blake_tables.h"

namespace blk {
constexpr int W_A1 = 0, W_D1 = 1, W_C1 = 2, W_B1 = 3, W_A2 = 4, W_D2 = 5, W_C2 = 6, W_B2 = 7;
constexpr int CAR0 = 4096, MS0 = 4160, MB0 = 4192, HL0 = 4256, D0 = 4272;
constexpr int ACT = 4280, FIN = 4281, FIRST = 4282, CAP = 4283, T = 4284, INC = 4285, TB0 = 4286, IB0 = 4318, NUM = 4326, FA = 4327, MK0 = 4328, CNT = 4336, COLS = 4337;
VX_HD constexpr int GB(int k, int w, int i) { return (k * 8 + w) * 64 + i; }
VX_HD constexpr int CAR(int k, int j) { return CAR0 + k * 8 + j; }
VX_HD constexpr int MS(int s, int h) { return MS0 + 2 * s + h; }
VX_HD constexpr int HL(int w, int h) { return HL0 + 2 * w + h; }  // chaining value as 16 limbs
// free G cells of the finalisation rows: row 13 holds T = H ^ v_lo (FT), V' = v_hi (FV), the bits of H (FH);
// row 14 holds the bits of h_out (FT)
VX_HD constexpr int FT(int w, int i) { return GB(w % 4, w / 4, i); }
VX_HD constexpr int FV(int w, int i) { return GB(w % 4, 2 + w / 4, i); }
VX_HD constexpr int FH(int w, int i) { return GB(w % 4, 4 + w / 4, i); }
// first bit column of out-state word v[w] (the diagonal-step outputs of a row)
VX_HD constexpr int OUT(int w) {
    return w < 4 ? GB(4 + w, W_A2, 0)
         : w < 8 ? GB(4 + ((w & 3) + 3) % 4, W_B2, 0)
         : w < 12 ? GB(4 + ((w & 3) + 2) % 4, W_C2, 0)
                  : GB(4 + ((w & 3) + 1) % 4, W_D2, 0);
}
#define BLK_IV_INIT {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL, \
                     0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL}
// the tables exist twice -- in device memory for the kernels, in host memory for the verifier
static __device__ const uint64_t IV[8] = BLK_IV_INIT;
static __device__ const uint8_t ORDER[16][16] = BLK_ORDER_INIT;
static __device__ const uint8_t MS_SRC[15][16] = BLK_MS_SRC_INIT;
static __device__ const uint8_t RC_SLOT[16] = BLK_RC_SLOT_INIT;
static const uint64_t IV_H[8] = BLK_IV_INIT;
static const uint8_t MS_SRC_H[15][16] = BLK_MS_SRC_INIT;
static const uint8_t RC_SLOT_H[16] = BLK_RC_SLOT_INIT;
#if defined(__HIP_DEVICE_COMPILE__)
VX_HD uint64_t iv(int i) { return IV[i]; }
VX_HD int ms_src(int r, int s) { return MS_SRC[r][s]; }
VX_HD int rc_slot(int r) { return RC_SLOT[r]; }
#else
VX_HD uint64_t iv(int i) { return IV_H[i]; }
VX_HD int ms_src(int r, int s) { return MS_SRC_H[r][s]; }
VX_HD int rc_slot(int r) { return RC_SLOT_H[r]; }
#endif
}  // namespace blk

struct BlakeAir {
    static constexpr int ID = 3, COLS = blk::COLS, PUB = 18, PERIODIC = 16, PERIOD_LOG = 4, QUOT_ROWS_PER_LANE = 1, AUX = 0, CHAL = 0, AUXPUB = 0;
    static constexpr int plog(int) { return 4; }  // 2 rows per lane: 266 VGPRs, 53 ms instead of 41

    template <class F, class Row, class C>
    __host__ __device__ static void eval(const Row& loc, const Row& nxt, const F* sel, const F* pub, const F*, const F*, C& c) {
        using namespace blk;
        const F one = F::from(1), two = F::from(2), two32 = F::from(1ULL << 32);
        F g_on = sel[0];
        for (int r = 1; r < 12; ++r) g_on = g_on + sel[r];
        auto at = [&](int which, int col) -> F { return which ? nxt[col] : loc[col]; };
        auto xorf = [&](F x, F y) -> F {  // x + y - 2xy (the doubling as an addition: a modular add is a third of a multiply)
            const F xy = x * y;
            return x + y - (xy + xy);
        };
        auto limb = [&](int which, int col0, int h) -> F {  // sum_i 2^i x_i over the 32 cells of one limb
            if constexpr (is_device_field<F>::value) {
                // device: the low and high halves of the 32 cells accumulate as plain integers (each sum < 2^64),
                // one multiply-add by 2^i per half; a single reduction at the end
                uint64_t lo[F::LANES], hi[F::LANES];
#pragma unroll
                for (int j = 0; j < F::LANES; ++j) lo[j] = hi[j] = 0;
#pragma unroll 8
                for (int i = 0; i < 32; ++i) {
                    const F x = at(which, col0 + 32 * h + i);
#pragma unroll
                    for (int j = 0; j < F::LANES; ++j) {
                        lo[j] += (uint64_t)(uint32_t)x.v[j] * (uint32_t)(1u << i);
                        hi[j] += (x.v[j] >> 32) * (uint32_t)(1u << i);
                    }
                }
                // lo + hi 2^32 = (lo + (hi << 32) mod 2^64) + ((hi >> 32) + carry) 2^64
                F r;
#pragma unroll
                for (int j = 0; j < F::LANES; ++j) {
                    uint64_t low;
                    const bool cy = __builtin_add_overflow(lo[j], hi[j] << 32, &low);
                    r.v[j] = gl_reduce128((hi[j] >> 32) + (cy ? 1 : 0), low);
                }
                return r;
            } else {
                F acc = at(which, col0 + 32 * h + 31);
                for (int i = 30; i >= 0; --i) acc = acc + acc + at(which, col0 + 32 * h + i);
                return acc;
            }
        };
        auto boolean = [&](int col) {
            F x = loc[col];
            c.constraint(x * (x - one));
        };
        // ---- 1. booleans
#ifndef VX_Q_UNROLL
#define VX_Q_UNROLL 4
#endif
#pragma unroll VX_Q_UNROLL
        for (int col = 0; col < 4096; ++col) boolean(col);
#pragma unroll 1
        for (int col = MB0; col < MB0 + 64; ++col) boolean(col);
#pragma unroll 1
        for (int col = TB0; col < TB0 + 32; ++col) boolean(col);
#pragma unroll 1
        for (int col = IB0; col < IB0 + 8; ++col) boolean(col);
        boolean(ACT);
        boolean(FIN);
        boolean(FIRST);
        boolean(CAP);
        boolean(FA);
        // ---- 2. carries
#pragma unroll 1
        for (int k = 0; k < 8; ++k)
#pragma unroll 1
            for (int j = 0; j < 8; ++j) {
                F x = loc[CAR(k, j)];
                if ((j & 2) == 0) c.constraint(x * (x - one) * (x - two));
                else c.constraint(x * (x - one));
            }
        // ---- 3. the eight G functions of the round held in the next row (every constraint gated by g_on)
        auto GON = c.open(g_on);
#pragma unroll 1
        for (int k = 0; k < 8; ++k) {
            int wa, ca, wb, cb, wc, cc, wd, cd, xs, ys;  // (row selector, first bit column) of the inputs
            if (k < 4) {
                wa = wb = wc = wd = 0;
                ca = OUT(k), cb = OUT(4 + k), cc = OUT(8 + k), cd = OUT(12 + k);
                xs = 2 * k, ys = 2 * k + 1;
            } else {
                const int j = k - 4;
                wa = wb = wc = wd = 1;
                ca = GB(j, W_A2, 0), cb = GB((j + 1) % 4, W_B2, 0), cc = GB((j + 2) % 4, W_C2, 0), cd = GB((j + 3) % 4, W_D2, 0);
                xs = 8 + 2 * j, ys = 8 + 2 * j + 1;
            }
            auto add3 = [&](int w1, int c1, int w2, int c2, int msg_slot, int res_slot, int car_j) {
                F cin = F::from(0);
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    F lhs = limb(w1, c1, h) + limb(w2, c2, h);
                    if (msg_slot >= 0) lhs = lhs + nxt[MS(msg_slot, h)];
                    if (h) lhs = lhs + cin;
                    F car = nxt[CAR(k, car_j + h)];
                    c.gated(GON, lhs - limb(1, GB(k, res_slot, 0), h) - two32 * car);
                    cin = car;
                }
            };
            auto xorrot = [&](int w1, int c1, int w2, int c2, int res_slot, int rot) {
#pragma unroll VX_Q_UNROLL
                for (int i = 0; i < 64; ++i) {
                    const int s = (i + rot) & 63;
                    c.gated(GON, nxt[GB(k, res_slot, i)] - xorf(at(w1, c1 + s), at(w2, c2 + s)));
                }
            };
            add3(wa, ca, wb, cb, xs, W_A1, 0);
            xorrot(wd, cd, 1, GB(k, W_A1, 0), W_D1, 32);
            add3(wc, cc, 1, GB(k, W_D1, 0), -1, W_C1, 2);
            xorrot(wb, cb, 1, GB(k, W_C1, 0), W_B1, 24);
            add3(1, GB(k, W_A1, 0), 1, GB(k, W_B1, 0), ys, W_A2, 4);
            xorrot(1, GB(k, W_D1, 0), 1, GB(k, W_A2, 0), W_D2, 16);
            add3(1, GB(k, W_C1, 0), 1, GB(k, W_D2, 0), -1, W_C2, 6);
            xorrot(1, GB(k, W_B1, 0), 1, GB(k, W_C2, 0), W_B2, 63);
        }
        c.close(GON);
        // ---- 4. INIT row (gated by sel[0])
        const F fin = loc[FIN];
        auto S0 = c.open(sel[0]);
#pragma unroll 1
        for (int wd = 0; wd < 16; ++wd) {
            const int col0 = OUT(wd);
            if (wd < 8) {  // v[0..8) = h_in: compared limb-wise with the H register
#pragma unroll 1
                for (int h = 0; h < 2; ++h) c.gated(S0, limb(0, col0, h) - loc[HL(wd, h)]);
                continue;
            }
#pragma unroll 1
            for (int i = 0; i < 64; ++i) {
                F cell = loc[col0 + i], want;
                const int bit = (int)((iv(wd - 8) >> i) & 1);
                if (wd == 12 && i < 32) want = bit ? one - loc[TB0 + i] : loc[TB0 + i];
                else if (wd == 14) want = bit ? one - fin : fin;
                else want = F::from((uint64_t)bit);
                c.gated(S0, cell - want);
            }
        }
        c.close(S0);
        // ---- 5. finalisation
        F keep_h = sel[15];
        for (int r = 0; r < 13; ++r) keep_h = keep_h + sel[r];
        auto S12 = c.open(sel[12]);
        auto S13 = c.open(sel[13]);
#pragma unroll 1
        for (int wd = 0; wd < 8; ++wd) {
            const int lo0 = OUT(wd), hi0 = OUT(8 + wd);
            const uint64_t ivp = wd == 0 ? (iv(0) ^ 0x01010020ULL) : iv(wd);
#pragma unroll 1
            for (int i = 0; i < 64; ++i) {
                c.gated(S12, nxt[FT(wd, i)] - xorf(nxt[FH(wd, i)], loc[lo0 + i]));
                c.gated(S12, nxt[FV(wd, i)] - loc[hi0 + i]);
                c.gated(S13, nxt[FT(wd, i)] - xorf(loc[FT(wd, i)], loc[FV(wd, i)]));
            }
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                const F hl = loc[HL(wd, h)], hn = nxt[HL(wd, h)];
                const F ivl = F::from((ivp >> (32 * h)) & 0xFFFFFFFFULL);
                c.gated(S13, hl - limb(0, FH(wd, 0), h));
                c.constraint(sel[14] * (hl - limb(0, FT(wd, 0), h)));
                c.constraint(sel[14] * (hn - (fin * ivl + (one - fin) * hl)));
                c.constraint(keep_h * (hn - hl));
            }
        }
        c.close(S12);
        c.close(S13);
        // ---- 6. message schedule, range check, link
#pragma unroll 1
        for (int s = 0; s < 16; ++s)
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                const F n_ = nxt[MS(s, h)];
                F acc = sel[0] * (n_ - loc[MS(ms_src(0, s), h)]);
#pragma unroll 1
                for (int r = 1; r < 15; ++r) acc = acc + sel[r] * (n_ - loc[MS(ms_src(r, s), h)]);
                c.constraint(acc);
            }
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            F acc = sel[0] * loc[MS(rc_slot(0), h)];
#pragma unroll 1
            for (int r = 1; r < 16; ++r) acc = acc + sel[r] * loc[MS(rc_slot(r), h)];
            c.constraint(acc - limb(0, MB0, h));
        }
        // ---- 6b. bytes at positions >= inc are zero (RFC 7693 zero padding of the last chunk): row r sees
        // word r's bits (MB); MK is a monotone mask over the 128 byte positions with popcount inc
        {
            const F in_blk = one - sel[15];
#pragma unroll 1
            for (int b = 0; b < 8; ++b) boolean(MK0 + b);
#pragma unroll 1
            for (int b = 0; b < 7; ++b) c.constraint(loc[MK0 + b + 1] * (one - loc[MK0 + b]));
            c.constraint(in_blk * nxt[MK0] * (one - loc[MK0 + 7]));
            F msum_l = loc[MK0], msum_n = nxt[MK0];
#pragma unroll 1
            for (int b = 1; b < 8; ++b) {
                msum_l = msum_l + loc[MK0 + b];
                msum_n = msum_n + nxt[MK0 + b];
            }
            c.constraint(sel[0] * (loc[CNT] - msum_l));
            c.constraint(in_blk * (nxt[CNT] - loc[CNT] - msum_n));
            c.constraint(sel[15] * (loc[CNT] - loc[INC]));
#pragma unroll 1
            for (int b = 0; b < 8; ++b) {
                F byte = loc[MB0 + 8 * b + 7];
#pragma unroll 1
                for (int i = 6; i >= 0; --i) byte = byte + byte + loc[MB0 + 8 * b + i];
                c.constraint((one - loc[MK0 + b]) * byte);
            }
        }
        const F first = loc[FIRST];
#pragma unroll 1
        for (int s = 0; s < 4; ++s)
#pragma unroll 1
            for (int h = 0; h < 2; ++h) c.constraint(sel[0] * first * (loc[MS(s, h)] - loc[D0 + 2 * s + h]));
        // block number: bytes 32..36 = SCALE compact int, 4-byte mode: 4 * number + 2 (decoder.rs:64-66)
        c.constraint(sel[0] * first * (loc[MS(4, 0)] - (F::from(4) * loc[NUM] + two)));
        // ---- 7. per-block registers
        const F in_block = one - sel[15];
        {
            const int regs[8] = {ACT, FIN, FIRST, CAP, T, INC, NUM, FA};
#pragma unroll 1
            for (int q = 0; q < 8; ++q) c.constraint(in_block * (nxt[regs[q]] - loc[regs[q]]));
        }
        c.constraint(loc[CAP] - loc[ACT] * fin);
        c.constraint(loc[FA] - first * loc[ACT]);
        c.transition(sel[15] * (nxt[NUM] - loc[NUM] - nxt[FA]));  // sequential numbers (subchain_verification.rs:166-168)
        // ACT is a property of a whole MESSAGE, not of a block: it may not change between the chunks of one message
        // (else a junk message could bump NUM through FA on its first chunk and skip the digest capture on its last),
        // and once the chain has gone inactive (padding) it stays inactive
        c.constraint(sel[15] * (one - fin) * (nxt[ACT] - loc[ACT]));
        c.transition(nxt[ACT] * (one - loc[ACT]));
        c.constraint(sel[15] * (nxt[FIRST] - fin));
        c.constraint(sel[15] * (nxt[T] - (one - fin) * loc[T] - nxt[INC]));
        {
            F tb = loc[TB0 + 31];
#pragma unroll 1
            for (int i = 30; i >= 0; --i) tb = tb + tb + loc[TB0 + i];
            c.constraint(loc[T] - tb);
            F ib = loc[IB0 + 7];
#pragma unroll 1
            for (int i = 6; i >= 0; --i) ib = ib + ib + loc[IB0 + i];
            c.constraint(loc[INC] - ib);
            const F c128 = F::from(128);
            c.constraint(loc[IB0 + 7] * (loc[INC] - c128));
            c.constraint((one - fin) * (loc[INC] - c128));
        }
        // ---- 8. digest register
        const F cap = loc[CAP];
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
            const F d = loc[D0 + j], dn = nxt[D0 + j];
            c.transition((one - sel[14]) * (dn - d));
            c.constraint(sel[14] * (dn - (cap * loc[HL(j / 2, j % 2)] + (one - cap) * d)));
        }
        // ---- 9. boundary
#pragma unroll 1
        for (int j = 0; j < 8; ++j) c.first_row(loc[D0 + j] - pub[j]);
#pragma unroll 1
        for (int j = 0; j < 8; ++j) c.last_row(loc[D0 + j] - pub[8 + j]);
        c.last_row(fin - one);
        c.first_row(loc[NUM] - pub[16]);
        c.last_row(loc[NUM] - pub[17]);
    }
};
