// ShaChainAir (AIR id 4): the chained SHA-256 authority-set commitment
//   h_0 = SHA256(pk_0), h_i = SHA256(h_{i-1} || pk_i)
// of compute_authority_set_commitment (/root/reference circuits/builder/justification.rs:127-162;
// native mirror circuits/input/mod.rs:250-260).  The reference proves SHA-256 with curta's STARK
// (starkyx v1.0.0, not vendored); this AIR is a from-scratch FIPS 180-4 arithmetisation, degree <= 3:
// one row per round, 64 rows per compression; three-input XORs as x + y + z = r + 2c, Ch as a
// degree-2 expression, Maj through (maj, parity) bits; block types FIRST / DATA / PAD / IDLE.
// Constraint ORDER is protocol: oracle/sha_air.py restates it independently.
#pragma once
#include "air.cuh"

namespace shc {
constexpr int NA0 = 256, NE0 = 288, W0 = 320, S0R = 832, S0C = 864, S1R = 896, S1C = 928, E1R = 960, E1C = 992, A0R = 1024, A0C = 1056;
constexpr int MAJ = 1088, PAR = 1120, CE0 = 1152, CA0 = 1155, CW0 = 1158, FF0 = 1160, FFC0 = 1416, HIN0 = 1424, DG0 = 1432;
constexpr int T_FIRST = 1440, T_DATA = 1441, T_PAD = 1442, T_IDLE = 1443, COLS = 1444;
VX_HD constexpr int ST(int w, int i) { return 32 * w + i; }
VX_HD constexpr int WW(int j, int i) { return W0 + 32 * j + i; }
VX_HD constexpr int FFB(int w, int i) { return FF0 + 32 * w + i; }
#define SHC_IV_INIT {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19}
#define SHC_K_INIT {0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, \
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, \
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, \
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, \
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, \
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, \
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2}
static __device__ const uint32_t IV[8] = SHC_IV_INIT;
static __device__ const uint32_t K[64] = SHC_K_INIT;
static const uint32_t IV_H[8] = SHC_IV_INIT;
static const uint32_t K_H[64] = SHC_K_INIT;
#if defined(__HIP_DEVICE_COMPILE__)
VX_HD uint32_t iv(int i) { return IV[i]; }
#else
VX_HD uint32_t iv(int i) { return IV_H[i]; }
#endif
// words 8..15 of the single block of a 32-byte message; second block of a 64-byte message
VX_HD constexpr uint32_t tail32(int j) { return j == 0 ? 0x80000000u : (j == 7 ? 256u : 0u); }
VX_HD constexpr uint32_t pad64(int j) { return j == 0 ? 0x80000000u : (j == 15 ? 512u : 0u); }
}  // namespace shc

struct ShaAir {
    static constexpr int ID = 4, COLS = shc::COLS, PUB = 8, PERIODIC = 4, PERIOD_LOG = 6, QUOT_ROWS_PER_LANE = 1, AUX = 0, CHAL = 0, AUXPUB = 0;
    static constexpr int plog(int) { return 6; }

    template <class F, class Row, class C>
    __host__ __device__ static void eval(const Row& loc, const Row& nxt, const F* per, const F* pub, const F*, const F*, C& c) {
        using namespace shc;
        const F sel0 = per[0], sel63 = per[1], sched_on = per[2], kr = per[3];
        const F one = F::from(1), two = F::from(2), two32 = F::from(1ULL << 32);
        const F in_block = one - sel63;
        auto val = [&](const Row& row, int col0, int nb) -> F {
            F acc = row[col0 + nb - 1];
#pragma unroll 1
            for (int i = nb - 2; i >= 0; --i) acc = acc + acc + row[col0 + i];
            return acc;
        };
        // ---- 1. booleans
#pragma unroll 1
        for (int col = 0; col < HIN0; ++col) {
            const F x = loc[col];
            c.constraint(x * (x - one));
        }
        {
            const int t[4] = {T_FIRST, T_DATA, T_PAD, T_IDLE};
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                const F x = loc[t[q]];
                c.constraint(x * (x - one));
            }
        }
        c.constraint(loc[T_FIRST] + loc[T_DATA] + loc[T_PAD] + loc[T_IDLE] - one);
        // ---- 2. three-input XORs: x + y + z = r + 2c
        auto xor3 = [&](int col0, int r0, int r1, int r2, int shift, int colr, int colc) {
#pragma unroll 1
            for (int i = 0; i < 32; ++i) {
                F acc = loc[col0 + ((i + r0) & 31)] + loc[col0 + ((i + r1) & 31)];
                if (shift < 0) acc = acc + loc[col0 + ((i + r2) & 31)];
                else if (i + shift < 32) acc = acc + loc[col0 + i + shift];
                c.constraint(acc - loc[colr + i] - two * loc[colc + i]);
            }
        };
        xor3(WW(1, 0), 7, 18, 0, 3, S0R, S0C);
        xor3(WW(14, 0), 17, 19, 0, 10, S1R, S1C);
        xor3(ST(4, 0), 6, 11, 25, -1, E1R, E1C);
        xor3(ST(0, 0), 2, 13, 22, -1, A0R, A0C);
#pragma unroll 1
        for (int i = 0; i < 32; ++i) c.constraint(loc[ST(0, i)] + loc[ST(1, i)] + loc[ST(2, i)] - two * loc[MAJ + i] - loc[PAR + i]);
        // ---- 3. the round
        {
            F ch = F::from(0);
#pragma unroll 1
            for (int i = 31; i >= 0; --i) {
                const F e = loc[ST(4, i)], f = loc[ST(5, i)], g = loc[ST(6, i)];
                ch = ch + ch + (e * f + (one - e) * g);
            }
            const F t1 = val(loc, ST(7, 0), 32) + val(loc, E1R, 32) + ch + kr + val(loc, WW(0, 0), 32);
            c.constraint(val(loc, NE0, 32) + two32 * val(loc, CE0, 3) - (val(loc, ST(3, 0), 32) + t1));
            c.constraint(val(loc, NA0, 32) + two32 * val(loc, CA0, 3) - (t1 + val(loc, A0R, 32) + val(loc, MAJ, 32)));
        }
        // ---- 4. state shift inside a block
#pragma unroll 1
        for (int i = 0; i < 32; ++i) {
            c.constraint(in_block * (nxt[ST(0, i)] - loc[NA0 + i]));
            c.constraint(in_block * (nxt[ST(4, i)] - loc[NE0 + i]));
            const int wds[6] = {1, 2, 3, 5, 6, 7};
#pragma unroll 1
            for (int q = 0; q < 6; ++q) c.constraint(in_block * (nxt[ST(wds[q], i)] - loc[ST(wds[q] - 1, i)]));
        }
        // ---- 5. message schedule
#pragma unroll 1
        for (int j = 0; j < 15; ++j)
#pragma unroll 1
            for (int i = 0; i < 32; ++i) c.constraint(in_block * (nxt[WW(j, i)] - loc[WW(j + 1, i)]));
        c.constraint(sched_on * (val(nxt, WW(15, 0), 32) + two32 * val(loc, CW0, 2) -
                                 (val(loc, S1R, 32) + val(loc, WW(9, 0), 32) + val(loc, S0R, 32) + val(loc, WW(0, 0), 32))));
        // ---- 6. feed-forward at r = 63
        {
            const int s64[8] = {NA0, ST(0, 0), ST(1, 0), ST(2, 0), NE0, ST(4, 0), ST(5, 0), ST(6, 0)};
#pragma unroll 1
            for (int wd = 0; wd < 8; ++wd)
                c.constraint(sel63 * (val(loc, FFB(wd, 0), 32) + two32 * loc[FFC0 + wd] - (loc[HIN0 + wd] + val(loc, s64[wd], 32))));
        }
        // ---- 7. block boundary
        const F tdata = loc[T_DATA];
#pragma unroll 1
        for (int wd = 0; wd < 8; ++wd) {
#pragma unroll 1
            for (int i = 0; i < 32; ++i) {
                const F ivb = F::from((uint64_t)((iv(wd) >> i) & 1));
                c.constraint(sel63 * (nxt[ST(wd, i)] - (tdata * loc[FFB(wd, i)] + (one - tdata) * ivb)));
            }
            c.constraint(sel0 * (loc[HIN0 + wd] - val(loc, ST(wd, 0), 32)));
            c.constraint(in_block * (nxt[HIN0 + wd] - loc[HIN0 + wd]));
        }
        // ---- 8. block types
        {
            const int t[4] = {T_FIRST, T_DATA, T_PAD, T_IDLE};
#pragma unroll 1
            for (int q = 0; q < 4; ++q) c.constraint(in_block * (nxt[t[q]] - loc[t[q]]));
        }
        c.constraint(sel63 * (nxt[T_PAD] - tdata));
        c.transition(sel63 * nxt[T_FIRST]);
        c.first_row(loc[T_FIRST] - one);
        c.last_row(tdata);
        // ---- 9. message contents at the first row of a block
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
            c.constraint(sel0 * tdata * (val(loc, WW(j, 0), 32) - loc[DG0 + j]));
            c.constraint(sel0 * loc[T_FIRST] * (val(loc, WW(8 + j, 0), 32) - F::from(tail32(j))));
        }
#pragma unroll 1
        for (int j = 0; j < 16; ++j) c.constraint(sel0 * loc[T_PAD] * (val(loc, WW(j, 0), 32) - F::from(pad64(j))));
        // ---- 10. digest register
        const F upd = loc[T_FIRST] + loc[T_PAD];
#pragma unroll 1
        for (int wd = 0; wd < 8; ++wd) {
            const F ff = val(loc, FFB(wd, 0), 32), dg = loc[DG0 + wd];
            const F nd = upd * ff + (one - upd) * dg;
            c.constraint(in_block * (nxt[DG0 + wd] - dg));
            c.constraint(sel63 * (nxt[DG0 + wd] - nd));
            c.last_row(nd - pub[wd]);
        }
    }
};
