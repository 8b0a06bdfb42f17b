// ShaChainAir (AIR id 4): the chained SHA-256 authority-set commitment
//   h_0 = SHA256(pk_0), h_i = SHA256(h_{i-1} || pk_i)
// of compute_authority_set_commitment (/root/reference circuits/builder/justification.rs:127-162;
// native mirror circuits/input/mod.rs:250-260).  The reference proves SHA-256 with curta's STARK
// (starkyx v1.0.0, not vendored); this AIR is a from-scratch FIPS 180-4 arithmetisation, degree <= 3:
// one row per round, 64 rows per compression; Ch as a
// degree-2 and every Sigma / sigma / Maj as degree-3 polynomials of bits; block types FIRST / DATA / PAD / IDLE.  412 (+2) columns: only the
// words an XOR / AND reads are bit-decomposed (see the layout note); the compression rows are shared with ShaTreeAir.
// Bus to EdAir (air_ed.cuh): a key's block carries a witness flag SGC ("this authority signed"); the key is sent as four
// tuples (4 index + j, l0 + 2^16 l1, l2 + 2^16 l3, 0, TAG_KEY) of little-endian 16-bit limbs from the rows where its words
// 2j / 2j+1 sit in window positions 0 / 1 (rows 0, 2, 4, 6 of FIRST, rows 8, 10, 12, 14 of DATA); index = key counter KC - 1,
// and KC ends as the number of committed keys (public input 8; the denominator of the 2/3 threshold, justification.rs:164-186).
// Constraint ORDER is protocol: oracle/sha_air.py restates it independently.
#pragma once
#include <vector>

#include "air.cuh"

namespace shc {
// Column layout.  Only the words an XOR / AND needs exist as 32 little-endian bit columns: a, b, c, e, f, g of the state
// (Sigma0, Maj / Sigma1, Ch), the new a and e, and positions 0, 1, 14 of the 16-word schedule window (w_r enters T1 and a
// bus, sigma0 reads w_{r+1}, sigma1 reads w_{r+14}).  d, h and the other 13 window positions are single VALUE columns:
// every such value was, or will be, a bit-decomposed word on another row, and everything downstream works modulo 2^32.
// Sigma0 / Sigma1 / Ch / Maj have no cells: they are degree-3 polynomials of the state bits inside the (unconditional) round
// equations.  The schedule's equation carries a selector, so sigma0(w_{r+1}) + sigma1(w_{r+14}) gets ONE value cell SV, itself
// defined by an unconditional degree-3 polynomial identity.
constexpr int A_ = 0, B_ = 32, C_ = 64, E_ = 96, F_ = 128, G_ = 160, DV = 192, HV = 193, NA0 = 194, NE0 = 226;
constexpr int W0B = 258, W1B = 290, W14B = 322, WV0 = 354, WV15 = 366;
constexpr int SV = 367;  // sigma0(W[1]) + sigma1(W[14]), a value below 2^33
constexpr int CE0 = 368, CA0 = 371, CW0 = 374, FFV0 = 376, FFC0 = 384, HIN0 = 392, DG0 = 400;
constexpr int T_FIRST = 408, T_DATA = 409, T_PAD = 410, T_IDLE = 411, COLS = 412;  // COLS: the compression layout every SHA-256 table shares
constexpr int SGC = 412, KC = 413, CHAIN_COLS = 414, TAG_KEY = 5;                    // ShaChainAir only
VX_HD constexpr int WV(int p) { return p == 15 ? WV15 : WV0 + p - 2; }  // value column of window position p (2..13, 15)
VX_HD constexpr int st_bits(int wd) { return wd == 0 ? A_ : wd == 1 ? B_ : wd == 2 ? C_ : wd == 4 ? E_ : wd == 5 ? F_ : wd == 6 ? G_ : -1; }
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

// Sections 1-7 shared by every SHA-256 table: the rows of a compression and the hand-over at a block boundary.
// data_flag: 1 where the block AFTER the local one continues from its output (DATA -> PAD), else it starts from IV.
template <class F, class Row, class C>
__host__ __device__ inline void sha_compression_constraints(const Row& loc, const Row& nxt, const F* per, const F& data_flag, C& c) {
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
    auto window = [&](const Row& row, int p) -> F { return p == 0 ? val(row, W0B, 32) : p == 1 ? val(row, W1B, 32) : p == 14 ? val(row, W14B, 32) : row[WV(p)]; };
    auto state_word = [&](const Row& row, int wd) -> F { return wd == 3 ? row[DV] : wd == 7 ? row[HV] : val(row, st_bits(wd), 32); };
    // ---- 1. booleans
    {
        const int lo[4] = {0, NA0, CE0, FFC0}, hi[4] = {DV, WV0, FFV0, HIN0};
#pragma unroll 1
        for (int q = 0; q < 4; ++q)
#pragma unroll 1
            for (int col = lo[q]; col < hi[q]; ++col) {
                const F x = loc[col];
                c.constraint(x * (x - one));
            }
    }
    // ---- 2. SV = sigma0(W[1]) + sigma1(W[14]): XORs as polynomials of the window bits (shifted-out bits are absent)
    {
        const F four = F::from(4);
        auto sig = [&](int col0, int r0, int r1, int shift) -> F {
            F acc = F::from(0);
#pragma unroll 1
            for (int i = 31; i >= 0; --i) {
                const F x = loc[col0 + ((i + r0) & 31)], y = loc[col0 + ((i + r1) & 31)];
                const F xy = x * y;
                if (i + shift >= 32) acc = acc + acc + (x + y - two * xy);
                else {
                    const F z = loc[col0 + i + shift];
                    acc = acc + acc + (x + y + z - two * (xy + (x + y) * z) + four * (xy * z));
                }
            }
            return acc;
        };
        c.constraint(loc[SV] - sig(W1B, 7, 18, 3) - sig(W14B, 17, 19, 10));
    }
    // ---- 3. the round (local, every row): Sigma / Ch / Maj as degree-3 polynomials of the state bits
    {
        const F four = F::from(4);
        auto x3 = [&](int col0, int r0, int r1, int r2) -> F {
            F acc = F::from(0);
#pragma unroll 1
            for (int i = 31; i >= 0; --i) {
                const F x = loc[col0 + ((i + r0) & 31)], y = loc[col0 + ((i + r1) & 31)], z = loc[col0 + ((i + r2) & 31)];
                const F xy = x * y;
                acc = acc + acc + (x + y + z - two * (xy + (x + y) * z) + four * (xy * z));
            }
            return acc;
        };
        F ch = F::from(0), mj = F::from(0);
#pragma unroll 1
        for (int i = 31; i >= 0; --i) {
            const F e = loc[E_ + i], f = loc[F_ + i], g = loc[G_ + i];
            ch = ch + ch + (e * f + (one - e) * g);
        }
#pragma unroll 1
        for (int i = 31; i >= 0; --i) {
            const F a = loc[A_ + i], b = loc[B_ + i], cc = loc[C_ + i];
            const F ab = a * b;
            mj = mj + mj + (ab + (a + b) * cc - two * (ab * cc));
        }
        const F t1 = loc[HV] + x3(E_, 6, 11, 25) + ch + kr + val(loc, W0B, 32);
        c.constraint(val(loc, NE0, 32) + two32 * val(loc, CE0, 3) - (loc[DV] + t1));
        c.constraint(val(loc, NA0, 32) + two32 * val(loc, CA0, 3) - (t1 + x3(A_, 2, 13, 22) + mj));
    }
    // ---- 4. state shift inside a block
#pragma unroll 1
    for (int i = 0; i < 32; ++i) {
        c.constraint(in_block * (nxt[A_ + i] - loc[NA0 + i]));
        c.constraint(in_block * (nxt[E_ + i] - loc[NE0 + i]));
        c.constraint(in_block * (nxt[B_ + i] - loc[A_ + i]));
        c.constraint(in_block * (nxt[C_ + i] - loc[B_ + i]));
        c.constraint(in_block * (nxt[F_ + i] - loc[E_ + i]));
        c.constraint(in_block * (nxt[G_ + i] - loc[F_ + i]));
    }
    c.constraint(in_block * (nxt[DV] - val(loc, C_, 32)));
    c.constraint(in_block * (nxt[HV] - val(loc, G_, 32)));
    // ---- 5. message schedule: window shift, and w_{r+16} while r <= 47
#pragma unroll 1
    for (int i = 0; i < 32; ++i) c.constraint(in_block * (nxt[W0B + i] - loc[W1B + i]));
#pragma unroll 1
    for (int p = 1; p < 15; ++p) c.constraint(in_block * (window(nxt, p) - window(loc, p + 1)));
    c.constraint(sched_on * (nxt[WV15] + two32 * val(loc, CW0, 2) - (loc[SV] + loc[WV(9)] + val(loc, W0B, 32))));
    // ---- 6. feed-forward at r = 63: FF = H_in + (NA, a, b, c, NE, e, f, g)
    {
        const int s64[8] = {NA0, A_, B_, C_, NE0, E_, F_, G_};
#pragma unroll 1
        for (int wd = 0; wd < 8; ++wd) c.constraint(sel63 * (loc[FFV0 + wd] + two32 * loc[FFC0 + wd] - (loc[HIN0 + wd] + val(loc, s64[wd], 32))));
    }
    // ---- 7. block boundary: next start state = FF where data_flag, IV otherwise; H_in register
#pragma unroll 1
    for (int wd = 0; wd < 8; ++wd) {
        const int sb = st_bits(wd);
        if (sb >= 0) {
#pragma unroll 1
            for (int i = 0; i < 32; ++i) c.constraint(sel63 * (one - data_flag) * (nxt[sb + i] - F::from((uint64_t)((iv(wd) >> i) & 1))));
            c.constraint(sel63 * data_flag * (val(nxt, sb, 32) - loc[FFV0 + wd]));
        } else {
            c.constraint(sel63 * (state_word(nxt, wd) - (data_flag * loc[FFV0 + wd] + (one - data_flag) * F::from((uint64_t)iv(wd)))));
        }
        c.constraint(sel0 * (loc[HIN0 + wd] - state_word(loc, wd)));
        c.constraint(in_block * (nxt[HIN0 + wd] - loc[HIN0 + wd]));
    }
}

struct ShaAir {
    static constexpr int ID = 4, COLS = shc::CHAIN_COLS, PUB = 10, PERIODIC = 7, PERIOD_LOG = 6, QUOT_ROWS_PER_LANE = 1, AUX = 4, CHAL = 4, AUXPUB = 1, EXACT_LOG = 0;
    static constexpr int plog(int) { return 6; }
    static void periodic_values(std::vector<uint64_t>& v) {
        v.assign(7 * 64, 0);
        v[0] = 1;                                      // sel_0
        v[64 + 63] = 1;                                // sel_63
        for (int r = 0; r <= 47; ++r) v[128 + r] = 1;  // schedule active
        for (int r = 0; r < 64; ++r) v[192 + r] = shc::K_H[r];
        for (int r = 0; r < 16; r += 2) v[(r < 8 ? 256 : 320) + r] = 1, v[384 + r] = (r & 7) >> 1;  // key-send rows of FIRST / DATA blocks, j
    }

    template <class F, class Row, class C>
    __host__ __device__ static void eval(const Row& loc, const Row& nxt, const F* per, const F* pub, const F* chal, const F* apub, C& c) {
        using namespace shc;
        const F sel0 = per[0], sel63 = per[1];
        const F one = F::from(1);
        const F in_block = one - sel63;
        auto val = [&](const Row& row, int col0, int nb) -> F {
            F acc = row[col0 + nb - 1];
#pragma unroll 1
            for (int i = nb - 2; i >= 0; --i) acc = acc + acc + row[col0 + i];
            return acc;
        };
        auto window = [&](const Row& row, int p) -> F { return p == 0 ? val(row, W0B, 32) : p == 1 ? val(row, W1B, 32) : p == 14 ? val(row, W14B, 32) : row[WV(p)]; };
        // ---- 0. the four type flags
        const int t[4] = {T_FIRST, T_DATA, T_PAD, T_IDLE};
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            const F x = loc[t[q]];
            c.constraint(x * (x - one));
        }
        c.constraint(loc[T_FIRST] + loc[T_DATA] + loc[T_PAD] + loc[T_IDLE] - one);
        const F tdata = loc[T_DATA];
        sha_compression_constraints<F>(loc, nxt, per, tdata, c);
        // ---- 8. block types
#pragma unroll 1
        for (int q = 0; q < 4; ++q) c.constraint(in_block * (nxt[t[q]] - loc[t[q]]));
        c.constraint(sel63 * (nxt[T_PAD] - tdata));
        c.transition(sel63 * nxt[T_FIRST]);
        c.first_row(loc[T_FIRST] - one);
        c.last_row(tdata);
        // ---- 9. message contents at the first row of a block
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
            c.constraint(sel0 * tdata * (window(loc, j) - loc[DG0 + j]));
            c.constraint(sel0 * loc[T_FIRST] * (window(loc, 8 + j) - F::from(tail32(j))));
        }
#pragma unroll 1
        for (int j = 0; j < 16; ++j) c.constraint(sel0 * loc[T_PAD] * (window(loc, j) - F::from(pad64(j))));
        // ---- 10. digest register
        const F upd = loc[T_FIRST] + loc[T_PAD];
#pragma unroll 1
        for (int wd = 0; wd < 8; ++wd) {
            const F ff = loc[FFV0 + wd], dg = loc[DG0 + wd];
            const F nd = upd * ff + (one - upd) * dg;
            c.constraint(in_block * (nxt[DG0 + wd] - dg));
            c.constraint(sel63 * (nxt[DG0 + wd] - nd));
            c.last_row(nd - pub[wd]);
        }
        // ---- 11. the "signed" flag of a key's block and the key counter
        const F sgc = loc[SGC], kc = loc[KC];
        c.constraint(sgc * (sgc - one));
        c.constraint(in_block * (nxt[SGC] - sgc));
        c.constraint(sgc * (loc[T_PAD] + loc[T_IDLE]));
        c.constraint(in_block * (nxt[KC] - kc));
        c.transition(sel63 * (nxt[KC] - kc - nxt[T_DATA]));
        c.first_row(kc - one);
        c.last_row(kc - pub[8]);
        // ---- 12. the bus: signed keys go to EdAir
        {
            const X2<F> beta{chal[0], chal[1]}, gamma{chal[2], chal[3]}, g2 = gamma * gamma, g4 = g2 * g2;
            const F k8 = F::from(256), k16 = F::from(65536);
            auto limbs = [&](int col0) -> F {  // bytes b0 b1 b2 b3 of the big-endian word: (b0 + 256 b1) + 2^16 (b2 + 256 b3)
                return val(loc, col0 + 24, 8) + val(loc, col0 + 16, 8) * k8 + (val(loc, col0 + 8, 8) + val(loc, col0, 8) * k8) * k16;
            };
            // public input 9 = bus mode: 0 nothing, 1 SEND the flagged keys (to EdAir), 2 RECEIVE every key (from the epoch-end table)
            const F mode = pub[9], inv2 = F::from(0x7FFFFFFF80000001ULL), on = mode * (F::from(3) - mode) * inv2, rcv = mode * (mode - one) * inv2;
            const F m = on * (per[4] * loc[T_FIRST] + per[5] * tdata) * (sgc * (one - rcv) - rcv);
            const X2<F> d = beta + ((kc - one) * F::from(4) + per[6]) + gamma * limbs(W0B) + g2 * limbs(W1B) + g4 * F::from(TAG_KEY);
            const X2<F> h{loc[CHAIN_COLS], loc[CHAIN_COLS + 1]}, z{loc[CHAIN_COLS + 2], loc[CHAIN_COLS + 3]}, zn{nxt[CHAIN_COLS + 2], nxt[CHAIN_COLS + 3]};
            c.constraint_x2(h * d - m);
            c.constraint_x2(zn - z - h + X2<F>{apub[0], apub[1]});
        }
    }
};
