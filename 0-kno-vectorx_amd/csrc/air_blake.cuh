// BlakeChainAir (AIR id 6): BLAKE2b-256 parent-hash chain over a sequence of encoded headers -- the hash-chain core of
// verify_subchain (/root/reference circuits/builder/subchain_verification.rs:150-177: hash_encoded_header + parent-hash
// link; circuits/builder/header.rs:14-19 curta_blake2b_variable; block numbering decoder.rs:64-66,
// subchain_verification.rs:166-168).  The reference proves Blake2b with curta's byte-lookup STARK (starkyx v1.0.0, not
// vendored); this is a from-scratch byte-lookup arithmetisation of RFC 7693 in the same spirit: every 64-bit word is
// 8 byte cells, XORs are lookups into 2^16-row tables through a logUp argument (auxiliary commitment round), additions
// are 32-bit limb identities, rotations by 32 / 24 / 16 are byte re-indexings and the rotation by 63 is carried by a
// (low 7 bits, top bit) split of the XOR bytes.  745 main + 276 auxiliary columns (the bit-decomposed AIR it replaces
// had 4337), degree <= 3, 16 rows per compression:
//   r = 0 INIT (out-state = initial work vector), r = 1..12 ROUND (row r holds round r-1's eight G evaluations),
//   r = 13 FIN1 (U = v_lo ^ v_hi), r = 14 FIN2 (h_out = U ^ h), r = 15 PAD (H = next h_in, digest register D updated).
// Per G nine groups of 8 cells: A1 D1 C1 B1 A2 D2 C2 L T with
//   A1 = a + b + x, D1 = (d ^ A1) >>> 32, C1 = c + D1, B1 = (b ^ C1) >>> 24, A2 = A1 + B1 + y, D2 = (D1 ^ A2) >>> 16,
//   C2 = C1 + D2, (L, T) = (low 7 bits, top bit) of each byte of B1 ^ C2;  byte j of B2 = (B1 ^ C2) >>> 63 is 2 L[j] + T[j-1].
// Tables (periodic, period 2^16, row i = (a = i & 255, b = i >> 8)): T1 (a, b, a ^ b), T2 (a, b, (a ^ b) & 127, (a ^ b) >> 7).
// Bus to the SHA-256 Merkle AIR (air_sha_tree.cuh): the running sum also carries what decode_header extracts
// (decoder.rs:104-157) -- the state root, the 32 bytes right behind the SCALE compact block number (1 / 2 / 4 / 5 bytes by its
// mode, :39-92; rows 4..8 of a header's first chunk), and the data root (the last 32 bytes): byte by byte as
// (leaf, k, byte, tree) under a witness flag E per message byte, k = position - KOF with the row's window offset KOF and tree id TR.
// The table's net bus total is published as S / n (apub) and must cancel against the other table's.
// Constraint ORDER is protocol: oracle/blake_air.py restates it independently.
#pragma once
#include <type_traits>

#include "air.cuh"
#include "blake_tables.h"

namespace blk {
constexpr int S_A1 = 0, S_D1 = 1, S_C1 = 2, S_B1 = 3, S_A2 = 4, S_D2 = 5, S_C2 = 6, S_L = 7, S_T = 8;
constexpr int CAR0 = 576, MS0 = 608, MB0 = 640, HL0 = 648, D0 = 664;
constexpr int ACT = 672, FIN = 673, FIRST = 674, CAP = 675, T = 676, INC = 677, NUM = 678, FA = 679;
constexpr int TB0 = 680, IB0 = 712, MK0 = 720, CNT = 728, M1 = 729, M2 = 730, SZ = 731, E0 = 732;
// SCALE compact mode of the block number (one-hot, only on first chunks; mode 2 = FIRST - the other three) and, per row, the
// offset its bus positions are counted from and the tree its bytes go to
constexpr int MDF0 = 740, MDF1 = 741, MDF3 = 742, KOF = 743, TR = 744, COLS = 745;
// helper elements: 128 of the G functions, 4 message-byte range checks, 4 root byte sends, the table helper, the running sum
constexpr int N_HELP = 138, HM0 = 128, HB0 = 132, HT = 136, ZZ = 137, AUX = 2 * N_HELP, TABLE_LOG = 16;
// bus tuples are (t0, t1, t2, t3, tag): fingerprint t0 + g t1 + g^2 t2 + g^3 t3 + g^4 tag
constexpr int TAG_T1 = 0, TAG_T2 = 1, TAG_BYTE = 2, TAG_WORD = 3;
VX_HD constexpr int GC(int k, int slot, int j) { return (k * 9 + slot) * 8 + j; }
VX_HD constexpr int CAR(int k, int q) { return CAR0 + 4 * k + q; }
VX_HD constexpr int MS(int s, int h) { return MS0 + 2 * s + h; }
VX_HD constexpr int HL(int w, int h) { return HL0 + 2 * w + h; }
VX_HD constexpr int AX(int e, int comp) { return COLS + 2 * e + comp; }  // component of auxiliary extension element e
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

// b0 + 2^8 b1 + 2^16 b2 + 2^24 b3 for arbitrary field elements (on the LDE coset a "byte" cell is any element).
// Device: a 128-bit integer sum of shifted words and ONE reduction instead of three modular multiplications.
template <class F>
VX_HD F limb4(const F& b0, const F& b1, const F& b2, const F& b3) {
    if constexpr (is_device_field<F>::value) {
        F r;
#pragma unroll
        for (int j = 0; j < F::LANES; ++j) {
            uint64_t lo = b0.v[j], hi = 0, t;
            bool cy;
            t = b1.v[j] << 8, cy = __builtin_add_overflow(lo, t, &lo), hi += (b1.v[j] >> 56) + (cy ? 1 : 0);
            t = b2.v[j] << 16, cy = __builtin_add_overflow(lo, t, &lo), hi += (b2.v[j] >> 48) + (cy ? 1 : 0);
            t = b3.v[j] << 24, cy = __builtin_add_overflow(lo, t, &lo), hi += (b3.v[j] >> 40) + (cy ? 1 : 0);
            r.v[j] = gl_reduce128(hi, lo);
        }
        return r;
    } else {
        const F k = F::from(256);
        return ((b3 * k + b2) * k + b1) * k + b0;
    }
}
}  // namespace blk

#ifndef VX_BLAKE_QR
#define VX_BLAKE_QR 1  // LDE points per lane in the quotient kernel
#endif
struct BlakeAir {
    static constexpr int ID = 6, COLS = blk::COLS, PUB = 20, PERIODIC = 20, PERIOD_LOG = 16, QUOT_ROWS_PER_LANE = VX_BLAKE_QR, AUX = blk::AUX, CHAL = 4, AUXPUB = 1, EXACT_LOG = 0;
    static constexpr int plog(int q) { return q < 16 ? 4 : 16; }

    template <class F, class Row, class C>
    __host__ __device__ static void eval(const Row& loc, const Row& nxt, const F* per, const F* pub, const F* chal, const F* apub, C& c) {
        using namespace blk;
        const F* sel = per;
        const F one = F::from(1), two = F::from(2), two32 = F::from(1ULL << 32), inv32 = F::from(0xFFFFFFFE00000002ULL);  // 2^-32 mod p
        F g_on = sel[0];
        for (int r = 1; r < 12; ++r) g_on = g_on + sel[r];
        auto at = [&](int which, int col) -> F { return which ? nxt[col] : loc[col]; };
        // byte j of out-state word v[w] (the diagonal-step outputs) of the local (which = 0) / next row
        auto out_byte = [&](int which, int w, int j) -> F {
            const int m = w & 3;
            if (w < 4) return at(which, GC(4 + w, S_A2, j));
            if (w < 8) {
                const int k = 4 + (m + 3) % 4;
                const F l = at(which, GC(k, S_L, j));
                return l + l + at(which, GC(k, S_T, (j + 7) & 7));
            }
            if (w < 12) return at(which, GC(4 + (m + 2) % 4, S_C2, j));
            return at(which, GC(4 + (m + 1) % 4, S_D2, j));
        };
        // byte j of operand op (0 a, 1 b, 2 c, 3 d) of G number k of the round held in the next row
        auto in_byte = [&](int k, int op, int j) -> F {
            if (k < 4) return out_byte(0, 4 * op + k, j);
            const int j0 = k - 4;
            if (op == 0) return nxt[GC(j0, S_A2, j)];
            if (op == 1) {
                const int kb = (j0 + 1) & 3;
                const F l = nxt[GC(kb, S_L, j)];
                return l + l + nxt[GC(kb, S_T, (j + 7) & 7)];
            }
            if (op == 2) return nxt[GC((j0 + 2) & 3, S_C2, j)];
            return nxt[GC((j0 + 3) & 3, S_D2, j)];
        };
        auto limb_in = [&](int k, int op, int h) -> F { return limb4(in_byte(k, op, 4 * h), in_byte(k, op, 4 * h + 1), in_byte(k, op, 4 * h + 2), in_byte(k, op, 4 * h + 3)); };
        auto limb_cells = [&](int which, int k, int slot, int h) -> F {
            return limb4(at(which, GC(k, slot, 4 * h)), at(which, GC(k, slot, 4 * h + 1)), at(which, GC(k, slot, 4 * h + 2)), at(which, GC(k, slot, 4 * h + 3)));
        };
        auto boolean = [&](int col) {
            const F x = loc[col];
            c.constraint(x * (x - one));
        };
        // ---- 1. booleans
#pragma unroll 1
        for (int col = TB0; col < TB0 + 32; ++col) boolean(col);
#pragma unroll 1
        for (int col = IB0; col < IB0 + 8; ++col) boolean(col);
#pragma unroll 1
        for (int col = MK0; col < MK0 + 8; ++col) boolean(col);
        boolean(ACT);
        boolean(FIN);
        boolean(FIRST);
        boolean(CAP);
        boolean(FA);
        boolean(MDF0);
        boolean(MDF1);
        boolean(MDF3);
        const F mdf2 = loc[FIRST] - loc[MDF0] - loc[MDF1] - loc[MDF3];  // exactly one mode on a first chunk, none elsewhere
        c.constraint(mdf2 * (mdf2 - one));
        // ---- 2. carries of the three-operand additions
#pragma unroll 1
        for (int col = CAR0; col < CAR0 + 32; ++col) {
            const F x = loc[col];
            c.constraint(x * (x - one) * (x - two));
        }
        // ---- 3. additions of the eight G functions of the round held in the next row (gated by g_on)
        {
            auto GON = c.open(g_on);
#pragma unroll 1
            for (int k = 0; k < 8; ++k) {
                const int xs = k < 4 ? 2 * k : 8 + 2 * (k - 4), ys = xs + 1;
                // add3: o1 + o2 + message word -> res, explicit carry cells
                F cin = one;
                // A1 = a + b + x
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    F lhs = limb_in(k, 0, h) + limb_in(k, 1, h) + nxt[MS(xs, h)];
                    if (h) lhs = lhs + cin;
                    const F car = nxt[CAR(k, h)];
                    c.gated(GON, lhs - limb_cells(1, k, S_A1, h) - car * two32);
                    cin = car;
                }
                // C1 = c + D1 (carry = a linear expression that must be 0 or 1)
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    F t = limb_in(k, 2, h) + limb_cells(1, k, S_D1, h) - limb_cells(1, k, S_C1, h);
                    if (h) t = t + cin;
                    const F cy = t * inv32;
                    c.gated(GON, cy * (cy - one));
                    cin = cy;
                }
                // A2 = A1 + B1 + y
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    F lhs = limb_cells(1, k, S_A1, h) + limb_cells(1, k, S_B1, h) + nxt[MS(ys, h)];
                    if (h) lhs = lhs + cin;
                    const F car = nxt[CAR(k, 2 + h)];
                    c.gated(GON, lhs - limb_cells(1, k, S_A2, h) - car * two32);
                    cin = car;
                }
                // C2 = C1 + D2
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    F t = limb_cells(1, k, S_C1, h) + limb_cells(1, k, S_D2, h) - limb_cells(1, k, S_C2, h);
                    if (h) t = t + cin;
                    const F cy = t * inv32;
                    c.gated(GON, cy * (cy - one));
                    cin = cy;
                }
            }
            c.close(GON);
        }
        // ---- 4. INIT row (gated by sel[0]): out-state = (H, IV[0..4), IV4 ^ t, IV5, IV6 ^ f, IV7), limb-wise
        const F fin = loc[FIN];
        {
            auto S0 = c.open(sel[0]);
#pragma unroll 1
            for (int w = 0; w < 16; ++w)
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    const F got = limb4(out_byte(0, w, 4 * h), out_byte(0, w, 4 * h + 1), out_byte(0, w, 4 * h + 2), out_byte(0, w, 4 * h + 3));
                    F want;
                    if (w < 8) want = loc[HL(w, h)];
                    else {
                        const uint64_t ivl = (iv(w - 8) >> (32 * h)) & 0xFFFFFFFFULL;
                        if (w == 12 && h == 0) {
                            // (IV4 ^ t) low limb: sum_i 2^i (iv_i xor tb_i), linear in the bits of t
                            want = F::from(0);
#pragma unroll 1
                            for (int i = 0; i < 32; ++i) {
                                const F tb = loc[TB0 + i];
                                want = want + (((ivl >> i) & 1) ? one - tb : tb) * F::from(1ULL << i);
                            }
                        } else if (w == 14) {
                            // fin ? ~iv : iv
                            want = fin * (F::from(0xFFFFFFFFULL - ivl) - F::from(ivl)) + F::from(ivl);
                        } else want = F::from(ivl);
                    }
                    c.gated(S0, got - want);
                }
            c.close(S0);
        }
        // ---- 5. finalisation through the same-row D2 lookups of rows 13 / 14
        {
            F keep_h = sel[15];
            for (int r = 0; r < 13; ++r) keep_h = keep_h + sel[r];
#pragma unroll 1
            for (int w = 0; w < 8; ++w) {
                const uint64_t ivp = w == 0 ? (iv(0) ^ 0x01010020ULL) : iv(w);
#pragma unroll 1
                for (int i = 0; i < 8; ++i) {
                    c.constraint(sel[12] * (nxt[GC(w, S_D1, i)] - out_byte(0, w, i)));
                    c.constraint(sel[12] * (nxt[GC(w, S_A2, i)] - out_byte(0, 8 + w, i)));
                    c.constraint(sel[13] * (nxt[GC(w, S_D1, i)] - loc[GC(w, S_D2, (i + 6) & 7)]));
                }
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    const F hl = loc[HL(w, h)], hn = nxt[HL(w, h)];
                    const F ivl = F::from((ivp >> (32 * h)) & 0xFFFFFFFFULL);
                    // h_out byte i = D2[(i + 6) mod 8] of the local row
                    const F hout = limb4(loc[GC(w, S_D2, (4 * h + 6) & 7)], loc[GC(w, S_D2, (4 * h + 7) & 7)], loc[GC(w, S_D2, (4 * h + 8) & 7)],
                                         loc[GC(w, S_D2, (4 * h + 9) & 7)]);
                    c.constraint(sel[13] * (limb_cells(1, w, S_A2, h) - hl));
                    c.constraint(sel[14] * (hl - hout));
                    c.constraint(sel[14] * (hn - (fin * ivl + (one - fin) * hl)));
                    c.constraint(keep_h * (hn - hl));
                }
            }
        }
        // ---- 6. message schedule, bytes of the natural word, link to the previous digest
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
            c.constraint(acc - limb4(loc[MB0 + 4 * h], loc[MB0 + 4 * h + 1], loc[MB0 + 4 * h + 2], loc[MB0 + 4 * h + 3]));
        }
        // ---- 6b. bytes at positions >= inc are zero (RFC 7693 zero padding of the last chunk)
        const F in_blk = one - sel[15];
        {
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
            for (int b = 0; b < 8; ++b) c.constraint((one - loc[MK0 + b]) * loc[MB0 + b]);
        }
        const F first = loc[FIRST];
#pragma unroll 1
        for (int s = 0; s < 4; ++s)
#pragma unroll 1
            for (int h = 0; h < 2; ++h) c.constraint(sel[0] * first * (loc[MS(s, h)] - loc[D0 + 2 * s + h]));
        // the block number, bytes 32.. of a first chunk = the natural word of row 4: SCALE compact, all four modes (decoder.rs:39-92)
        {
            const F num = loc[NUM], k8 = F::from(256), k16 = F::from(65536), k24 = F::from(1ULL << 24), four = F::from(4);
            const F b0 = loc[MB0], b1 = loc[MB0 + 1], b2 = loc[MB0 + 2], b3 = loc[MB0 + 3], b4 = loc[MB0 + 4];
            c.constraint(sel[4] * (loc[MDF0] * (b0 - num * four) + loc[MDF1] * (b0 + b1 * k8 - num * four - one) +
                                   mdf2 * (b0 + b1 * k8 + b2 * k16 + b3 * k24 - num * four - two) + loc[MDF3] * (b1 + b2 * k8 + b3 * k16 + b4 * k24 - num)));
            c.constraint(sel[4] * loc[MDF3] * (b0 - F::from(3)));
            // the window the row's bus positions are counted from, and its tree: state root on rows 4..8 of a first chunk (right
            // behind the compact number), data root = the last 32 bytes everywhere else
            const F s48 = sel[4] + sel[5] + sel[6] + sel[7] + sel[8], srw = first * s48;
            const F clen = loc[MDF0] + loc[MDF1] * two + mdf2 * four + loc[MDF3] * F::from(5);
            c.constraint(loc[TR] - (one - srw));
            // public input 19 = bus mode: 0 nothing, 1 the roots of every header, 2 a WINDOW of message bytes from byte pub[18] on (rotate)
            const F mode = pub[19], win = mode * (mode - one) * F::from(0x7FFFFFFF80000001ULL);  // m (m - 1) / 2
            c.constraint(loc[KOF] - (s48 * (first * F::from(32) + clen) + (one - srw) * ((one - win) * (loc[SZ] - F::from(32)) + win * pub[18])));
        }
        // ---- 7. per-block registers
        {
            const int regs[11] = {ACT, FIN, FIRST, CAP, T, INC, NUM, FA, MDF0, MDF1, MDF3};
#pragma unroll 1
            for (int q = 0; q < 11; ++q) c.constraint(in_blk * (nxt[regs[q]] - loc[regs[q]]));
        }
        c.constraint(loc[CAP] - loc[ACT] * fin);
        c.constraint(loc[FA] - first * loc[ACT]);
        c.transition(sel[15] * (nxt[NUM] - loc[NUM] - nxt[FA]));  // sequential numbers (subchain_verification.rs:166-168)
        c.constraint(sel[15] * (one - fin) * (nxt[ACT] - loc[ACT]));  // ACT belongs to a whole message ...
        c.transition(nxt[ACT] * (one - loc[ACT]));                    // ... and padding stays padding
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
        // ---- 7b. message size register: constant across the chunks of a message, equal to the byte counter at its end
        c.constraint(in_blk * (nxt[SZ] - loc[SZ]));
        c.constraint(sel[15] * (one - fin) * (nxt[SZ] - loc[SZ]));
        c.constraint(fin * (loc[SZ] - loc[T]));
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
        // ---- 10. lookups (logUp): helpers live in the next row, the table side in the local row, Z closes cyclically
        {
            const X2<F> beta{chal[0], chal[1]}, gamma{chal[2], chal[3]}, g2 = gamma * gamma, g3 = g2 * gamma, g4 = g2 * g2;
            const X2<F> bt2 = beta + g4;  // table-2 tuples carry the tag gamma^4
            const F m3 = g_on + sel[12] + sel[13];
            // denominator beta + fingerprint of lookup i (a byte position) of group grp of G number k
            auto denom = [&](int k, int grp, int i) -> X2<F> {
                if (grp == 0) return beta + in_byte(k, 3, i) + gamma * nxt[GC(k, S_A1, i)] + g2 * nxt[GC(k, S_D1, (i + 4) & 7)];
                if (grp == 1) return beta + in_byte(k, 1, i) + gamma * nxt[GC(k, S_C1, i)] + g2 * nxt[GC(k, S_B1, (i + 5) & 7)];
                if (grp == 2) return beta + nxt[GC(k, S_D1, i)] + gamma * nxt[GC(k, S_A2, i)] + g2 * nxt[GC(k, S_D2, (i + 6) & 7)];
                return bt2 + nxt[GC(k, S_B1, i)] + gamma * nxt[GC(k, S_C2, i)] + g2 * nxt[GC(k, S_L, i)] + g3 * nxt[GC(k, S_T, i)];
            };
            X2<F> hsum{F::from(0), F::from(0)};
#pragma unroll 1
            for (int k = 0; k < 8; ++k)
#pragma unroll 1
                for (int grp = 0; grp < 4; ++grp)
#pragma unroll 1
                    for (int pair = 0; pair < 4; ++pair) {
                        const int e = (k * 4 + grp) * 4 + pair;
                        const X2<F> du = denom(k, grp, 2 * pair), dv = denom(k, grp, 2 * pair + 1);
                        const X2<F> h{nxt[AX(e, 0)], nxt[AX(e, 1)]};
                        c.constraint_x2(h * du * dv - (du + dv) * (grp == 2 ? m3 : g_on));
                        hsum = hsum + h;
                    }
            // the 8 bytes of the row's natural message word are range checked as (byte, 0, byte) in T1: always active
#pragma unroll 1
            for (int pair = 0; pair < 4; ++pair) {
                const int e = HM0 + pair;
                const F b0 = nxt[MB0 + 2 * pair], b1 = nxt[MB0 + 2 * pair + 1];
                const X2<F> du = beta + b0 + g2 * b0, dv = beta + b1 + g2 * b1;
                const X2<F> h{nxt[AX(e, 0)], nxt[AX(e, 1)]};
                c.constraint_x2(h * du * dv - (du + dv));
                hsum = hsum + h;
            }
            // ---- bus sends of the next row: byte b of its natural message word under the flag E[b], tuple (leaf, position in the
            // root, byte, tree) -- tree 0 = state root, tree 1 = data root; rows are numbered by the LOCAL row's selectors
            {
                F r8n = sel[0] * F::from(8);
                for (int r = 1; r < 15; ++r) r8n = r8n + sel[r] * F::from(8 * (r + 1));
                // leaves are counted from public input 18 in bus mode 1 (the first block of the whole RANGE: a table may hold one map
                // segment of it) and from the table's own first block otherwise (mode 2: one header, leaf 0)
                const F is1 = pub[19] * (two - pub[19]);  // [mode = 1]
                const F leaf = nxt[NUM] - (pub[16] + is1 * (pub[18] - pub[16])), bus_on = pub[19] * (F::from(3) - pub[19]) * F::from(0x7FFFFFFF80000001ULL);  // m (3 - m) / 2: 0 = a stand-alone proof
                const F pos0 = nxt[T] - nxt[INC] + r8n - nxt[KOF];
                const X2<F> base = beta + leaf + g3 * nxt[TR] + g4 * F::from(TAG_BYTE);
                const F live = nxt[ACT] * bus_on;  // an inactive (padding / junk) message shares its block number with the last real header: it must not send
#pragma unroll 1
                for (int pair = 0; pair < 4; ++pair) {
                    const int e = HB0 + pair, b0 = 2 * pair, b1 = b0 + 1;
                    const X2<F> du = base + gamma * (pos0 + F::from(b0)) + g2 * nxt[MB0 + b0], dv = base + gamma * (pos0 + F::from(b1)) + g2 * nxt[MB0 + b1];
                    const X2<F> h{nxt[AX(e, 0)], nxt[AX(e, 1)]};
                    c.constraint_x2(h * du * dv - dv * (nxt[E0 + b0] * live) - du * (nxt[E0 + b1] * live));
                    hsum = hsum + h;
                }
            }
            const F ta = per[16], tb_ = per[17], tl = per[18], tt = per[19];
            const X2<F> dt1 = beta + ta + gamma * tb_ + g2 * (tl + tt * F::from(128));
            const X2<F> dt2 = bt2 + ta + gamma * tb_ + g2 * tl + g3 * tt;
            const X2<F> ht{loc[AX(HT, 0)], loc[AX(HT, 1)]};
            c.constraint_x2(ht * dt1 * dt2 - dt2 * loc[M1] - dt1 * loc[M2]);
            const X2<F> z{loc[AX(ZZ, 0)], loc[AX(ZZ, 1)]}, zn{nxt[AX(ZZ, 0)], nxt[AX(ZZ, 1)]};
            c.constraint_x2(zn - z - hsum + ht + X2<F>{apub[0], apub[1]});  // apub = (net bus total of this table) / n
        }
    }
};
