// EdAir (AIR ids 10 / 12 for 2^17 / 2^16 rows): Ed25519 verification of the justification's signed precommits --
// "[S_i] B = R_i + [h_i] A_i, h_i = H_i mod l" for every signed slot i: the curve half of verify_simple_justification's
// conditional signature checks (/root/reference circuits/builder/justification.rs:229-243, native mirror
// circuits/input/mod.rs:241-247).  The reference proves this with curta's EdDSA gadget (starkyx v1.0.0, not vendored);
// this is a from-scratch arithmetisation.  Field elements mod q = 2^255 - 19 are 16 limbs of 16 bits; a GADGET proves
// c = a * b (mod q) for limb vectors a, b that are LINEAR in trace cells:
//     F_k = sum_{i+j=k} a_i b_j + 38 sum_{i+j=k+16} a_i b_j,    F_k + rin_k - c_k = 2^16 r_k,  rin_0 = 38 r_15, rin_k = r_(k-1)
// (sums to F(2^16) - c = 2 q r_15); c_k and both halves of r_k = rlo + 2^16 rhi - 2^31 are 16-bit cells range-checked
// by logUp against a periodic table 0..65535, so every term stays below 2^49 and the identity holds over the integers.
// A ZERO-CHECK is the gadget without c on twice the expression.  14 gadgets (672 cells, 672 lookups) per row.
// A slot is 256 rows: row 0 SETUP-A (A on the curve, canonical, sign; -A and B - A in cached form), row 1 SETUP-B
// (H = qq l + hr, hr < l; accumulator := identity), rows 2..254 STEP (Q' = 2Q + addend selected by the bits of S and
// hr; dbl-2008-hwcd + madd-2008-hwcd-3), row 255 FINAL ((xR Z, yR Z) = (X, Y), canonical, sign).  Slots are compact: slot s
// verifies the s-th chosen signature, whose authority index is the slot register AIDX (2/3 of 300 fit 2^16 rows).  Layout, bus tuples
// and the constraint ORDER (protocol) are restated independently in oracle/ed_air.py -- read its header for the map.
#pragma once
#include <vector>

#include "air.cuh"
#include "ed25519_constants.h"

namespace edc {
constexpr int NG = 14, CELLS = NG * 48;
constexpr int XA0 = 672, YA0 = 688, NT0 = 704, X30 = 720, Y30 = 736, BT0 = 752, HR0 = 768, SEL0 = 784;
constexpr int BS = 832, BH = 833, LAH = 834, SG = 835, CNT = 836, MULT = 837, AIDX = 838, COLS = 839;
constexpr int N_RANGE = CELLS / 2, N_BUS = 6, HB0 = N_RANGE, HT = N_RANGE + N_BUS, ZZ = HT + 1, N_HELP = ZZ + 1, AUX = 2 * N_HELP;
constexpr int TAG_R16 = 4, TAG_KEY = 5, TAG_EDMSG = 6, TAG_EDH = 7;
enum { P_S0N, P_S1N, P_STN, P_FINN, P_KEEP, P_STEP, P_LST, P_R0, P_R1, P_R255, P_LE0, P_SLOT = 26, P_T = 27, N_PERIODIC = 28 };
VX_HD constexpr int C(int g, int k) { return g * 48 + k; }
VX_HD constexpr int RL(int g, int k) { return g * 48 + 16 + k; }
VX_HD constexpr int RH(int g, int k) { return g * 48 + 32 + k; }
VX_HD constexpr int AX(int e, int comp) { return COLS + 2 * e + comp; }
// SETUP-B: cell of digest byte j (gadget 5, then gadget 6's c cells) and of 256 * byte j (gadget 6's rl / rh, gadget 7's c / rl)
VX_HD constexpr int BYA(int j) { return j < 48 ? 5 * 48 + j : C(6, j - 48); }
VX_HD constexpr int BYB(int j) { return j < 32 ? 6 * 48 + 16 + j : 7 * 48 + (j - 32); }
// constant limb vectors: 2d, -2d, d, -d xB yB, xB, yB, the base point in cached form (y - x, y + x, 2d x y), q - 1, l - 1, l
enum { K_2D, K_2DN, K_D, K_BD, K_XB, K_YB, K_BC0, K_BC1, K_BC2, K_QM1, K_LM1, K_LL, N_CONST };
#define EDA_TABLE_INIT {EDA_K2D_INIT, EDA_K2DN_INIT, EDA_KD_INIT, EDA_KBD_INIT, EDA_XB_INIT, EDA_YB_INIT, EDA_BC0_INIT, EDA_BC1_INIT, EDA_BC2_INIT, \
                        EDA_QM1_INIT, EDA_LM1_INIT, EDA_LL_INIT}
static __device__ const uint16_t KT[N_CONST][16] = EDA_TABLE_INIT;
static const uint16_t KT_H[N_CONST][16] = EDA_TABLE_INIT;
#if defined(__HIP_DEVICE_COMPILE__)
VX_HD uint64_t kc(int t, int k) { return KT[t][k]; }
#else
VX_HD uint64_t kc(int t, int k) { return KT_H[t][k]; }
#endif

// out_k = sum_{i+j=k} a_i b_j + 38 sum_{i+j=k+16} a_i b_j.  Device: sixteen 160-bit accumulations and one reduction per
// coefficient (a call, not inlined: an evaluation makes 31 of them).
template <class F>
__host__ __device__ __attribute__((noinline)) void fold16(const F* a, const F* b, F* out) {
    if constexpr (is_device_field<F>::value) {
#pragma unroll 1
        for (int l = 0; l < F::LANES; ++l) {
            uint64_t b38[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) b38[j] = gl_mul_small(b[j].v[l], 38);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                gl_acc acc;
                gl_acc_zero(acc);
#pragma unroll
                for (int i = 0; i < 16; ++i) gl_mac(acc, a[i].v[l], i <= k ? b[k - i].v[l] : b38[k - i + 16]);
                out[k].v[l] = gl_acc_reduce(acc);
            }
        }
    } else {
        const F k38 = F::from(38);
        for (int k = 0; k < 16; ++k) {
            F acc = F::from(0);
            for (int i = 0; i < 16; ++i) acc = acc + (i <= k ? a[i] * b[k - i] : a[i] * b[k - i + 16] * k38);
            out[k] = acc;
        }
    }
}
}  // namespace edc

template <int LOGN, int ID_>
struct EdAirT {
    static constexpr int ID = ID_, COLS = edc::COLS, PUB = 2, PERIODIC = edc::N_PERIODIC, PERIOD_LOG = LOGN, QUOT_ROWS_PER_LANE = 1, AUX = edc::AUX, CHAL = 4, AUXPUB = 1, EXACT_LOG = 1;
    static constexpr int plog(int q) { return q < edc::P_SLOT ? 8 : (q == edc::P_SLOT ? LOGN : 16); }

    static void periodic_values(std::vector<uint64_t>& v) {
        using namespace edc;
        const size_t n = (size_t)1 << LOGN;
        v.assign(26 * 256 + n + 65536, 0);
        uint64_t* p = v.data();
        for (int r = 0; r < 256; ++r) {
            const int nb = 252 - (r + 1 - 2);  // the next row's scalar bit
            const bool stn = r >= 1 && r <= 253;
            p[P_S0N * 256 + r] = r == 255, p[P_S1N * 256 + r] = r == 0, p[P_STN * 256 + r] = stn, p[P_FINN * 256 + r] = r == 254;
            p[P_KEEP * 256 + r] = r != 255, p[P_STEP * 256 + r] = r >= 2 && r <= 254;
            p[P_LST * 256 + r] = stn && (nb == 252 || (nb & 15) == 15);
            p[P_R0 * 256 + r] = r == 0, p[P_R1 * 256 + r] = r == 1, p[P_R255 * 256 + r] = r == 255;
            for (int k = 0; k < 16; ++k) p[(P_LE0 + k) * 256 + r] = stn && nb == 16 * k;
        }
        for (size_t i = 0; i < n; ++i) p[26 * 256 + i] = i >> 8;
        for (size_t i = 0; i < 65536; ++i) p[26 * 256 + n + i] = i;
    }

    template <class F, class Row, class Cn>
    __host__ __device__ static void eval(const Row& loc, const Row& nxt, const F* per, const F* pub, const F* chal, const F* apub, Cn& c) {
        using namespace edc;
        const F one = F::from(1), zero = F::from(0), two = F::from(2), k16 = F::from(65536), k31 = F::from(1ULL << 31), k38 = F::from(38);
        const F s0n = per[P_S0N], s1n = per[P_S1N], stn = per[P_STN], finn = per[P_FINN], keep = per[P_KEEP];
        auto kv = [&](int t, int k) -> F { return F::from(kc(t, k)); };
        auto cells = [&](const Row& row, int g, F* out) {
#pragma unroll 1
            for (int k = 0; k < 16; ++k) out[k] = row[C(g, k)];
        };
        auto colv = [&](const Row& row, int col0, F* out) {
#pragma unroll 1
            for (int k = 0; k < 16; ++k) out[k] = row[col0 + k];
        };
        auto konst = [&](int t, F* out) {
#pragma unroll 1
            for (int k = 0; k < 16; ++k) out[k] = kv(t, k);
        };
        // ---- 1. booleans
        {
            const int cols[3] = {BS, BH, SG};
#pragma unroll 1
            for (int q = 0; q < 3; ++q) {
                const F x = loc[cols[q]];
                c.constraint(x * (x - one));
            }
        }
        // ---- 2. the 14 gadgets of the next row, by its type
        // gadget g: st = its F coefficients as a STEP row (always), s0 as SETUP-A (or null), fn as FINAL (or null)
        auto emit = [&](int g, const F* st, const F* s0, bool s0_has_c, const F* fn) {
            F r[16];
#pragma unroll 1
            for (int k = 0; k < 16; ++k) r[k] = nxt[RL(g, k)] + nxt[RH(g, k)] * k16 - k31;
#pragma unroll 1
            for (int k = 0; k < 16; ++k) {
                const F tail = (k == 0 ? r[15] * k38 : r[k - 1]) - r[k] * k16;
                const F cg = nxt[C(g, k)];
                F acc = stn * (st[k] + tail - cg);
                if (s0) acc = acc + s0n * (s0_has_c ? s0[k] + tail - cg : s0[k] + tail);
                if (fn) acc = acc + finn * (fn[k] + tail);
                c.constraint(acc);
            }
        };
        {
            F X[16], Y[16], Z[16], a[16], b[16], st[16], s0[16], fn[16], t1[16], t2[16];
            cells(loc, 11, X), cells(loc, 12, Y), cells(loc, 13, Z);
            // SETUP-A cells of the next row: xA = c5, yA = c7, x3 = c8, y3 = c11, u = c0, xx = c2, yy = c3, dxx = c4, t = c6, v = c9
            // g0: X X | xA yA | 2 (xR Z - X), xR = c0
            fold16<F>(X, X, st);
            cells(nxt, 5, a), cells(nxt, 7, b);
            fold16<F>(a, b, s0);
            cells(nxt, 0, a);
            fold16<F>(a, Z, fn);
#pragma unroll 1
            for (int k = 0; k < 16; ++k) fn[k] = (fn[k] - X[k]) * two;
            emit(0, st, s0, true, fn);
            // g1: Y Y | u K2DN | 2 (yR Z - Y), yR = c1
            fold16<F>(Y, Y, st);
            cells(nxt, 0, a), konst(K_2DN, b);
            fold16<F>(a, b, s0);
            cells(nxt, 1, a);
            fold16<F>(a, Z, fn);
#pragma unroll 1
            for (int k = 0; k < 16; ++k) fn[k] = (fn[k] - Y[k]) * two;
            emit(1, st, s0, true, fn);
            // g2: Z Z | xA xA
            fold16<F>(Z, Z, st);
            cells(nxt, 5, a);
            fold16<F>(a, a, s0);
            emit(2, st, s0, true, nullptr);
            // g3: (X + Y)^2 | yA yA
#pragma unroll 1
            for (int k = 0; k < 16; ++k) a[k] = X[k] + Y[k];
            fold16<F>(a, a, st);
            cells(nxt, 7, a);
            fold16<F>(a, a, s0);
            emit(3, st, s0, true, nullptr);
            // E = c3 - c0 - c1, G = c1 - c0, Fd = G - 2 c2, H = -c0 - c1 (t1 = E, t2 = H kept for g6)
            F Gd[16], Fd[16];
#pragma unroll 1
            for (int k = 0; k < 16; ++k) {
                const F c0 = nxt[C(0, k)], c1 = nxt[C(1, k)], c2 = nxt[C(2, k)], c3 = nxt[C(3, k)];
                t1[k] = c3 - c0 - c1, Gd[k] = c1 - c0, Fd[k] = Gd[k] - c2 - c2, t2[k] = zero - c0 - c1;
            }
            // g4: E F | xx KD
            fold16<F>(t1, Fd, st);
            cells(nxt, 2, a), konst(K_D, b);
            fold16<F>(a, b, s0);
            emit(4, st, s0, true, nullptr);
            // g5: G H | zero-check 2 (yy - xx - 1 - dxx yy)
            fold16<F>(Gd, t2, st);
            cells(nxt, 4, a), cells(nxt, 3, b);
            fold16<F>(a, b, s0);
#pragma unroll 1
            for (int k = 0; k < 16; ++k) s0[k] = (b[k] - nxt[C(2, k)] - (k == 0 ? one : zero) - s0[k]) * two;
            emit(5, st, s0, false, nullptr);
            // g6: E H | u KBD
            fold16<F>(t1, t2, st);
            cells(nxt, 0, a), konst(K_BD, b);
            fold16<F>(a, b, s0);
            emit(6, st, s0, true, nullptr);
            // g7: F G | zero-check 2 (x3 (1 + t) - yA xB + xA yB)
            fold16<F>(Fd, Gd, st);
            {
                F xa[16], ya[16], tt[16], xb[16], yb[16], f2[16], f3[16];
                cells(nxt, 5, xa), cells(nxt, 7, ya), cells(nxt, 6, tt), konst(K_XB, xb), konst(K_YB, yb);
                cells(nxt, 8, a);
#pragma unroll 1
                for (int k = 0; k < 16; ++k) b[k] = (k == 0 ? one : zero) + tt[k];
                fold16<F>(a, b, s0);
                fold16<F>(ya, xb, f2);
                fold16<F>(xa, yb, f3);
#pragma unroll 1
                for (int k = 0; k < 16; ++k) s0[k] = (s0[k] - f2[k] + f3[k]) * two;
                emit(7, st, s0, false, nullptr);
                // g8: (c5 - c4) sel_ym | zero-check 2 (y3 (1 - t) - yA yB + xA xB)
#pragma unroll 1
                for (int k = 0; k < 16; ++k) a[k] = nxt[C(5, k)] - nxt[C(4, k)];
                colv(nxt, SEL0, b);
                fold16<F>(a, b, st);
                cells(nxt, 11, a);
#pragma unroll 1
                for (int k = 0; k < 16; ++k) b[k] = (k == 0 ? one : zero) - tt[k];
                fold16<F>(a, b, s0);
                fold16<F>(ya, yb, f2);
                fold16<F>(xa, xb, f3);
#pragma unroll 1
                for (int k = 0; k < 16; ++k) s0[k] = (s0[k] - f2[k] + f3[k]) * two;
                emit(8, st, s0, false, nullptr);
            }
            // g9: (c5 + c4) sel_yp | x3 y3
#pragma unroll 1
            for (int k = 0; k < 16; ++k) a[k] = nxt[C(5, k)] + nxt[C(4, k)];
            colv(nxt, SEL0 + 16, b);
            fold16<F>(a, b, st);
            cells(nxt, 8, a), cells(nxt, 11, b);
            fold16<F>(a, b, s0);
            emit(9, st, s0, true, nullptr);
            // g10: c6 sel_t2d | v K2D
            cells(nxt, 6, a), colv(nxt, SEL0 + 32, b);
            fold16<F>(a, b, st);
            cells(nxt, 9, a), konst(K_2D, b);
            fold16<F>(a, b, s0);
            emit(10, st, s0, true, nullptr);
            // D = 2 c7, E' = c9 - c8, F' = D - c10, G' = D + c10, H' = c9 + c8
#pragma unroll 1
            for (int k = 0; k < 16; ++k) {
                const F c7 = nxt[C(7, k)], c8 = nxt[C(8, k)], c9 = nxt[C(9, k)], c10 = nxt[C(10, k)], dd = c7 + c7;
                t1[k] = c9 - c8, Fd[k] = dd - c10, Gd[k] = dd + c10, t2[k] = c9 + c8;
            }
            fold16<F>(t1, Fd, st);
            emit(11, st, nullptr, false, nullptr);
            fold16<F>(Gd, t2, st);
            emit(12, st, nullptr, false, nullptr);
            fold16<F>(Fd, Gd, st);
            emit(13, st, nullptr, false, nullptr);
        }
        // x + w = top as 16 limb identities with the 15 carry cells cy (booleans): constraint k under the selector
        auto canonical = [&](const F& sel, auto&& x, auto&& w, auto&& cy, int top) {
#pragma unroll 1
            for (int k = 0; k < 16; ++k) {
                F e = x(k) + w(k) - kv(top, k);
                if (k) e = e + cy(k - 1);
                if (k < 15) e = e - cy(k) * k16;
                c.constraint(sel * e);
            }
#pragma unroll 1
            for (int k = 0; k < 15; ++k) {
                const F b = cy(k);
                c.constraint(sel * b * (b - one));
            }
        };
        // ---- 3. SETUP-A extras: canonical xA, yA; the sign bit; the carried columns take their values
        canonical(s0n, [&](int k) { return nxt[C(5, k)]; }, [&](int k) { return nxt[C(12, k)]; }, [&](int k) { return nxt[RL(12, k)]; }, K_QM1);
        canonical(s0n, [&](int k) { return nxt[C(7, k)]; }, [&](int k) { return nxt[C(13, k)]; }, [&](int k) { return nxt[RL(13, k)]; }, K_QM1);
        c.constraint(s0n * (nxt[C(5, 0)] - nxt[RH(12, 0)] * two - nxt[BS]));
        {
            const int dst[6] = {XA0, YA0, NT0, X30, Y30, BT0}, src[6] = {5, 7, 1, 8, 11, 10};
#pragma unroll 1
            for (int q = 0; q < 6; ++q)
#pragma unroll 1
                for (int k = 0; k < 16; ++k) c.constraint(s0n * (nxt[dst[q] + k] - nxt[C(src[q], k)]));
        }
        // ---- 4. SETUP-B: H = qq l + hr with hr < l; the accumulator starts at the identity
        {
            auto hl = [&](int k) -> F { return k < 16 ? nxt[C(0, k)] : nxt[RL(0, k - 16)]; };
            auto qq = [&](int k) -> F { return k < 16 ? nxt[RH(0, k)] : nxt[C(1, 0)]; };
            auto cr = [&](int k) -> F { return (k < 16 ? nxt[C(2, k)] : nxt[RL(2, k - 16)]) + (k < 16 ? nxt[C(3, k)] : nxt[RL(3, k - 16)]) * k16 - k31; };
#pragma unroll 1
            for (int k = 0; k < 33; ++k) {
                F e = zero;
#pragma unroll 1
                for (int i = 0; i < 17; ++i) {
                    const int j = k - i;
                    if (j >= 0 && j < 16 && kc(K_LL, j)) e = e + qq(i) * kv(K_LL, j);
                }
                if (k < 16) e = e + nxt[RL(1, k)];
                if (k < 32) e = e - hl(k) - cr(k) * k16;
                if (k) e = e + cr(k - 1);
                c.constraint(s1n * e);
            }
            canonical(s1n, [&](int k) { return nxt[RL(1, k)]; }, [&](int k) { return nxt[RH(1, k)]; }, [&](int k) { return nxt[C(4, k)]; }, K_LM1);
#pragma unroll 1
            for (int k = 0; k < 16; ++k) c.constraint(s1n * (nxt[HR0 + k] - nxt[RL(1, k)]));
#pragma unroll 1
            for (int g = 11; g < 14; ++g)
#pragma unroll 1
                for (int k = 0; k < 16; ++k) c.constraint(s1n * (nxt[C(g, k)] - ((g > 11 && k == 0) ? one : zero)));
            const F k8 = F::from(256);
#pragma unroll 1
            for (int j = 0; j < 64; ++j) c.constraint(s1n * (nxt[BYB(j)] - nxt[BYA(j)] * k8));  // b and 256 b are 16-bit cells: b is a byte
#pragma unroll 1
            for (int k = 0; k < 32; ++k) c.constraint(s1n * (hl(k) - nxt[BYA(2 * k)] - nxt[BYA(2 * k + 1)] * k8));  // H's limbs are byte pairs
        }
        // ---- 5. FINAL extras: canonical xR, yR, the sign bit
        canonical(finn, [&](int k) { return nxt[C(0, k)]; }, [&](int k) { return nxt[C(2, k)]; }, [&](int k) { return nxt[RL(2, k)]; }, K_QM1);
        canonical(finn, [&](int k) { return nxt[C(1, k)]; }, [&](int k) { return nxt[C(3, k)]; }, [&](int k) { return nxt[RL(3, k)]; }, K_QM1);
        c.constraint(finn * (nxt[C(0, 0)] - nxt[RH(2, 0)] * two - nxt[BS]));
        // ---- 6. slot registers, the addend selection, the scalar bits
#pragma unroll 1
        for (int col = XA0; col < SEL0; ++col) c.constraint(keep * (nxt[col] - loc[col]));
        c.constraint(keep * (nxt[SG] - loc[SG]));
        c.constraint(keep * (nxt[AIDX] - loc[AIDX]));
        {
            const F bs = loc[BS], bh = loc[BH], w11 = bs * bh, w10 = bs - w11, w01 = bh - w11, w00 = one - bs - bh + w11;
#pragma unroll 1
            for (int t = 0; t < 3; ++t)
#pragma unroll 1
                for (int k = 0; k < 16; ++k) {
                    const F xa = loc[XA0 + k], ya = loc[YA0 + k], x3 = loc[X30 + k], y3 = loc[Y30 + k];
                    const F na = t == 0 ? ya + xa : (t == 1 ? ya - xa : loc[NT0 + k]);
                    const F ba = t == 0 ? y3 - x3 : (t == 1 ? y3 + x3 : loc[BT0 + k]);
                    const F idc = (t < 2 && k == 0) ? one : zero;
                    c.constraint(loc[SEL0 + 16 * t + k] - (w00 * idc + w10 * kv(K_BC0 + t, k) + w01 * na + w11 * ba));
                }
            c.constraint((one - loc[SG]) * bh);
            c.constraint((one - loc[SG]) * per[P_STEP] * bs);
            c.constraint(stn * (nxt[LAH] - (one - per[P_LST]) * (loc[LAH] * two) - nxt[BH]));
            F acc = zero;
#pragma unroll 1
            for (int k = 0; k < 16; ++k) acc = acc + per[P_LE0 + k] * (nxt[LAH] - nxt[HR0 + k]);
            c.constraint(acc);
        }
        // ---- 7. the count of signed slots
        c.transition(nxt[CNT] - loc[CNT] - s0n * nxt[SG]);
        c.first_row(loc[CNT] - loc[SG]);
        c.last_row(loc[CNT] - pub[0]);
        // ---- 8. lookups of the local row: 672 range checks, 6 bus lookups, the table, the running sum
        {
            const X2<F> beta{chal[0], chal[1]}, gamma{chal[2], chal[3]}, g2 = gamma * gamma, g3 = g2 * gamma, g4 = g2 * g2;
            const X2<F> br = beta + g4 * F::from(TAG_R16);
            X2<F> hsum{zero, zero};
#pragma unroll 1
            for (int e = 0; e < N_RANGE; ++e) {
                const X2<F> h{loc[AX(e, 0)], loc[AX(e, 1)]};
                const X2<F> du = br + loc[2 * e], dv = br + loc[2 * e + 1];
                c.constraint_x2(h * du * dv - du - dv);
                hsum = hsum + h;
            }
            const F r0 = per[P_R0], r1 = per[P_R1], r255 = per[P_R255], slot4 = per[P_SLOT] * F::from(4), on = loc[SG] * pub[1];
            const F k32 = F::from(1ULL << 32);
            auto enc_a = [&](int k) -> F { return k < 15 ? loc[C(7, k)] : loc[C(7, 15)] + loc[BS] * F::from(32768); };
            auto enc_r = [&](int k) -> F { return k < 15 ? loc[C(1, k)] : loc[C(1, 15)] + loc[BS] * F::from(32768); };
            // half q of the digest (q = 2 word + (0 lo | 1 hi)): the big-endian sum of its four byte cells
            auto dh = [&](int q) -> F {
                if (q >= 16) return zero;
                const int b0 = 8 * (q >> 1) + ((q & 1) ? 0 : 4);
                return loc[BYA(b0)] * F::from(1ULL << 24) + loc[BYA(b0 + 1)] * k16 + loc[BYA(b0 + 2)] * F::from(256) + loc[BYA(b0 + 3)];
            };
            auto p3 = [&](auto&& f, int i) -> F { return f(i) + f(i + 1) * k16 + f(i + 2) * k32; };
            auto p2 = [&](auto&& f, int i) -> F { return f(i) + f(i + 1) * k16; };
            const F slot8 = per[P_SLOT] * F::from(8);
#pragma unroll 1
            for (int b = 0; b < 4; ++b) {
                const F m = on * (zero - r0 - r1);
                const F tag = r0 * F::from(TAG_KEY) + r1 * F::from(TAG_EDH);
                const F t0 = r0 * (loc[AIDX] * F::from(4) + F::from((uint64_t)b)) + r1 * (slot8 + F::from((uint64_t)b));
                const F u1 = r0 * p2(enc_a, 4 * b) + r1 * dh(3 * b), u2 = r0 * p2(enc_a, 4 * b + 2) + r1 * dh(3 * b + 1), u3 = r1 * dh(3 * b + 2);
                const X2<F> d = beta + t0 + gamma * u1 + g2 * u2 + g3 * u3 + g4 * tag;
                const X2<F> h{loc[AX(HB0 + b, 0)], loc[AX(HB0 + b, 1)]};
                c.constraint_x2(h * d - m);
                hsum = hsum + h;
            }
#pragma unroll 1
            for (int b = 4; b < 6; ++b) {
                const F m = on * (r0 + r255 - r1);
                const F tag = (r0 + r255) * F::from(TAG_EDMSG) + r1 * F::from(TAG_EDH);
                const F t0 = r0 * (slot4 + F::from((uint64_t)(b - 2))) + r255 * (slot4 + F::from((uint64_t)(b - 4))) + r1 * (slot8 + F::from((uint64_t)b));
                const int o = 8 * (b - 4);
                const F u1 = r0 * p3(enc_a, o) + r255 * p3(enc_r, o) + r1 * dh(3 * b), u2 = r0 * p3(enc_a, o + 3) + r255 * p3(enc_r, o + 3) + r1 * dh(3 * b + 1);
                const F u3 = r0 * p2(enc_a, o + 6) + r255 * p2(enc_r, o + 6) + r1 * dh(3 * b + 2);
                const X2<F> d = beta + t0 + gamma * u1 + g2 * u2 + g3 * u3 + g4 * tag;
                const X2<F> h{loc[AX(HB0 + b, 0)], loc[AX(HB0 + b, 1)]};
                c.constraint_x2(h * d - m);
                hsum = hsum + h;
            }
            const X2<F> ht{loc[AX(HT, 0)], loc[AX(HT, 1)]};
            c.constraint_x2(ht * (br + per[P_T]) - loc[MULT]);
            const X2<F> z{loc[AX(ZZ, 0)], loc[AX(ZZ, 1)]}, zn{nxt[AX(ZZ, 0)], nxt[AX(ZZ, 1)]};
            c.constraint_x2(zn - z - hsum + ht + X2<F>{apub[0], apub[1]});
        }
    }
};
using EdAir17 = EdAirT<17, 10>;
using EdAir16 = EdAirT<16, 12>;
