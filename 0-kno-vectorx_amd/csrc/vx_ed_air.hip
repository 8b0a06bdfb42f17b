// Witness generation for EdAir (air_ed.cuh): the rows of the Ed25519 verification table and its auxiliary (logUp) columns.
// Reference path: /root/reference circuits/builder/justification.rs:229-243 (300 conditional EdDSA verifications; the
// reference generates this witness inside curta's EdDSA gadget, starkyx v1.0.0, not vendored).
//   k_ed_slots   one lane per slot: decode A and R (RFC 8032 5.1.3), B - A in affine form, H = SHA-512(R || A || M),
//                H = qq l + hr with its carries, the canonical-range witnesses -- everything that is not a gadget result
//   k_ed_rows    one wave per slot, four groups of 16 lanes (lane k of a group owns limb k of every field element): the 256 rows
//                of a slot in order, the four groups working on the (up to four) independent gadgets of a dependency layer side by
//                side; a gadget = 16 multiply-accumulates per lane from operands broadcast through LDS, then the carry
//                normalisation (c = F mod 2q, r_k) which every lane runs redundantly on the 16 coefficients.  Rows go to a
//                row-major int32 staging buffer (16 lanes write 16 neighbouring cells)
//   k_ed_expand  staging [n][838] int32 -> trace [838][n] field elements through an LDS tile (negative limbs -> p + v)
//   k_ed_hist    multiplicities of the 16-bit range table (global atomics on a few interleaved copies)
//   k_ed_aux     one lane per row: 1 / (beta_r + v) comes from a 2^16-entry table built once per proof, so the 336 range
//                helpers of a row are gathers and additions; the six bus helpers exist on rows 0, 1, 255 of signed slots
#include <string.h>

#include "air_ed.cuh"
#include "ed25519.cuh"
#include "vx_internal.h"

namespace {
using namespace edc;

struct EdSlot {
    int32_t xa[16], ya[16], x3[16], y3[16], xr[16], yr[16];
    int32_t wxa[16], wya[16], wxr[16], wyr[16];  // q - 1 - x
    int32_t hl[32], qq[17], hr[16], hw[16], crlo[32], crhi[32];
    uint8_t cxa[16], cya[16], cxr[16], cyr[16], chr[16];  // carries of x + w = top (15 used)
    uint8_t dig[64];                                       // the SHA-512 digest H
    uint32_t s[8], h[8];                                   // the scalars whose bits the STEP rows consume
    uint32_t sign_a, sign_r, sg, cnt, aidx, pad;
};

__device__ void limbs16(const ed::U256& x, int32_t* out) {
    for (int i = 0; i < 8; ++i) out[2 * i] = x.w[i] & 0xFFFF, out[2 * i + 1] = x.w[i] >> 16;
}
// w = top - x (>= 0 by construction) and the carries of x + w = top
__device__ void canon_witness(const int32_t* x, int t, int32_t* w, uint8_t* cy) {
    int32_t borrow = 0;
    for (int k = 0; k < 16; ++k) {
        int32_t v = (int32_t)KT[t][k] - x[k] - borrow;
        borrow = v < 0;
        w[k] = v + (borrow << 16);
        cy[k] = (uint8_t)borrow;  // x_k + w_k + cy_(k-1) = top_k + 2^16 cy_k
    }
}
// slot s < k verifies the signature of authority idx[s]; slots k.. are idle (all identical: only slot k is made)
__global__ __launch_bounds__(64) void k_ed_slots(const uint8_t* pubkeys, const uint8_t* sigs, const uint8_t* msg, uint32_t msg_len, const uint32_t* idx, size_t k_active,
                                                 EdSlot* out, uint32_t* bad) {
    using namespace ed;
    const size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (s > k_active) return;
    EdSlot& o = out[s];
    const bool on = s < k_active;
    const size_t au = on ? idx[s] : 0;
    U256 one, zero;
    for (int j = 0; j < 8; ++j) one.w[j] = j == 0, zero.w[j] = 0;
    const U256 bx = fe_from(ED_BX), by = fe_from(ED_BY);
    Pt A = {bx, by, one, fe_mul(bx, by)}, R = {zero, one, one, zero};
    uint8_t dig[64];
    for (int j = 0; j < 64; ++j) dig[j] = 0;
    for (int j = 0; j < 8; ++j) o.s[j] = 0;
    if (on) {
        const uint8_t* pk = pubkeys + 32 * au;
        const uint8_t* sg = sigs + 64 * au;
        if (!pt_decode(pk, &A) || !pt_decode(sg, &R)) {
            atomicAdd(bad, 1u);
            return;
        }
        uint8_t buf[64 + 128];
        for (int j = 0; j < 32; ++j) buf[j] = sg[j], buf[32 + j] = pk[j];
        for (uint32_t j = 0; j < msg_len; ++j) buf[64 + j] = msg[j];
        sha512(buf, 64 + msg_len, dig);
        for (int j = 0; j < 8; ++j) o.s[j] = (uint32_t)sg[32 + 4 * j] | ((uint32_t)sg[33 + 4 * j] << 8) | ((uint32_t)sg[34 + 4 * j] << 16) | ((uint32_t)sg[35 + 4 * j] << 24);
        o.s[7] &= 0x1FFFFFFF;  // the table consumes 253 bits; a canonical S has no more
    }
    fe_canon(A.X), fe_canon(A.Y), fe_canon(R.X), fe_canon(R.Y);
    // B - A in affine form
    const Pt B = {bx, by, one, fe_mul(bx, by)};
    const Pt nA = {fe_sub(zero, A.X), A.Y, one, fe_sub(zero, fe_mul(A.X, A.Y))};
    const Pt d = pt_add(B, nA);
    const uint32_t em2[8] = ED_EXP_PM2_INIT;
    const U256 zi = fe_pow(d.Z, em2);
    U256 x3 = fe_mul(d.X, zi), y3 = fe_mul(d.Y, zi);
    fe_canon(x3), fe_canon(y3);
    limbs16(A.X, o.xa), limbs16(A.Y, o.ya), limbs16(x3, o.x3), limbs16(y3, o.y3), limbs16(R.X, o.xr), limbs16(R.Y, o.yr);
    canon_witness(o.xa, K_QM1, o.wxa, o.cxa), canon_witness(o.ya, K_QM1, o.wya, o.cya);
    canon_witness(o.xr, K_QM1, o.wxr, o.cxr), canon_witness(o.yr, K_QM1, o.wyr, o.cyr);
    o.sign_a = A.X.w[0] & 1, o.sign_r = R.X.w[0] & 1, o.sg = on, o.cnt = (uint32_t)(on ? s + 1 : k_active), o.aidx = (uint32_t)au, o.pad = 0;
    // H = qq l + hr, bit-serial: r = 2r + bit; if r >= l: r -= l, quotient bit 1
    uint32_t r[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[16];
    for (int j = 0; j < 16; ++j) q[j] = 0;
    for (int bit = 511; bit >= 0; --bit) {
        uint32_t carry = (dig[bit >> 3] >> (bit & 7)) & 1;
        for (int i = 0; i < 8; ++i) {
            const uint32_t nc = r[i] >> 31;
            r[i] = (r[i] << 1) | carry;
            carry = nc;
        }
        if (u256_geq(r, SC_L)) {
            u256_sub(r, r, SC_L);
            q[bit >> 5] |= 1u << (bit & 31);
        }
    }
    for (int j = 0; j < 8; ++j) o.h[j] = r[j];
    for (int j = 0; j < 64; ++j) o.dig[j] = dig[j];
    for (int j = 0; j < 32; ++j) o.hl[j] = (int32_t)dig[2 * j] | ((int32_t)dig[2 * j + 1] << 8);
    for (int j = 0; j < 17; ++j) o.qq[j] = (q[j >> 1] >> (16 * (j & 1))) & 0xFFFF;
    for (int j = 0; j < 16; ++j) o.hr[j] = (r[j >> 1] >> (16 * (j & 1))) & 0xFFFF;
    canon_witness(o.hr, K_LM1, o.hw, o.chr);
    int64_t cr = 0;
    for (int k = 0; k < 32; ++k) {
        int64_t e = cr;
        for (int i = 0; i < 17; ++i) {
            const int j = k - i;
            if (j >= 0 && j < 16) e += (int64_t)o.qq[i] * KT[K_LL][j];
        }
        if (k < 16) e += o.hr[k];
        e -= o.hl[k];
        cr = e >> 16;  // exact: the identity holds over the integers
        const int64_t v = cr + (1LL << 31);
        o.crlo[k] = (int32_t)(v & 0xFFFF), o.crhi[k] = (int32_t)(v >> 16);
    }
}

// c = F mod 2q as 16 limbs and the carries r_k of the gadget identity, for the coefficients F[0..16) (LDS); returns limb k
__device__ __forceinline__ void ed_normalise(const int64_t* F, int k, int32_t& c_out, int64_t& r_out) {
    int64_t L[16], carry = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int64_t v = F[j] + carry;
        L[j] = v & 0xFFFF, carry = v >> 16;
    }
    int64_t tot = 0;
    while (carry != 0) {  // 2^256 = 2q + 38
        tot += carry;
        const int64_t add = carry * 38;
        carry = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t v = L[j] + (j == 0 ? add : 0) + carry;
            L[j] = v & 0xFFFF, carry = v >> 16;
        }
    }
    bool big = L[0] >= 0xFFDA;
#pragma unroll
    for (int j = 1; j < 16; ++j) big &= L[j] == 0xFFFF;
    if (big) {
        L[0] = L[0] + 38 - 65536;
#pragma unroll
        for (int j = 1; j < 16; ++j) L[j] = 0;
        tot += 1;
    }
    int64_t prev = tot * 38;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int64_t rj = (F[j] + prev - L[j]) >> 16;
        if (j == k) c_out = (int32_t)L[j], r_out = rj;
        prev = rj;
    }
}

// One wave per slot: four groups of 16 lanes (lane k of a group owns limb k).  The gadgets of a row depend on each other in
// four layers (squares -> doubling products -> addend products -> sums), so the four groups work on the gadgets of one layer
// side by side and exchange their results through LDS: four gadget latencies per row instead of fourteen.
__global__ __launch_bounds__(64) void k_ed_rows(const EdSlot* slots, int32_t* stage, size_t k_active, uint32_t* bad) {
    __shared__ int32_t As[4][16], Bs[4][16], Cs[4][16];
    __shared__ int64_t Fs[4][16];
    const int gi = threadIdx.x >> 4, k = threadIdx.x & 15;
    const size_t slot = blockIdx.x;                                // slots 0 .. k_active; the last one is the idle slot
    const EdSlot& in = slots[slot];
    int32_t* row = stage + slot * 256 * (size_t)COLS;
    // F_k of a * b within this lane's group (both operands: this lane's limb)
    auto fold = [&](int32_t a, int32_t b) -> int64_t {
        __syncthreads();
        As[gi][k] = a, Bs[gi][k] = b;
        __syncthreads();
        int64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = k - i;
            acc += (int64_t)As[gi][i] * (j >= 0 ? (int64_t)Bs[gi][j] : 38 * (int64_t)Bs[gi][j + 16]);
        }
        return acc;
    };
    // gadget g (< 0: this group idles) of the current row from its coefficient F_k: writes c (or `store` for a zero-check) and
    // the carry cells, publishes c to the other groups; returns c_k
    auto finish = [&](int g, int64_t Fk, bool has_c, int32_t store) -> int32_t {
        __syncthreads();
        Fs[gi][k] = has_c ? Fk : 2 * Fk;
        __syncthreads();
        int32_t c = 0;
        int64_t r = 0;
        ed_normalise(Fs[gi], k, c, r);
        if (g >= 0) {
            if (!has_c && c != 0) atomicAdd(bad, 1u);  // a zero-check that does not vanish: the signature equation fails
            const int64_t rr = r + (1LL << 31);
            row[C(g, k)] = has_c ? c : store;
            row[RL(g, k)] = (int32_t)(rr & 0xFFFF), row[RH(g, k)] = (int32_t)(rr >> 16);
        }
        Cs[gi][k] = c;
        __syncthreads();
        return c;
    };
    auto kk = [&](int t) -> int32_t { return (int32_t)KT[t][k]; };
    auto pick = [&](int32_t v0, int32_t v1, int32_t v2, int32_t v3) -> int32_t { return gi == 0 ? v0 : gi == 1 ? v1 : gi == 2 ? v2 : v3; };
    const int32_t xa = in.xa[k], ya = in.ya[k], x3 = in.x3[k], y3 = in.y3[k], one_k = k == 0;
    auto clear = [&]() {
        for (int col = threadIdx.x; col < COLS; col += 64) row[col] = 0;
        __syncthreads();
    };
    int32_t nt = 0, bt = 0;
    // the slot registers of a row (group 0 writes them)
    auto registers = [&](int32_t bs, int32_t bh, int32_t lah) {
        if (gi) return;
        row[XA0 + k] = xa, row[YA0 + k] = ya, row[NT0 + k] = nt, row[X30 + k] = x3, row[Y30 + k] = y3, row[BT0 + k] = bt, row[HR0 + k] = in.hr[k];
        const int32_t w11 = bs & bh, w10 = bs - w11, w01 = bh - w11, w00 = 1 - bs - bh + w11;
        row[SEL0 + k] = w00 * one_k + w10 * kk(K_BC0) + w01 * (ya + xa) + w11 * (y3 - x3);
        row[SEL0 + 16 + k] = w00 * one_k + w10 * kk(K_BC1) + w01 * (ya - xa) + w11 * (y3 + x3);
        row[SEL0 + 32 + k] = w10 * kk(K_BC2) + w01 * nt + w11 * bt;
        if (k < 7) row[BS + k] = k == 0 ? bs : k == 1 ? bh : k == 2 ? lah : k == 3 ? (int32_t)in.sg : k == 4 ? (int32_t)in.cnt : k == 5 ? 0 : (int32_t)in.aidx;  // BS BH LAH SG CNT MULT AIDX
    };
    // ---- row 0: SETUP-A.  Layers: (u, xx, yy, v) -> (nt2d, dxx, t, bt2d) -> the three zero-checks
    clear();
    {
        const int g1[4] = {0, 2, 3, 9};
        (void)finish(g1[gi], fold(pick(xa, xa, ya, x3), pick(ya, xa, ya, y3)), true, 0);
        const int32_t u = Cs[0][k], xx = Cs[1][k], yy = Cs[2][k], v = Cs[3][k];
        const int g2[4] = {1, 4, 6, 10};
        (void)finish(g2[gi], fold(pick(u, xx, u, v), pick(kk(K_2DN), kk(K_D), kk(K_BD), kk(K_2D))), true, 0);
        nt = Cs[0][k], bt = Cs[3][k];
        const int32_t dxx = Cs[1][k], t = Cs[2][k];
        // group 0: yy - xx - 1 - dxx yy; group 1: x3 (1 + t) - yA xB + xA yB; group 2: y3 (1 - t) - yA yB + xA xB
        const int64_t f1 = fold(pick(dxx, x3, y3, 0), pick(yy, one_k + t, one_k - t, 0));
        const int64_t f2 = fold(pick(0, ya, ya, 0), pick(0, kk(K_XB), kk(K_YB), 0));
        const int64_t f3 = fold(pick(0, xa, xa, 0), pick(0, kk(K_YB), kk(K_XB), 0));
        const int g3[4] = {5, 7, 8, -1};
        (void)finish(g3[gi], gi == 0 ? (int64_t)yy - xx - one_k - f1 : f1 - f2 + f3, false, pick(xa, ya, x3, 0));
        if (gi == 0) {
            row[C(11, k)] = y3;
            row[C(12, k)] = in.wxa[k], row[C(13, k)] = in.wya[k];
            if (k < 15) row[RL(12, k)] = in.cxa[k], row[RL(13, k)] = in.cya[k];
            if (k == 0) row[RH(12, 0)] = xa >> 1;
        }
        registers((int32_t)in.sign_a, 0, 0);
    }
    // ---- row 1: SETUP-B
    row += COLS;
    clear();
    if (gi == 0) {
        row[C(0, k)] = in.hl[k], row[RL(0, k)] = in.hl[16 + k], row[RH(0, k)] = in.qq[k];
        row[RL(1, k)] = in.hr[k], row[RH(1, k)] = in.hw[k];
        row[C(2, k)] = in.crlo[k], row[RL(2, k)] = in.crlo[16 + k], row[C(3, k)] = in.crhi[k], row[RL(3, k)] = in.crhi[16 + k];
        if (k < 15) row[C(4, k)] = in.chr[k];
        if (k == 0) row[C(1, 0)] = in.qq[16], row[C(12, 0)] = 1, row[C(13, 0)] = 1;
        for (int j = k; j < 64; j += 16) row[BYA(j)] = in.dig[j], row[BYB(j)] = 256 * (int32_t)in.dig[j];  // (cell columns are congruent to j mod 16)
    }
    registers(0, 0, 0);
    // ---- rows 2..254: STEP
    int32_t X = 0, Y = one_k, Z = one_k, lah = 0;
    for (int r = 2; r < 255; ++r) {
        row += COLS;
        const int bit = 252 - (r - 2);
        const int32_t bs = (in.s[bit >> 5] >> (bit & 31)) & 1, bh = (in.h[bit >> 5] >> (bit & 31)) & 1;
        lah = ((bit == 252 || (bit & 15) == 15) ? 0 : 2 * lah) + bh;
        registers(bs, bh, lah);
        const int32_t w11 = bs & bh, w10 = bs - w11, w01 = bh - w11, w00 = 1 - bs - bh + w11;
        const int32_t s0 = w00 * one_k + w10 * kk(K_BC0) + w01 * (ya + xa) + w11 * (y3 - x3), s1 = w00 * one_k + w10 * kk(K_BC1) + w01 * (ya - xa) + w11 * (y3 + x3);
        const int32_t s2 = w10 * kk(K_BC2) + w01 * nt + w11 * bt;
        const int32_t q = pick(X, Y, Z, X + Y);
        (void)finish(gi, fold(q, q), true, 0);
        const int32_t c0 = Cs[0][k], c1 = Cs[1][k], c2 = Cs[2][k], c3 = Cs[3][k];
        const int32_t E = c3 - c0 - c1, G = c1 - c0, F = G - 2 * c2, H = -c0 - c1;
        (void)finish(4 + gi, fold(pick(E, G, E, F), pick(F, H, H, G)), true, 0);
        const int32_t c4 = Cs[0][k], c5 = Cs[1][k], c6 = Cs[2][k], c7 = Cs[3][k];
        (void)finish(gi < 3 ? 8 + gi : -1, fold(pick(c5 - c4, c5 + c4, c6, 0), pick(s0, s1, s2, 0)), true, 0);
        const int32_t c8 = Cs[0][k], c9 = Cs[1][k], c10 = Cs[2][k];
        const int32_t E2 = c9 - c8, F2 = 2 * c7 - c10, G2 = 2 * c7 + c10, H2 = c9 + c8;
        (void)finish(gi < 3 ? 11 + gi : -1, fold(pick(E2, G2, F2, 0), pick(F2, H2, G2, 0)), true, 0);
        X = Cs[0][k], Y = Cs[1][k], Z = Cs[2][k];
    }
    // ---- row 255: FINAL
    row += COLS;
    clear();
    {
        const int32_t xr = in.xr[k], yr = in.yr[k];
        const int64_t f = fold(pick(xr, yr, 0, 0), pick(Z, Z, 0, 0));
        (void)finish(gi < 2 ? gi : -1, f - pick(X, Y, 0, 0), false, pick(xr, yr, 0, 0));
        if (gi == 0) {
            row[C(2, k)] = in.wxr[k], row[C(3, k)] = in.wyr[k];
            if (k < 15) row[RL(2, k)] = in.cxr[k], row[RL(3, k)] = in.cyr[k];
            if (k == 0) row[RH(2, 0)] = xr >> 1;
        }
        registers((int32_t)in.sign_r, 0, lah);
    }
}

// staging [(k_active + 1) * 256][COLS] int32 -> trace [COLS][n] field elements; one 64 x 64 tile per block; every idle slot
// is a copy of staging slot k_active
__global__ __launch_bounds__(256) void k_ed_expand(const int32_t* stage, uint64_t* trace, size_t n, size_t k_active) {
    __shared__ int32_t tile[64][65];
    const size_t row0 = blockIdx.x * (size_t)64, slot = row0 >> 8, src0 = (slot < k_active ? slot : k_active) * 256 + (row0 & 255);
    const int col0 = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) tile[r][tx] = col0 + tx < COLS ? stage[(src0 + r) * COLS + col0 + tx] : 0;
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {
        if (col0 + c >= COLS) break;
        const int32_t v = tile[tx][c];
        trace[(size_t)(col0 + c) * n + row0 + tx] = v < 0 ? GL_P - (uint64_t)(-(int64_t)v) : (uint64_t)v;
    }
}
// one block per staged row; the rows of the idle slot count once for every idle slot of the trace
constexpr int ED_HIST_COPIES = 32;
__global__ __launch_bounds__(256) void k_ed_hist(const int32_t* stage, size_t k_active, uint32_t idle_weight, uint32_t* hist) {
    const size_t row = blockIdx.x;
    const uint32_t w = (row >> 8) < k_active ? 1u : idle_weight;
    uint32_t* h = hist + (size_t)(blockIdx.x % ED_HIST_COPIES) * 65536;
    for (int col = threadIdx.x; col < CELLS; col += 256) atomicAdd(&h[(uint32_t)stage[row * COLS + col] & 65535u], w);
}
__global__ __launch_bounds__(256) void k_ed_mult(const uint32_t* hist, uint64_t* mult_col) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    uint64_t s = 0;
    for (int c = 0; c < ED_HIST_COPIES; ++c) s += hist[(size_t)c * 65536 + i];
    mult_col[i] = s;
}

__global__ __launch_bounds__(256) void k_ed_inv_table(gl2 br, gl2* inv) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    inv[v] = gl2_inv(gl2{gl_add(br.a, v), br.b});
}
struct EdAuxArgs {
    const uint64_t* tr;
    uint64_t* aux;
    const gl2* inv;
    size_t n;
    gl2 beta, gamma;
    uint64_t bus_on;
};
__global__ __launch_bounds__(256) void k_ed_aux(EdAuxArgs a) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, n = a.n;
    if (i >= n) return;
    auto T = [&](int col) -> uint64_t { return a.tr[(size_t)col * n + i]; };
    gl2 hsum{0, 0};
#pragma unroll 4
    for (int e = 0; e < N_RANGE; ++e) {
        const gl2 h = gl2_add(a.inv[T(2 * e) & 65535], a.inv[T(2 * e + 1) & 65535]);  // (masked: a foreign trace must not index out of the table)
        a.aux[(size_t)(2 * e) * n + i] = h.a, a.aux[(size_t)(2 * e + 1) * n + i] = h.b;
        hsum = gl2_add(hsum, h);
    }
    const int r = (int)(i & 255);
    gl2 hb[N_BUS];
    for (int b = 0; b < N_BUS; ++b) hb[b] = gl2{0, 0};
    if ((r == 0 || r == 1 || r == 255) && a.bus_on && T(SG)) {
        const gl2 g2 = gl2_mul(a.gamma, a.gamma), g3 = gl2_mul(g2, a.gamma), g4 = gl2_mul(g2, g2);
        const uint64_t slot4 = 4 * (uint64_t)(i >> 8), sign = T(BS), aidx4 = 4 * T(AIDX);
        auto enc = [&](int g, int k) -> uint64_t { return k < 15 ? T(C(g, k)) : T(C(g, 15)) + 32768 * sign; };
        auto fp = [&](uint64_t t0, uint64_t t1, uint64_t t2, uint64_t t3, int tag) -> gl2 {
            gl2 d = gl2_add(a.beta, gl2_add(gl2_scale(a.gamma, t1), gl2_add(gl2_scale(g2, t2), gl2_add(gl2_scale(g3, t3), gl2_scale(g4, (uint64_t)tag)))));
            d.a = gl_add(d.a, t0);
            return d;
        };
        auto neg = [](gl2 x) -> gl2 { return gl2{gl_neg(x.a), gl_neg(x.b)}; };
        if (r == 0) {
            for (int b = 0; b < 4; ++b)
                hb[b] = neg(gl2_inv(fp(aidx4 + b, enc(7, 4 * b) | (enc(7, 4 * b + 1) << 16), enc(7, 4 * b + 2) | (enc(7, 4 * b + 3) << 16), 0, TAG_KEY)));
            for (int b = 0; b < 2; ++b)
                hb[4 + b] = gl2_inv(fp(slot4 + b + 2, enc(7, 8 * b) | (enc(7, 8 * b + 1) << 16) | (enc(7, 8 * b + 2) << 32),
                                       enc(7, 8 * b + 3) | (enc(7, 8 * b + 4) << 16) | (enc(7, 8 * b + 5) << 32), enc(7, 8 * b + 6) | (enc(7, 8 * b + 7) << 16), TAG_EDMSG));
        } else if (r == 1) {
            auto dh = [&](int q) -> uint64_t {  // half q of the digest: big-endian sum of four byte cells
                if (q >= 16) return 0;
                const int b0 = 8 * (q >> 1) + ((q & 1) ? 0 : 4);
                return (T(BYA(b0)) << 24) | (T(BYA(b0 + 1)) << 16) | (T(BYA(b0 + 2)) << 8) | T(BYA(b0 + 3));
            };
            for (int b = 0; b < 6; ++b) hb[b] = neg(gl2_inv(fp(2 * slot4 + b, dh(3 * b), dh(3 * b + 1), dh(3 * b + 2), TAG_EDH)));
        } else {
            for (int b = 0; b < 2; ++b)
                hb[4 + b] = gl2_inv(fp(slot4 + b, enc(1, 8 * b) | (enc(1, 8 * b + 1) << 16) | (enc(1, 8 * b + 2) << 32),
                                       enc(1, 8 * b + 3) | (enc(1, 8 * b + 4) << 16) | (enc(1, 8 * b + 5) << 32), enc(1, 8 * b + 6) | (enc(1, 8 * b + 7) << 16), TAG_EDMSG));
        }
    }
    for (int b = 0; b < N_BUS; ++b) {
        a.aux[(size_t)(2 * (HB0 + b)) * n + i] = hb[b].a, a.aux[(size_t)(2 * (HB0 + b) + 1) * n + i] = hb[b].b;
        hsum = gl2_add(hsum, hb[b]);
    }
    const gl2 ht = gl2_scale(a.inv[i & 65535], T(MULT));
    a.aux[(size_t)(2 * HT) * n + i] = ht.a, a.aux[(size_t)(2 * HT + 1) * n + i] = ht.b;
    const gl2 inc = gl2_sub(hsum, ht);
    a.aux[(size_t)(2 * ZZ) * n + i] = inc.a, a.aux[(size_t)(2 * ZZ + 1) * n + i] = inc.b;  // increments; the scan makes them the running sum
}
}  // namespace

int32_t vx_ed_air_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub) {
    const size_t n = (size_t)1 << log_n;
    gl2* inv = (gl2*)vx_pool_alloc(ctx, 65536 * sizeof(gl2));
    if (!inv) return vx_fail(ctx, VX_ERR_OOM, "ed aux: out of device memory");
    // beta_r = beta + gamma^4 * TAG_R16 (host arithmetic in the quadratic extension X^2 = 7)
    auto xmul = [](const uint64_t* x, const uint64_t* y, uint64_t* o) {
        const uint64_t a = glh::add(glh::mul(x[0], y[0]), glh::mul(7, glh::mul(x[1], y[1]))), b = glh::add(glh::mul(x[0], y[1]), glh::mul(x[1], y[0]));
        o[0] = a, o[1] = b;
    };
    uint64_t g2[2], g4[2];
    xmul(chal + 2, chal + 2, g2), xmul(g2, g2, g4);
    const gl2 br{glh::add(chal[0], glh::mul(g4[0], TAG_R16)), glh::add(chal[1], glh::mul(g4[1], TAG_R16))};
    hipLaunchKernelGGL(k_ed_inv_table, dim3(256), dim3(256), 0, ctx->stream, br, inv);
    EdAuxArgs a{trace, aux, inv, n, gl2{chal[0], chal[1]}, gl2{chal[2], chal[3]}, pub[1]};
    hipLaunchKernelGGL(k_ed_aux, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
    const hipError_t e = hipGetLastError();
    vx_pool_free(ctx, inv);  // recycled only by later work on the same stream
    if (e != hipSuccess) return vx_fail(ctx, VX_ERR_DEVICE, "ed aux: %s", hipGetErrorString(e));
    return vx_bus_close_dev(ctx, aux + (size_t)(2 * ZZ) * n, log_n, aux_pub);
}

int32_t vx_ed_trace_dev(vx_ctx* ctx, const uint8_t* pubkeys, const uint8_t* sigs, const uint8_t* msg, uint32_t msg_len, const uint8_t* signed_flags, size_t n_sigs,
                        int log_n, uint64_t bus_on, uint64_t* trace_d, uint64_t pub_out[2]) {
    const size_t n = (size_t)1 << log_n, m = n >> 8;
    VX_CHECK(log_n >= 16 && log_n <= 20, "ed trace: log_n %d out of range [16, 20] (the trace holds one copy of the 2^16-row range table)", log_n);
    VX_CHECK(msg_len <= 64, "ed trace: message of %u bytes (the precommit has 53)", msg_len);
    std::vector<uint32_t> idx;  // compact slots: slot s verifies the s-th flagged authority
    for (size_t s = 0; s < n_sigs; ++s)
        if (signed_flags[s]) idx.push_back((uint32_t)s);
    const size_t k = idx.size(), k4 = k + 1;  // staged slots: the active ones + the idle one
    VX_CHECK(k < m, "ed trace: %zu signatures do not fit the %zu slots of 2^%d rows (one slot stays idle)", k, m - 1, log_n);
    // device scratch: keys | sigs | msg | idx | bad | slots | hist, then the staging buffer from the pool
    const size_t w_keys = 4 * n_sigs + 1, w_sigs = 8 * n_sigs + 1, w_msg = 8, w_idx = (k + 1) / 2 + 1, w_slots = ((k + 1) * sizeof(EdSlot) + 7) / 8;
    const size_t w_hist = (size_t)ED_HIST_COPIES * 65536 / 2;
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, w_keys + w_sigs + w_msg + w_idx + 1 + w_slots + w_hist, &sc));
    uint8_t* d_keys = (uint8_t*)sc;
    uint8_t* d_sigs = (uint8_t*)(sc + w_keys);
    uint8_t* d_msg = (uint8_t*)(sc + w_keys + w_sigs);
    uint32_t* d_idx = (uint32_t*)(sc + w_keys + w_sigs + w_msg);
    uint32_t* d_bad = (uint32_t*)(sc + w_keys + w_sigs + w_msg + w_idx);
    EdSlot* d_slots = (EdSlot*)(sc + w_keys + w_sigs + w_msg + w_idx + 1);
    uint32_t* d_hist = (uint32_t*)(sc + w_keys + w_sigs + w_msg + w_idx + 1 + w_slots);
    if (n_sigs) {
        VX_HIP(hipMemcpyAsync(d_keys, pubkeys, 32 * n_sigs, hipMemcpyHostToDevice, ctx->stream));
        VX_HIP(hipMemcpyAsync(d_sigs, sigs, 64 * n_sigs, hipMemcpyHostToDevice, ctx->stream));
    }
    if (k) VX_HIP(hipMemcpyAsync(d_idx, idx.data(), k * 4, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemcpyAsync(d_msg, msg, msg_len, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemsetAsync(d_bad, 0, 8, ctx->stream));
    VX_HIP(hipMemsetAsync(d_hist, 0, w_hist * 8, ctx->stream));
    int32_t* stage = (int32_t*)vx_pool_alloc(ctx, k4 * 256 * (size_t)COLS * 4);
    if (!stage) return vx_fail(ctx, VX_ERR_OOM, "ed trace: out of device memory");
    hipLaunchKernelGGL(k_ed_slots, dim3((unsigned)((k + 1 + 63) / 64)), dim3(64), 0, ctx->stream, d_keys, d_sigs, d_msg, msg_len, (const uint32_t*)d_idx, k, d_slots, d_bad);
    hipLaunchKernelGGL(k_ed_rows, dim3((unsigned)k4), dim3(64), 0, ctx->stream, (const EdSlot*)d_slots, stage, k, d_bad);
    hipLaunchKernelGGL(k_ed_expand, dim3((unsigned)(n / 64), (COLS + 63) / 64), dim3(256), 0, ctx->stream, (const int32_t*)stage, trace_d, n, k);
    hipLaunchKernelGGL(k_ed_hist, dim3((unsigned)((k + 1) * 256)), dim3(256), 0, ctx->stream, (const int32_t*)stage, k, (uint32_t)(m - k), d_hist);
    hipLaunchKernelGGL(k_ed_mult, dim3(256), dim3(256), 0, ctx->stream, (const uint32_t*)d_hist, trace_d + (size_t)MULT * n);
    const hipError_t e = hipGetLastError();
    vx_pool_free(ctx, stage);
    if (e != hipSuccess) return vx_fail(ctx, VX_ERR_DEVICE, "ed trace: %s", hipGetErrorString(e));
    uint32_t bad = 0;
    VX_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));  // idx (host vector) must outlive the copy; bad is read below
    if (bad) return vx_fail(ctx, VX_ERR_STATEMENT, "ed trace: a signed slot does not verify (undecodable key / R, or [S]B != R + [h]A)");
    pub_out[0] = k, pub_out[1] = bus_on;
    return VX_OK;
}

extern "C" int32_t vx_ed_trace(vx_ctx* ctx, const uint8_t* pubkeys, const uint8_t* signatures, const uint8_t* message, uint32_t message_len, const uint8_t* signed_flags,
                               size_t n_signatures, int log_n, uint32_t bus_on, vx_buf* trace_out, uint64_t public_inputs_out[2]) {
    if (!ctx || !message || !trace_out || !public_inputs_out || (n_signatures && (!pubkeys || !signatures || !signed_flags))) return VX_ERR_ARG;
    VX_CHECK(log_n >= 16 && log_n <= 20 && trace_out->n >= ((size_t)COLS << log_n), "ed trace: trace buffer holds %zu elements, 2^%d rows need %zu", trace_out->n, log_n,
             (size_t)COLS << log_n);
    return vx_ed_trace_dev(ctx, pubkeys, signatures, message, message_len, signed_flags, n_signatures, log_n, bus_on ? 1 : 0, trace_out->d, public_inputs_out);
}
