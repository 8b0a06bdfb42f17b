// K8: witness/trace generation for BlakeChainAir on the GPU, and the header_range prove entry.
//   kernel A  k_blake_chain: one lane per header -- digest + the chaining value before every
//             128-byte chunk (the only sequential part of BLAKE2b);
//   kernel B  k_blake_trace: one lane per TRACE ROW (block b, r = row mod 16): recomputes
//             the <= 12 rounds it needs from the chunk's chaining value and writes its 729
//             cells; lanes of a wave write 64 consecutive rows of a column, so every store
//             instruction is a coalesced 512-byte segment of the column-major trace;
//   kernel C  k_blake_aux: the logUp helper columns once the lookup challenges are known.
// Replaces the curta Blake2b witness generation behind hash_encoded_header
// (/root/reference circuits/builder/header.rs:14-19) for the synthetic header chain.
#include <string.h>

#include "air_blake.cuh"
#include <thread>

#include "vx_internal.h"


struct BlockDesc {
    uint64_t msg_off;  // byte offset of the 128-byte chunk in the headers buffer; ~0 = padding block
    uint32_t t, inc;
    uint32_t D[8];     // digest register before this block
    uint8_t fin, first, act, pad;
    uint32_t num;      // block number of the header this chunk belongs to
};

__device__ __forceinline__ uint64_t b_rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

struct GRec {
    uint64_t w[8];  // a1 d1 c1 b1 a2 d2 c2 b2
    uint64_t in_b, in_d;  // the operands that enter an XOR lookup
    uint8_t car[8];
};
__device__ __forceinline__ void carries(uint64_t o1, uint64_t o2, uint64_t o3, uint8_t* out) {
    uint64_t lo = (o1 & 0xFFFFFFFFULL) + (o2 & 0xFFFFFFFFULL) + (o3 & 0xFFFFFFFFULL);
    uint64_t klo = lo >> 32;
    uint64_t hi = (o1 >> 32) + (o2 >> 32) + (o3 >> 32) + klo;
    out[0] = (uint8_t)klo;
    out[1] = (uint8_t)(hi >> 32);
}
__device__ __forceinline__ void g_mix(uint64_t* v, int ia, int ib, int ic, int id, uint64_t x, uint64_t y, GRec* rec) {
    uint64_t a = v[ia], b = v[ib], c = v[ic], d = v[id];
    uint64_t a1 = a + b + x, d1 = b_rotr(d ^ a1, 32), c1 = c + d1, b1 = b_rotr(b ^ c1, 24);
    uint64_t a2 = a1 + b1 + y, d2 = b_rotr(d1 ^ a2, 16), c2 = c1 + d2, b2 = b_rotr(b1 ^ c2, 63);
    if (rec) {
        rec->w[0] = a1, rec->w[1] = d1, rec->w[2] = c1, rec->w[3] = b1, rec->w[4] = a2, rec->w[5] = d2, rec->w[6] = c2, rec->w[7] = b2;
        rec->in_b = b, rec->in_d = d;
        carries(a, b, x, rec->car);
        carries(c, d1, 0, rec->car + 2);
        carries(a1, b1, y, rec->car + 4);
        carries(c1, d2, 0, rec->car + 6);
    }
    v[ia] = a2, v[ib] = b2, v[ic] = c2, v[id] = d2;
}
__device__ void blake_round(uint64_t* v, const uint64_t* m, int round, GRec* rec) {
    const uint8_t* s = blk::ORDER[round + 1];  // ORDER[r] for r = 1..12 is sigma[r-1]
    for (int k = 0; k < 4; ++k) g_mix(v, k, 4 + k, 8 + k, 12 + k, m[s[2 * k]], m[s[2 * k + 1]], rec ? rec + k : nullptr);
    for (int j = 0; j < 4; ++j)
        g_mix(v, j, 4 + (j + 1) % 4, 8 + (j + 2) % 4, 12 + (j + 3) % 4, m[s[8 + 2 * j]], m[s[8 + 2 * j + 1]], rec ? rec + 4 + j : nullptr);
}
__device__ void blake_init_v(uint64_t* v, const uint64_t* h, uint64_t t, bool fin) {
    for (int i = 0; i < 8; ++i) v[i] = h[i], v[8 + i] = blk::IV[i];
    v[12] ^= t;
    if (fin) v[14] = ~v[14];
}

// one lane per header: digest and the chaining value in front of each chunk
__global__ __launch_bounds__(64) void k_blake_chain(const uint8_t* msgs, size_t stride, const uint32_t* sizes, size_t n,
                                                    const uint32_t* block_base, uint64_t* hchain, uint8_t* digests) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t* p = (const uint64_t*)(msgs + i * stride);
    const uint32_t len = sizes[i];
    const uint32_t nchunks = len == 0 ? 1 : (len + 127) / 128;
    uint64_t h[8], m[16], v[16];
    for (int k = 0; k < 8; ++k) h[k] = blk::IV[k];
    h[0] ^= 0x01010020ULL;
    for (uint32_t cidx = 0; cidx < nchunks; ++cidx) {
        uint64_t* hc = hchain + 8 * ((size_t)block_base[i] + cidx);
        for (int k = 0; k < 8; ++k) hc[k] = h[k];
        const bool fin = cidx + 1 == nchunks;
        const uint32_t off = 128 * cidx, rem = fin ? len - off : 128;
        for (int k = 0; k < 16; ++k) {
            uint32_t b = 8 * k;
            uint64_t w = 0;
            if (b < rem) {
                w = p[(off >> 3) + k];
                if (rem - b < 8) w &= (1ULL << (8 * (rem - b))) - 1;
            }
            m[k] = w;
        }
        blake_init_v(v, h, fin ? len : off + 128, fin);
        for (int r = 0; r < 12; ++r) blake_round(v, m, r, nullptr);
        for (int k = 0; k < 8; ++k) h[k] ^= v[k] ^ v[k + 8];
    }
    uint64_t* d = (uint64_t*)(digests + 32 * i);
    d[0] = h[0], d[1] = h[1], d[2] = h[2], d[3] = h[3];
}

// ---- trace rows ---------------------------------------------------------------------------------------------------
// One lane per trace row (block b, r = row mod 16): recomputes the <= 12 rounds it needs from the chunk's chaining value
// and stores its 729 cells byte by byte (lanes of a wave write 64 consecutive rows of a column: every store instruction
// is one coalesced 512-byte segment of the column-major trace).  The same lane knows every XOR its row looks up, so it
// also bumps the multiplicity histograms of the two tables (hist[0 .. 2^16) for T1, hist[2^16 .. 2^17) for T2).
__global__ __launch_bounds__(256) void k_blake_trace(const uint8_t* msgs, const BlockDesc* descs, const uint64_t* hchain, size_t n_real,
                                                     uint64_t* __restrict__ tr, uint32_t* __restrict__ hist, size_t n) {
    using namespace blk;
    const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (row >= n) return;
    const size_t b = row >> 4;
    const int r = (int)(row & 15);
    const BlockDesc d = descs[b];
    uint64_t h[8], m[16], v[16];
    if (b < n_real) {
        for (int k = 0; k < 8; ++k) h[k] = hchain[8 * b + k];
        const uint64_t* p = (const uint64_t*)(msgs + d.msg_off);
        for (int k = 0; k < 16; ++k) {
            uint32_t bb = 8 * k;
            uint64_t w = 0;
            if (bb < d.inc) {
                w = p[k];
                if (d.inc - bb < 8) w &= (1ULL << (8 * (d.inc - bb))) - 1;
            }
            m[k] = w;
        }
    } else {  // padding block: the 36-byte message D || compact(number)
        for (int k = 0; k < 8; ++k) h[k] = IV[k];
        h[0] ^= 0x01010020ULL;
        for (int k = 0; k < 16; ++k) m[k] = k < 4 ? ((uint64_t)d.D[2 * k] | ((uint64_t)d.D[2 * k + 1] << 32)) : 0;
        m[4] = 4ULL * d.num + 2;  // bytes 32..36: SCALE compact (4-byte mode) of the last block number
    }
    auto cell = [&](int col) -> uint64_t& { return tr[(size_t)col * n + row]; };
    auto put = [&](int k, int slot, uint64_t word) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cell(GC(k, slot, j)) = (word >> (8 * j)) & 0xFF;
    };
    auto put_lt = [&](int k, uint64_t x) {  // (low 7 bits, top bit) of every byte of x
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t bv = (uint32_t)(x >> (8 * j)) & 0xFF;
            cell(GC(k, S_L, j)) = bv & 127;
            cell(GC(k, S_T, j)) = bv >> 7;
        }
    };
    auto look1 = [&](uint64_t a, uint64_t bq) {  // 8 byte lookups (a_i, b_i, .) into T1
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&hist[((a >> (8 * j)) & 0xFF) | (((bq >> (8 * j)) & 0xFF) << 8)], 1u);
    };
    auto look2 = [&](uint64_t a, uint64_t bq) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&hist[65536 + (((a >> (8 * j)) & 0xFF) | (((bq >> (8 * j)) & 0xFF) << 8))], 1u);
    };
    // ---- G area + carries: zero unless this row uses the cell
    for (int col = 0; col < MS0; ++col) cell(col) = 0;  // (a lane's later store to the same address wins)
    blake_init_v(v, h, d.t, d.fin);
    if (r == 0) {
        for (int w = 0; w < 16; ++w) {
            const int mm = w & 3;
            if (w < 4) put(4 + w, S_A2, v[w]);
            else if (w < 8) put_lt(4 + (mm + 3) % 4, b_rotr(v[w], 1));  // 2 L[j] + T[j-1] = byte j of v[w]
            else if (w < 12) put(4 + (mm + 2) % 4, S_C2, v[w]);
            else put(4 + (mm + 1) % 4, S_D2, v[w]);
        }
    } else {
        GRec rec[8];
        uint64_t vin[16];
        const int last = r <= 12 ? r - 1 : 11;  // rounds 0 .. last
        for (int q = 0; q <= last; ++q) {
            if (q == last)
                for (int k = 0; k < 16; ++k) vin[k] = v[k];
            blake_round(v, m, q, q == last && r <= 12 ? rec : nullptr);
        }
        if (r <= 12) {
            for (int k = 0; k < 8; ++k) {
                const uint64_t *w = rec[k].w;  // a1 d1 c1 b1 a2 d2 c2 b2
                put(k, S_A1, w[0]), put(k, S_D1, w[1]), put(k, S_C1, w[2]), put(k, S_B1, w[3]), put(k, S_A2, w[4]), put(k, S_D2, w[5]), put(k, S_C2, w[6]);
                put_lt(k, w[3] ^ w[6]);
                cell(CAR(k, 0)) = rec[k].car[0], cell(CAR(k, 1)) = rec[k].car[1], cell(CAR(k, 2)) = rec[k].car[4], cell(CAR(k, 3)) = rec[k].car[5];
                // the row's lookups: (d, A1), (b, C1), (D1, A2) into T1 and (B1, C2) into T2
                look1(rec[k].in_d, w[0]);
                look1(rec[k].in_b, w[2]);
                look1(w[1], w[4]);
                look2(w[3], w[6]);
            }
        } else if (r == 13 || r == 14) {
            for (int w = 0; w < 8; ++w) {
                const uint64_t u = v[w] ^ v[8 + w];
                const uint64_t x = r == 13 ? v[w] : u, y = r == 13 ? v[8 + w] : h[w];
                put(w, S_D1, x), put(w, S_A2, y), put(w, S_D2, b_rotr(x ^ y, 16));
                look1(x, y);
            }
        }
        (void)vin;
    }
    uint64_t h_out[8];
    if (r >= 13)
        for (int k = 0; k < 8; ++k) h_out[k] = h[k] ^ v[k] ^ v[k + 8];
    // ---- message schedule + bytes of natural word r (range checked as (byte, 0, byte) in T1)
    for (int s = 0; s < 16; ++s) {
        const uint64_t w = m[ORDER[r][s]];
        cell(MS(s, 0)) = w & 0xFFFFFFFFULL;
        cell(MS(s, 1)) = w >> 32;
    }
    for (int j = 0; j < 8; ++j) cell(MB0 + j) = (m[r] >> (8 * j)) & 0xFF;
    look1(m[r], 0);
    for (int bq = 0; bq < 8; ++bq) cell(MK0 + bq) = (uint32_t)(8 * r + bq) < d.inc ? 1 : 0;
    cell(CNT) = d.inc < (uint32_t)(8 * (r + 1)) ? d.inc : (uint32_t)(8 * (r + 1));
    // ---- H register
    for (int w = 0; w < 8; ++w) {
        const uint64_t hv = r <= 13 ? h[w] : (r == 14 ? h_out[w] : (d.fin ? (w == 0 ? IV[0] ^ 0x01010020ULL : IV[w]) : h_out[w]));
        cell(HL(w, 0)) = hv & 0xFFFFFFFFULL;
        cell(HL(w, 1)) = hv >> 32;
    }
    // ---- digest register, flags, counters
    const bool cap = d.act && d.fin;
    for (int j = 0; j < 8; ++j) {
        uint32_t dv = d.D[j];
        if (r == 15 && cap) dv = (uint32_t)(h_out[j / 2] >> (32 * (j & 1)));
        cell(D0 + j) = dv;
    }
    cell(ACT) = d.act, cell(FIN) = d.fin, cell(FIRST) = d.first, cell(CAP) = cap ? 1 : 0;
    cell(T) = d.t, cell(INC) = d.inc, cell(NUM) = d.num, cell(FA) = (d.first && d.act) ? 1 : 0;
    for (int i = 0; i < 32; ++i) cell(TB0 + i) = (d.t >> i) & 1;
    for (int i = 0; i < 8; ++i) cell(IB0 + i) = (d.inc >> i) & 1;
}
// the multiplicity columns: all counts in the first copy of the periodic tables
__global__ __launch_bounds__(256) void k_blake_mult(const uint32_t* hist, uint64_t* tr, size_t n) {
    const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (row >= n) return;
    tr[(size_t)blk::M1 * n + row] = row < 65536 ? hist[row] : 0;
    tr[(size_t)blk::M2 * n + row] = row < 65536 ? hist[65536 + row] : 0;
}

// ---- auxiliary columns (logUp) --------------------------------------------------------------------------------------
// One lane per row i (as the "next" row of the pair (i-1, i)): the 132 helper elements of row i (two lookups each,
// h = m (1/D_u + 1/D_v); the four pairs of a group share ONE extension-field inversion, Montgomery's trick), the table
// helper of row i, and the running-sum increment Z(i) - Z(i-1) = sum_e h_e(i) - ht(i-1), stored at row i-1 and turned
// into Z by an exclusive scan.
struct AuxArgs {
    const uint64_t* tr;
    uint64_t* aux;
    size_t n;
    gl2 beta, gamma;
};
__device__ __forceinline__ gl2 gl2_from(uint64_t x) { return {x, 0}; }
__global__ __launch_bounds__(256) void k_blake_aux(AuxArgs a) {
    using namespace blk;
    const size_t n = a.n, i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t ip = (i + n - 1) & (n - 1);
    const int rl = (int)(ip & 15);  // row index (mod 16) of the LOCAL row: selectors are taken there
    const bool g_on = rl <= 11, m3 = rl <= 13;
    const gl2 beta = a.beta, gamma = a.gamma, g2 = gl2_mul(gamma, gamma), g3 = gl2_mul(g2, gamma), g4 = gl2_mul(g2, g2);
    const gl2 bt2 = gl2_add(beta, g4);
    auto N = [&](int col) -> uint64_t { return a.tr[(size_t)col * n + i]; };
    auto L = [&](int col) -> uint64_t { return a.tr[(size_t)col * n + ip]; };
    auto out_byte_loc = [&](int w, int j) -> uint64_t {
        const int m = w & 3;
        if (w < 4) return L(GC(4 + w, S_A2, j));
        if (w < 8) {
            const int k = 4 + (m + 3) % 4;
            return 2 * L(GC(k, S_L, j)) + L(GC(k, S_T, (j + 7) & 7));
        }
        if (w < 12) return L(GC(4 + (m + 2) % 4, S_C2, j));
        return L(GC(4 + (m + 1) % 4, S_D2, j));
    };
    auto in_byte = [&](int k, int op, int j) -> uint64_t {
        if (k < 4) return out_byte_loc(4 * op + k, j);
        const int j0 = k - 4;
        if (op == 1) {
            const int kb = (j0 + 1) & 3;
            return 2 * N(GC(kb, S_L, j)) + N(GC(kb, S_T, (j + 7) & 7));
        }
        return N(GC((j0 + 3) & 3, S_D2, j));  // op == 3 (a and c never enter a lookup)
    };
    auto fp1 = [&](uint64_t x, uint64_t y, uint64_t z) -> gl2 {  // small operands: products by < 2^9 stay cheap but exact
        gl2 d = gl2_add(beta, gl2_add(gl2_scale(gamma, y), gl2_scale(g2, z)));
        d.a = gl_add(d.a, x);
        return d;
    };
    auto denom = [&](int k, int grp, int q) -> gl2 {
        if (grp == 0) return fp1(in_byte(k, 3, q), N(GC(k, S_A1, q)), N(GC(k, S_D1, (q + 4) & 7)));
        if (grp == 1) return fp1(in_byte(k, 1, q), N(GC(k, S_C1, q)), N(GC(k, S_B1, (q + 5) & 7)));
        if (grp == 2) return fp1(N(GC(k, S_D1, q)), N(GC(k, S_A2, q)), N(GC(k, S_D2, (q + 6) & 7)));
        gl2 d = gl2_add(bt2, gl2_add(gl2_scale(gamma, N(GC(k, S_C2, q))), gl2_add(gl2_scale(g2, N(GC(k, S_L, q))), gl2_scale(g3, N(GC(k, S_T, q))))));
        d.a = gl_add(d.a, N(GC(k, S_B1, q)));
        return d;
    };
    auto store = [&](int e, gl2 h) {
        a.aux[(size_t)(2 * e) * n + i] = h.a;
        a.aux[(size_t)(2 * e + 1) * n + i] = h.b;
    };
    gl2 hsum{0, 0};
    // four pairs (p_q = D_u D_v, s_q = D_u + D_v) -> h_q = s_q / p_q with one inversion
    auto four = [&](gl2* p, gl2* s, int e0) {
        const gl2 c1 = gl2_mul(p[0], p[1]), c2 = gl2_mul(c1, p[2]), c3 = gl2_mul(c2, p[3]);
        gl2 inv = gl2_inv(c3);
        const gl2 i3 = gl2_mul(inv, c2);
        inv = gl2_mul(inv, p[3]);
        const gl2 i2 = gl2_mul(inv, c1);
        inv = gl2_mul(inv, p[2]);
        const gl2 i1 = gl2_mul(inv, p[0]), i0 = gl2_mul(inv, p[1]);
        const gl2 iq[4] = {i0, i1, i2, i3};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const gl2 h = gl2_mul(s[q], iq[q]);
            store(e0 + q, h);
            hsum = gl2_add(hsum, h);
        }
    };
#pragma unroll 1
    for (int k = 0; k < 8; ++k)
#pragma unroll 1
        for (int grp = 0; grp < 4; ++grp) {
            const int e0 = (k * 4 + grp) * 4;
            if (grp == 2 ? m3 : g_on) {
                gl2 p[4], s[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const gl2 du = denom(k, grp, 2 * q), dv = denom(k, grp, 2 * q + 1);
                    p[q] = gl2_mul(du, dv), s[q] = gl2_add(du, dv);
                }
                four(p, s, e0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) store(e0 + q, gl2{0, 0});
            }
        }
    {
        gl2 p[4], s[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t b0 = N(MB0 + 2 * q), b1 = N(MB0 + 2 * q + 1);
            const gl2 du = fp1(b0, 0, b0), dv = fp1(b1, 0, b1);
            p[q] = gl2_mul(du, dv), s[q] = gl2_add(du, dv);
        }
        four(p, s, HM0);
    }
    // table helper of this row and of the local row: ht = M1 / D_t1 + M2 / D_t2
    auto table_h = [&](size_t row, uint64_t m1, uint64_t m2) -> gl2 {
        const uint64_t ti = row & 65535, ta = ti & 255, tb = ti >> 8, x = ta ^ tb;
        const gl2 d1 = fp1(ta, tb, x);
        gl2 d2 = gl2_add(bt2, gl2_add(gl2_scale(gamma, tb), gl2_add(gl2_scale(g2, x & 127), gl2_scale(g3, x >> 7))));
        d2.a = gl_add(d2.a, ta);
        if ((m1 | m2) == 0) return gl2{0, 0};
        const gl2 num = gl2_add(gl2_scale(d2, m1), gl2_scale(d1, m2));
        return gl2_mul(num, gl2_inv(gl2_mul(d1, d2)));
    };
    const gl2 ht = table_h(i, N(M1), N(M2));
    store(HT, ht);
    const gl2 dz = gl2_sub(hsum, table_h(ip, L(M1), L(M2)));
    a.aux[(size_t)(2 * ZZ) * n + ip] = dz.a;
    a.aux[(size_t)(2 * ZZ + 1) * n + ip] = dz.b;
}

int32_t vx_blake_air_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, uint64_t* aux, uint64_t* aux_pub) {
    (void)aux_pub;
    const size_t n = (size_t)1 << log_n;
    AuxArgs a{trace, aux, n, gl2{chal[0], chal[1]}, gl2{chal[2], chal[3]}};
    hipLaunchKernelGGL(k_blake_aux, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a);
    VX_HIP(hipGetLastError());
    return vx_scan_cols_dev(ctx, aux + (size_t)(2 * blk::ZZ) * n, log_n, 2, nullptr);
}

extern "C" {

int32_t vx_blake_chain_trace(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes, size_t n_headers,
                             const uint8_t trusted_hash[32], uint32_t first_block_number, int log_n, vx_buf* trace_out,
                             uint64_t public_inputs_out[18], uint8_t* digests_out) {
    if (!ctx || !headers || !sizes || !trusted_hash || !trace_out || !public_inputs_out) return VX_ERR_ARG;
    VX_CHECK(stride % 128 == 0 && stride > 0, "blake trace: stride %zu must be a positive multiple of 128", stride);
    VX_CHECK(n_headers >= 1 && n_headers * stride <= headers->n * 8, "blake trace: headers exceed the buffer");
    VX_CHECK(log_n >= blk::TABLE_LOG && log_n <= 24, "blake trace: log_n %d out of range [16, 24] (the trace holds one copy of the 2^16-row lookup tables)", log_n);
    VX_CHECK(first_block_number >= (1u << 14) && (uint64_t)first_block_number + n_headers <= (1u << 30),
             "blake trace: block numbers %u.. are outside the 4-byte SCALE compact range [2^14, 2^30) this AIR covers", first_block_number);
    const size_t n = (size_t)1 << log_n, n_blocks = n >> 4;
    VX_CHECK(trace_out->n >= n * blk::COLS, "blake trace: trace buffer holds %zu < %zu elements", trace_out->n, n * (size_t)blk::COLS);
    std::vector<uint32_t> base(n_headers);
    size_t n_real = 0;
    for (size_t i = 0; i < n_headers; ++i) {
        VX_CHECK(sizes[i] <= stride && sizes[i] >= 36, "blake trace: header %zu has size %u", i, sizes[i]);
        base[i] = (uint32_t)n_real;
        n_real += (sizes[i] + 127) / 128;
    }
    VX_CHECK(n_real <= n_blocks, "blake trace: %zu compressions do not fit 2^%d rows (%zu blocks)", n_real, log_n, n_blocks);
    // device scratch: sizes | block_base | digests | hchain | descs
    const size_t w_sizes = (n_headers * 4 + 7) / 8, w_dig = n_headers * 4, w_hc = n_real * 8;
    const size_t w_desc = (n_blocks * sizeof(BlockDesc) + 7) / 8;
    const size_t w_hist = 65536;  // two tables x 2^16 uint32 counters
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, 2 * w_sizes + w_dig + w_hc + w_desc + w_hist, &sc));
    uint32_t* d_sizes = (uint32_t*)sc;
    uint32_t* d_base = (uint32_t*)(sc + w_sizes);
    uint8_t* d_dig = (uint8_t*)(sc + 2 * w_sizes);
    uint64_t* d_hc = sc + 2 * w_sizes + w_dig;
    BlockDesc* d_desc = (BlockDesc*)(d_hc + w_hc);
    uint32_t* d_hist = (uint32_t*)(d_hc + w_hc + w_desc);
    VX_HIP(hipMemcpyAsync(d_sizes, sizes, n_headers * 4, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemcpyAsync(d_base, base.data(), n_headers * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_blake_chain, dim3((unsigned)((n_headers + 63) / 64)), dim3(64), 0, ctx->stream, (const uint8_t*)headers->d,
                       stride, (const uint32_t*)d_sizes, n_headers, (const uint32_t*)d_base, d_hc, d_dig);
    VX_HIP(hipGetLastError());
    std::vector<uint8_t> dig(32 * n_headers);
    VX_HIP(hipMemcpyAsync(dig.data(), d_dig, dig.size(), hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    // block descriptors (host): the link rule is checked here too, so a broken chain fails loudly
    std::vector<BlockDesc> descs(n_blocks);
    uint8_t D[32];
    memcpy(D, trusted_hash, 32);
    size_t bi = 0;
    for (size_t i = 0; i < n_headers; ++i) {
        const uint32_t nch = (sizes[i] + 127) / 128;
        for (uint32_t cidx = 0; cidx < nch; ++cidx, ++bi) {
            BlockDesc& d = descs[bi];
            memset(&d, 0, sizeof d);
            d.fin = cidx + 1 == nch;
            d.first = cidx == 0;
            d.act = 1;
            d.inc = d.fin ? sizes[i] - 128 * cidx : 128;
            d.t = d.fin ? sizes[i] : 128 * (cidx + 1);
            d.msg_off = i * stride + 128 * (size_t)cidx;
            d.num = first_block_number + (uint32_t)i;
            memcpy(d.D, D, 32);
        }
        memcpy(D, dig.data() + 32 * i, 32);
    }
    for (; bi < n_blocks; ++bi) {
        BlockDesc& d = descs[bi];
        memset(&d, 0, sizeof d);
        d.fin = d.first = 1;
        d.act = 0;
        d.inc = d.t = 36;
        d.msg_off = ~0ULL;
        d.num = first_block_number + (uint32_t)n_headers - 1;
        memcpy(d.D, D, 32);
    }
    VX_HIP(hipMemcpyAsync(d_desc, descs.data(), n_blocks * sizeof(BlockDesc), hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemsetAsync(d_hist, 0, w_hist * 8, ctx->stream));
    hipLaunchKernelGGL(k_blake_trace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint8_t*)headers->d,
                       (const BlockDesc*)d_desc, (const uint64_t*)d_hc, n_real, trace_out->d, d_hist, n);
    VX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_blake_mult, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_hist, trace_out->d, n);
    VX_HIP(hipGetLastError());
    VX_HIP(hipStreamSynchronize(ctx->stream));  // descs must outlive the kernel
    for (int j = 0; j < 8; ++j) {
        uint32_t a, b;
        memcpy(&a, trusted_hash + 4 * j, 4);
        memcpy(&b, D + 4 * j, 4);
        public_inputs_out[j] = a;
        public_inputs_out[8 + j] = b;
    }
    public_inputs_out[16] = first_block_number;
    public_inputs_out[17] = first_block_number + (uint64_t)n_headers - 1;
    if (digests_out) memcpy(digests_out, dig.data(), dig.size());
    return VX_OK;
}

static const uint64_t VX_HR_MAGIC = 0x3345474e41525248ULL;  // "HRRANGE3"
static const size_t VX_HR_HDR = 18;  // magic, max_headers, trusted, target, out96 (12), len(blake proof), len(sha proof)

static int sha_log_n(size_t n_keys) {
    int log_n = 6;
    while (((size_t)1 << log_n) < 64 * (2 * n_keys - 1)) ++log_n;
    return log_n;
}

int32_t vx_header_range_proof_bound(const vx_stark_config* cfg, size_t n_chunks, size_t n_authorities, size_t* n_words) {
    if (!cfg || !n_words || n_chunks == 0) return VX_ERR_ARG;
    int log_n = blk::TABLE_LOG;
    while (((size_t)1 << log_n) < 16 * n_chunks) ++log_n;
    size_t w1 = 0, w2 = 0;
    int32_t rc = vx_stark_proof_bound(VX_AIR_BLAKE_CHAIN, cfg, log_n, &w1);
    if (rc == VX_OK && n_authorities) rc = vx_stark_proof_bound(VX_AIR_SHA_CHAIN, cfg, sha_log_n(n_authorities), &w2);
    *n_words = w1 + w2 + VX_HR_HDR;
    return rc;
}

int32_t vx_header_range_prove(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes, size_t n_fetched,
                              uint32_t max_headers, uint32_t trusted_block, const uint8_t trusted_hash[32], uint32_t target_block,
                              const vx_justification* just, const vx_stark_config* cfg, uint8_t out96[96], uint64_t* proof_out,
                              size_t proof_cap, size_t* proof_len) {
    if (!ctx || !cfg || !proof_len || !out96) return VX_ERR_ARG;
    // 1. statement + public outputs (map/reduce chain rules, Merkle roots)
    VX_TRY(vx_verify_subchain(ctx, headers, stride, sizes, n_fetched, max_headers, trusted_block, trusted_hash, target_block, out96));
    // 1b. the target header is justified by > 2/3 of the committed authority set (header_range.rs:49-54): checked on
    //     the side context together with the commitment proof (below), while this context proves the hash chain
    const bool room = proof_out && proof_cap > VX_HR_HDR;
    // 3. authority-set commitment STARK (compute_authority_set_commitment, justification.rs:127-162): independent of
    //    the hash-chain proof and small, so it runs on the side context from a host thread while this one proves
    std::vector<uint64_t> sha_proof;
    size_t len1 = 0, len2 = 0;
    int32_t rc_sha = VX_OK;
    std::thread sha_thread;
    vx_ctx* side = just ? vx_side_ctx(ctx) : nullptr;
    auto prove_sha = [&](vx_ctx* c) -> int32_t {
        (void)hipSetDevice(c->device);
        int32_t rj = vx_verify_simple_justification(c, target_block, out96, just->authority_set_id, just->authority_set_hash, just->precommit,
                                                    just->pubkeys, just->signatures, just->validator_signed, just->num_authorities,
                                                    just->max_authorities);
        if (rj != VX_OK) return rj;
        const int sl = sha_log_n(just->num_authorities);
        size_t bound = 0;
        int32_t r = vx_stark_proof_bound(VX_AIR_SHA_CHAIN, cfg, sl, &bound);
        if (r != VX_OK) return r;
        sha_proof.resize(bound);
        vx_buf* st = nullptr;
        r = vx_alloc(c, ((size_t)VX_SHA_AIR_COLS) << sl, &st);
        if (r != VX_OK) return r;
        uint64_t spub[8];
        uint8_t com[32];
        r = vx_sha_chain_trace(c, just->pubkeys, just->num_authorities, sl, st, spub, com);
        if (r == VX_OK && memcmp(com, just->authority_set_hash, 32) != 0) r = vx_fail(c, VX_ERR_STATEMENT, "header_range: authority-set commitment mismatch");
        if (r == VX_OK) r = vx_stark_prove_impl(c, VX_AIR_SHA_CHAIN, cfg, st->d, st->n, /*consume_trace=*/1, sl, spub, 8, sha_proof.data(), sha_proof.size(), &len2);
        (void)vx_free(c, st);
        return r;
    };
    if (side) {
        try {
            sha_thread = std::thread([&] { rc_sha = prove_sha(side); });
        } catch (...) {  // no thread to be had: prove one after the other below
            side = nullptr;
        }
    }
    // 2. Blake2b parent-hash-chain STARK over every compression of every header
    size_t chunks = 0;
    for (size_t i = 0; i < n_fetched; ++i) chunks += (sizes[i] + 127) / 128;
    int log_n = blk::TABLE_LOG;  // at least one copy of the lookup tables
    while (((size_t)1 << log_n) < 16 * chunks) ++log_n;
    vx_buf* trace = nullptr;
    int32_t rc = vx_alloc(ctx, ((size_t)blk::COLS) << log_n, &trace);
    uint64_t pub[18];
    if (rc == VX_OK) rc = vx_blake_chain_trace(ctx, headers, stride, sizes, n_fetched, trusted_hash, trusted_block + 1, log_n, trace, pub, nullptr);
    if (rc == VX_OK) {
        uint8_t tgt[32];
        for (int j = 0; j < 8; ++j) {
            uint32_t l = (uint32_t)pub[8 + j];
            memcpy(tgt + 4 * j, &l, 4);
        }
        if (memcmp(tgt, out96, 32) != 0) rc = vx_fail(ctx, VX_ERR_STATEMENT, "header_range: chain digest differs from the subchain target hash");
    }
    if (rc == VX_OK)
        rc = vx_stark_prove_impl(ctx, VX_AIR_BLAKE_CHAIN, cfg, trace->d, trace->n, /*consume_trace=*/1, log_n, pub, 18, room ? proof_out + VX_HR_HDR : nullptr, room ? proof_cap - VX_HR_HDR : 0, &len1);
    if (trace) (void)vx_free(ctx, trace);
    if (sha_thread.joinable()) sha_thread.join();
    else if (just) rc_sha = prove_sha(ctx);  // no side context: one after the other
    if (just) {
        if (rc_sha != VX_OK && side) (void)vx_fail(ctx, rc_sha, "%s", vx_last_error(side));
        if ((rc == VX_OK || rc == VX_ERR_BUFSZ) && rc_sha != VX_OK) rc = rc_sha;
        if (rc == VX_OK) {
            if (proof_out && proof_cap >= VX_HR_HDR + len1 + len2) memcpy(proof_out + VX_HR_HDR + len1, sha_proof.data(), len2 * 8);
            else rc = vx_fail(ctx, VX_ERR_BUFSZ, "header_range: proof needs %zu words, buffer has %zu", VX_HR_HDR + len1 + len2, proof_cap);
        }
    }
    *proof_len = VX_HR_HDR + len1 + len2;
    if (rc != VX_OK) return rc;
    proof_out[0] = VX_HR_MAGIC;
    proof_out[1] = max_headers;
    proof_out[2] = trusted_block;
    proof_out[3] = target_block;
    memcpy(proof_out + 4, out96, 96);
    proof_out[16] = len1;
    proof_out[17] = len2;
    return VX_OK;
}
}
