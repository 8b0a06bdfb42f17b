// K2/K3: batched Goldilocks NTT and low-degree extension for gfx950.
//
// Replaces plonky2_field::fft::{fft,ifft} and PolynomialBatch::{from_values,
// from_coeffs,lde} (plonky2 v0.2.0 fri/oracle.rs), reached from every
// `circuit.prove` in the reference (circuits/header_range.rs:167).
//
// Design (not a translation of the CPU radix-2 loop): a transform of size
// n = 2^L is decomposed four-step style into <= 3 in-place passes over HBM;
// each pass stages a [R rows x T columns] tile (R*T = 4096 elements, 32 KiB)
// in LDS, runs log2(R) butterfly stages there, and applies the inter-pass
// twiddle w_S^(i*k) from a three-level power table while the tile moves
// between LDS and HBM.  Rows of a tile are strided in HBM, the T columns are
// contiguous (>= 128 B segments), so every global access is coalesced.
//   DIF-like plan: natural in  -> bit-reversed out   (used for the inverse)
//   DIT-like plan: bit-reversed in -> natural out    (used for the forward/LDE)
// so values -> coefficients -> coset evaluations needs NO permutation pass:
// coefficients simply live in bit-reversed positions in between.  The LDE's
// zero padding and coset scaling (c_k *= shift^k) are fused into the first
// DIT pass, which reads only the n = N >> rate_bits real coefficients.
#include <stdlib.h>
#include <utility>

#include "gl.cuh"
#include "vx_internal.h"

struct PassArgs {
    const uint64_t* src;
    uint64_t* dst;
    size_t src_col_stride, dst_col_stride;
    int log_sub;   // S = 2^log_sub: size of the independent sub-arrays at this level
    int log_rows;  // R = 2^log_rows rows per tile (butterfly stages done in LDS)
    int log_T;     // T = 2^log_T contiguous columns per tile
    const uint64_t* w12;  // w_4096^e, e < 2048 (forward or inverse)
    const uint64_t* tw;   // three-level powers of w_{2^32} (forward or inverse)
    uint64_t scale;       // DIF mode: multiply on store when > 1 (1/n of the inverse)
    int expand_bits;      // DIT first pass of an LDE: src holds S >> expand_bits coefficients
    int log_coeff;        // log2 of the coefficient count (bit-reversal width for shift^k)
    const uint64_t* shift_tab;  // three-level powers of the coset shift, or null
    const uint64_t* tw2_lo;     // two-level table of w_{2^log_sub}: lo[j] = w^j, hi[j] = w^(j << tw2_bits)
    const uint64_t* tw2_hi;
    int tw2_bits;
    unsigned tw2_total;  // entries of lo | hi back to back (k_ntt3 stages them in LDS when they fit)
    size_t n_cols;       // k_ntt3: a block may take several columns (blockIdx.y counts groups of n3::cols_per_block)
    size_t n_tiles;  // tiles per column in this pass (a block takes NTT_TPB consecutive ones)
    int dbg_skip;  // timing experiments only (VX_NTT_SKIP bit mask): 1 = tile twiddles, 2 = inter-pass twiddle, 4 = butterflies
};

__device__ __forceinline__ uint64_t tab3_pow(const uint64_t* tab, uint64_t e) {
    uint64_t r = tab[e & 2047];
    uint32_t e1 = (uint32_t)(e >> 11) & 2047, e2 = (uint32_t)(e >> 22);
    if (e1) r = gl_mul(r, tab[2048 + e1]);
    if (e2) r = gl_mul(r, tab[4096 + e2]);
    return r;
}
// w_{2^log_s}^e from the w_{2^32} table
__device__ __forceinline__ uint64_t root_pow(const uint64_t* tw, uint64_t e, int log_s) {
    uint64_t E = (e << (32 - log_s)) & 0xFFFFFFFFULL;
    uint64_t r = tw[4096 + (E >> 22)];
    if (log_s > 10) r = gl_mul(r, tw[2048 + ((E >> 11) & 2047)]);
    if (log_s > 21) r = gl_mul(r, tw[E & 2047]);
    return r;
}

// MODE 0: DIF-like pass, MODE 1: DIT-like pass (see file header).
template <int MODE>
__global__ __launch_bounds__(256) void k_ntt_pass(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const int lr = a.log_rows, lT = a.log_T;
    const int T = 1 << lT;
    const int pitch = T > 1 ? T + 1 : 1;
    const int nelem = 1 << (lr + lT);
    const size_t m = (size_t)1 << (a.log_sub - lr);  // row stride inside a sub-array
    const size_t tiles_per_sub = m >> lT;
    const size_t tile = blockIdx.x;
    const size_t sub = tile / tiles_per_sub;
    const size_t col0 = (tile - sub * tiles_per_sub) << lT;
    const size_t base = (sub << a.log_sub) + col0;
    const uint64_t* src = a.src + blockIdx.y * a.src_col_stride;
    uint64_t* dst = a.dst + blockIdx.y * a.dst_col_stride;
    const int tid = threadIdx.x;

    for (int idx = tid; idx < nelem; idx += 256) {
        int col = idx & (T - 1), row = idx >> lT;
        size_t g = base + (size_t)row * m + col;
        uint64_t v;
        if (MODE == 1 && a.expand_bits) {
            v = 0;
            if ((g & (((size_t)1 << a.expand_bits) - 1)) == 0) {
                size_t q = g >> a.expand_bits;
                v = src[q];
                if (a.shift_tab) v = gl_mul(v, tab3_pow(a.shift_tab, brev32((uint32_t)q, a.log_coeff)));
            }
        } else {
            v = src[g];
            if (MODE == 1 && a.shift_tab && m == 1)  // plain coset transform: first pass, c_k *= shift^k
                v = gl_mul(v, tab3_pow(a.shift_tab, brev32((uint32_t)g, a.log_coeff)));
        }
        if (MODE == 1 && m > 1) {
            uint64_t e = (uint64_t)(col0 + col) * brev32((uint32_t)row, lr);
            v = gl_mul(v, root_pow(a.tw, e, a.log_sub));
        }
        lds[row * pitch + col] = v;
    }
    __syncthreads();

    const int nbf = nelem >> 1;
    if (MODE == 0) {
        for (int s = lr - 1; s >= 0; --s) {
            const int h = 1 << s;
            for (int bf = tid; bf < nbf; bf += 256) {
                int col = bf & (T - 1), pr = bf >> lT;
                int j = pr & (h - 1);
                int r0 = ((pr >> s) << (s + 1)) + j;
                uint64_t u = lds[r0 * pitch + col], v = lds[(r0 + h) * pitch + col];
                lds[r0 * pitch + col] = gl_add(u, v);
                lds[(r0 + h) * pitch + col] = gl_mul(gl_sub(u, v), a.w12[j << (11 - s)]);
            }
            __syncthreads();
        }
    } else {
        for (int s = 0; s < lr; ++s) {
            const int h = 1 << s;
            for (int bf = tid; bf < nbf; bf += 256) {
                int col = bf & (T - 1), pr = bf >> lT;
                int j = pr & (h - 1);
                int r0 = ((pr >> s) << (s + 1)) + j;
                uint64_t u = lds[r0 * pitch + col];
                uint64_t v = gl_mul(lds[(r0 + h) * pitch + col], a.w12[j << (11 - s)]);
                lds[r0 * pitch + col] = gl_add(u, v);
                lds[(r0 + h) * pitch + col] = gl_sub(u, v);
            }
            __syncthreads();
        }
    }

    for (int idx = tid; idx < nelem; idx += 256) {
        int col = idx & (T - 1), row = idx >> lT;
        uint64_t v = lds[row * pitch + col];
        if (MODE == 0) {
            if (m > 1) {
                uint64_t e = (uint64_t)(col0 + col) * brev32((uint32_t)row, lr);
                v = gl_mul(v, root_pow(a.tw, e, a.log_sub));
            }
            if (a.scale > 1) v = gl_mul(v, a.scale);
        }
        dst[base + (size_t)row * m + col] = v;
    }
}


// ---------------------------------------------------------------------------------------------
// v2 tile kernel: register-resident radix-16 rounds.  A 4096-element tile (index iota = rho*T +
// tau, rho = row, tau = column) is held 16 elements per lane; one ROUND transforms the 4-bit
// window [f, f+4) of iota (f = 8, 4, 0) entirely in registers, so a 12-stage pass needs 2-3 LDS
// exchanges instead of 12 barrier-separated LDS stages.  Inside a round every twiddle is a
// power of w_16 = 2^156 = -2^60 (mod p): multiplications by w_16^k are 64-bit shifts plus the
// 2^64 = 2^32 - 1 reduction, no integer multiplier.  General multiplications remain only for
// the between-round twiddle w^(rho_low * k) (one per element per round, table w_4096^e) and
// the between-pass twiddle of the four-step plan.  DIT rounds are the transposed flow graph.
template <int R_>
__device__ __forceinline__ uint64_t gl_shl(uint64_t x) {  // x * 2^R_, canonical in/out, 0 <= R_ < 96
    if constexpr (R_ == 0) return x;
    else if constexpr (R_ < 64) return gl_reduce128(x >> (64 - R_), x << R_);
    else {
        constexpr int s = R_ - 64;
        const uint64_t A = s ? (x >> (64 - s)) : 0, B = x << s;
        const uint32_t bh = (uint32_t)(B >> 32), bl = (uint32_t)B;
        uint64_t t = ((uint64_t)bl << 32) - bl;  // bl * (2^32 - 1) < p
        t = gl_sub(t, bh);
        return gl_sub(t, A << 32);
    }
}
// |w_16^K| as a shift, and its sign: forward (156K mod 192), inverse (-156K mod 192)
template <int K, int INV>
struct W16 {
    static constexpr int E = ((INV ? 192 - 156 : 156) * K) % 192;
    static constexpr int SH = E % 96;
    static constexpr bool NEG = E >= 96;
};
template <int K, int INV>
__device__ __forceinline__ void bfly_dif(uint64_t& u, uint64_t& v) {  // (u+v, (u-v) w_16^K)
    const uint64_t s = gl_add(u, v);
    const uint64_t d = W16<K, INV>::NEG ? gl_sub(v, u) : gl_sub(u, v);
    u = s;
    v = gl_shl<W16<K, INV>::SH>(d);
}
template <int K, int INV>
__device__ __forceinline__ void bfly_dit(uint64_t& u, uint64_t& v) {  // (u + v w, u - v w)
    const uint64_t t = gl_shl<W16<K, INV>::SH>(v);
    const uint64_t a = gl_add(u, t), b = gl_sub(u, t);
    u = W16<K, INV>::NEG ? b : a;
    v = W16<K, INV>::NEG ? a : b;
}
// DFT over the top Q bits of the register index e (2^(4-Q) independent groups)
template <int Q, int INV, int S, int E>
__device__ __forceinline__ void dif_stage_pair(uint64_t* x) {
    // S = stage (span 2^S in field index), E = register index of the upper element's partner base
    constexpr int G = 4 - Q;                 // low bits = group
    constexpr int a = E >> G;                // field index
    if constexpr (((a >> S) & 1) == 0) {
        constexpr int h = 1 << S;
        constexpr int j = a & (h - 1);
        constexpr int K = j * (8 >> S);      // w_{2h}^j = w_16^(j * 8/h)
        bfly_dif<K, INV>(x[E], x[E + (h << G)]);
    }
}
template <int Q, int INV, int S, int E>
__device__ __forceinline__ void dit_stage_pair(uint64_t* x) {
    constexpr int G = 4 - Q;
    constexpr int a = E >> G;
    if constexpr (((a >> S) & 1) == 0) {
        constexpr int h = 1 << S;
        constexpr int j = a & (h - 1);
        constexpr int K = j * (8 >> S);
        bfly_dit<K, INV>(x[E], x[E + (h << G)]);
    }
}
template <int Q, int INV, int S, int... Es>
__device__ __forceinline__ void dif_stage(uint64_t* x, std::integer_sequence<int, Es...>) {
    (dif_stage_pair<Q, INV, S, Es>(x), ...);
}
template <int Q, int INV, int S, int... Es>
__device__ __forceinline__ void dit_stage(uint64_t* x, std::integer_sequence<int, Es...>) {
    (dit_stage_pair<Q, INV, S, Es>(x), ...);
}
template <int Q, int INV>
__device__ __forceinline__ void dif_round_q(uint64_t* x) {
    using I = std::make_integer_sequence<int, 16>;
    if constexpr (Q >= 4) dif_stage<Q, INV, 3>(x, I{});
    if constexpr (Q >= 3) dif_stage<Q, INV, 2>(x, I{});
    if constexpr (Q >= 2) dif_stage<Q, INV, 1>(x, I{});
    dif_stage<Q, INV, 0>(x, I{});
}
template <int Q, int INV>
__device__ __forceinline__ void dit_round_q(uint64_t* x) {
    using I = std::make_integer_sequence<int, 16>;
    dit_stage<Q, INV, 0>(x, I{});
    if constexpr (Q >= 2) dit_stage<Q, INV, 1>(x, I{});
    if constexpr (Q >= 3) dit_stage<Q, INV, 2>(x, I{});
    if constexpr (Q >= 4) dit_stage<Q, INV, 3>(x, I{});
}
template <int MODE, int INV>
__device__ __forceinline__ void tile_round(uint64_t* x, int q) {
    switch (q) {  // wave-uniform
    case 4: MODE ? dit_round_q<4, INV>(x) : dif_round_q<4, INV>(x); break;
    case 3: MODE ? dit_round_q<3, INV>(x) : dif_round_q<3, INV>(x); break;
    case 2: MODE ? dit_round_q<2, INV>(x) : dif_round_q<2, INV>(x); break;
    default: MODE ? dit_round_q<1, INV>(x) : dif_round_q<1, INV>(x); break;
    }
}
__device__ __forceinline__ int tile_iota(int tid, int e, int f) {
    return (tid & ((1 << f) - 1)) | (e << f) | ((tid >> f) << (f + 4));
}
__device__ __forceinline__ int tile_pad(int i) { return i + (i >> 4); }
__device__ __forceinline__ void tile_exchange(uint64_t* x, uint64_t* lds, int tid, int f_from, int f_to) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) lds[tile_pad(tile_iota(tid, e, f_from))] = x[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = lds[tile_pad(tile_iota(tid, e, f_to))];
}
// multiply x[e] by w_{2^(b+q)}^(rho_low * k), k = bitrev_q(field index of e), rho_low = rho mod 2^b
__device__ __forceinline__ void tile_twiddle(uint64_t* x, const uint64_t* w12, int tid, int f, int q, int b, int lT) {
    const int g = 4 - q;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = (int)(__brev((unsigned)(e >> g)) >> (32 - q));
        const int rho_low = (tile_iota(tid, e, f) >> lT) & ((1 << b) - 1);
        const int idx = (rho_low * k) << (12 - b - q);
        if (idx) {
            const uint64_t w = w12[idx & 2047];
            const uint64_t v = gl_mul(x[e], w);
            x[e] = (idx & 2048) ? gl_neg(v) : v;
        }
    }
}

// One block walks NTT_TPB consecutive tiles of a column, so the 16 KB twiddle table is staged into LDS once per
// NTT_TPB tiles of 32 KB.  (Tried: loading tile t+1 into registers while tile t is in the ALU rounds -- 256+ VGPRs,
// spills, two waves per SIMD: 10.3 ms instead of 6.2 for 2^19 x 1024.)
#ifndef VX_NTT_TPB
#define VX_NTT_TPB 4
#endif
constexpr int NTT_TPB = VX_NTT_TPB;
template <int MODE, int INV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_ntt_tile(PassArgs a) {  // LDS allows 3 blocks per CU

    __shared__ __attribute__((aligned(16))) uint64_t lds[4096 + 256 + 2048];
    uint64_t* const w12s = lds + 4096 + 256;  // w_4096^e, e < 2048, staged once: the per-element
    const int lr = a.log_rows, lT = 12 - lr, T = 1 << lT;  // twiddle lookups are LDS reads, not gathers
    int tid = threadIdx.x;
    if (lr > 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) w12s[tid + 256 * j] = a.w12[tid + 256 * j];
        // visible after the first tile_exchange barrier, which precedes every use... except a
        // twiddle right after round A: make it explicit
        __syncthreads();
    }
    const size_t m = (size_t)1 << (a.log_sub - lr);
    const size_t tiles_per_sub = m >> lT;
    const uint64_t* src = a.src + blockIdx.y * a.src_col_stride;
    uint64_t* dst = a.dst + blockIdx.y * a.dst_col_stride;
    const int nr = (lr + 3) >> 2;
    const int qA = lr < 4 ? lr : 4, qB = lr - 4 < 4 ? lr - 4 : 4, qC = lr - 8;
    const size_t tile0 = (size_t)blockIdx.x * NTT_TPB;
    const int n_here = (int)(a.n_tiles - tile0 < (size_t)NTT_TPB ? a.n_tiles - tile0 : (size_t)NTT_TPB);
    auto tile_base = [&](size_t tile, size_t& col0) -> size_t {
        const size_t sub = tile / tiles_per_sub;
        col0 = (tile - sub * tiles_per_sub) << lT;
        return (sub << a.log_sub) + col0;
    };
    const int f_first = nr == 1 ? 8 : (nr == 2 ? 4 : 0);
    const int f_load = MODE == 0 ? 8 : (f_first == 0 ? 8 : f_first);
    // raw loads of one tile (no arithmetic): element e of this lane sits at iota(tid, e, f_load)
    auto load_raw = [&](size_t tile, uint64_t* raw) {
        size_t col0;
        const size_t base = tile_base(tile, col0);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = tile_iota(tid, e, f_load);
            const size_t g = base + (size_t)(i >> lT) * m + (i & (T - 1));
            if (MODE == 1 && a.expand_bits) {
                raw[e] = 0;
                if ((g & (((size_t)1 << a.expand_bits) - 1)) == 0) raw[e] = src[g >> a.expand_bits];
            } else {
                raw[e] = src[g];
            }
        }
    };
    // VX_NTT_PREFETCH=1: the DIT passes (96 VGPRs) hold the next tile's 16 raw values per lane while the current tile is
    // in its rounds (126 VGPRs, still 3 blocks per CU).  Measured: LDE 2^19 x 1024 21.97 ms with, 21.67 ms without -- the
    // pass is not waiting for its loads; off by default.
#ifndef VX_NTT_PREFETCH
#define VX_NTT_PREFETCH 0
#endif
    constexpr bool PF = VX_NTT_PREFETCH && MODE == 1;
    uint64_t x[16], nx[PF ? 16 : 1];
    if (PF) load_raw(tile0, nx);
#pragma unroll 1
    for (int t = 0; t < n_here; ++t) {
        // keep per-element index math and twiddle lookups inside the loop: hoisted they cost 100+ VGPRs (occupancy)
        asm volatile("" : "+v"(tid));
        size_t col0;
        const size_t base = tile_base(tile0 + t, col0);
        if (PF) {
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = nx[PF ? e : 0];
            if (t + 1 < n_here) load_raw(tile0 + t + 1, nx);  // in flight during the rounds below
        } else {
            load_raw(tile0 + t, x);
        }
        if (MODE == 0) {
            const bool sk_tw = a.dbg_skip & 1, sk_ip = a.dbg_skip & 2, sk_bf = a.dbg_skip & 4;
            if (!sk_bf) tile_round<0, INV>(x, qA);
            if (lr - qA > 0 && !sk_tw) tile_twiddle(x, w12s, tid, 8, qA, lr - qA, lT);
            int f_last = 8;
            if (nr >= 2) {
                tile_exchange(x, lds, tid, 8, 4);
                if (!sk_bf) tile_round<0, INV>(x, qB);
                if (lr - 4 - qB > 0 && !sk_tw) tile_twiddle(x, w12s, tid, 4, qB, lr - 4 - qB, lT);
                f_last = 4;
            }
            if (nr == 3) {
                tile_exchange(x, lds, tid, 4, 0);
                if (!sk_bf) tile_round<0, INV>(x, qC);
                tile_exchange(x, lds, tid, 0, 8);  // back to the coalesced mapping for the store
                f_last = 8;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = tile_iota(tid, e, f_last);
                const int rho = i >> lT, tau = i & (T - 1);
                uint64_t v = x[e];
                if (m > 1 && !sk_ip) {
                    const uint64_t ex = (uint64_t)(col0 + tau) * brev32((uint32_t)rho, lr);
                    const uint64_t w = gl_mul_nc(a.tw2_hi[ex >> a.tw2_bits], a.tw2_lo[ex & (((uint64_t)1 << a.tw2_bits) - 1)]);
                    v = gl_mul(v, w);
                }
                if (a.scale > 1) v = gl_mul(v, a.scale);
                dst[base + (size_t)rho * m + tau] = v;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = tile_iota(tid, e, f_load);
                const int rho = i >> lT, tau = i & (T - 1);
                const size_t g = base + (size_t)rho * m + tau;
                uint64_t v = x[e];
                if (a.expand_bits) {
                    if (a.shift_tab && (g & (((size_t)1 << a.expand_bits) - 1)) == 0)
                        v = gl_mul(v, tab3_pow(a.shift_tab, brev32((uint32_t)(g >> a.expand_bits), a.log_coeff)));
                } else if (a.shift_tab && m == 1) {
                    v = gl_mul(v, tab3_pow(a.shift_tab, brev32((uint32_t)g, a.log_coeff)));
                }
                if (m > 1) {
                    const uint64_t ex = (uint64_t)(col0 + tau) * brev32((uint32_t)rho, lr);
                    const uint64_t w = gl_mul_nc(a.tw2_hi[ex >> a.tw2_bits], a.tw2_lo[ex & (((uint64_t)1 << a.tw2_bits) - 1)]);
                    v = gl_mul(v, w);
                }
                x[e] = v;
            }
            if (nr == 3) {
                tile_exchange(x, lds, tid, 8, 0);
                tile_round<1, INV>(x, qC);
                tile_exchange(x, lds, tid, 0, 4);
            }
            if (nr >= 2) {
                if (lr - 4 - qB > 0) tile_twiddle(x, w12s, tid, 4, qB, lr - 4 - qB, lT);
                tile_round<1, INV>(x, qB);
                tile_exchange(x, lds, tid, 4, 8);
            }
            if (lr - qA > 0) tile_twiddle(x, w12s, tid, 8, qA, lr - qA, lT);
            tile_round<1, INV>(x, qA);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = tile_iota(tid, e, 8);
                dst[base + (size_t)(i >> lT) * m + (i & (T - 1))] = x[e];
            }
        }
    }
}

#include "ntt3.cuh"

// out[i] = in[bitrev(i)] (per column); in != out
__global__ void k_bitrev_copy(const uint64_t* in, uint64_t* out, int log_n, size_t in_stride, size_t out_stride) {
    size_t n = (size_t)1 << log_n;
    const uint64_t* s = in + blockIdx.y * in_stride;
    uint64_t* d = out + blockIdx.y * out_stride;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        d[i] = s[brev32((uint32_t)i, log_n)];
}
// c_k = c_k * mulc * base^k (natural order); tab may be null (base = 1)
__global__ void k_scale_pow(uint64_t* d, int log_n, size_t stride, const uint64_t* tab, uint64_t mulc) {
    size_t n = (size_t)1 << log_n;
    uint64_t* c = d + blockIdx.y * stride;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t v = c[i];
        if (mulc != 1) v = gl_mul(v, mulc);
        if (tab) v = gl_mul(v, tab3_pow(tab, i));
        c[i] = v;
    }
}
__global__ void k_gather_rows(const uint64_t* lde, int log_N, size_t n_cols, const uint64_t* idx, size_t n_idx,
                              uint64_t* out) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= n_idx * n_cols) return;
    size_t k = t / n_cols, c = t - k * n_cols;
    size_t row = brev32((uint32_t)idx[k], log_N);
    out[t] = lde[(c << log_N) + row];
}

// ---------------------------------------------------------------- host planning
static void plan_chunks(int L, std::vector<int>& outer, int& last) {
    last = L < 12 ? L : 12;
    int rem = L - last;
    outer.clear();
    if (rem > 0) {
        int k = (rem + 7) / 8;
        for (int i = 0; i < k; ++i) outer.push_back(rem / k + (i < rem % k ? 1 : 0));
    }
}
static inline size_t lds_bytes(int lr, int lT) {
    size_t T = (size_t)1 << lT;
    return ((size_t)1 << lr) * (T > 1 ? T + 1 : 1) * 8;
}
static bool g_ntt_v1 = getenv("VX_NTT_V1") != nullptr;  // debugging aid: force the LDS-stage kernel
static bool g_ntt_v2 = getenv("VX_NTT_V2") != nullptr;  // A/B aid: the run-time-shape tile kernel instead of k_ntt3
template <int MODE, int LR>
static void launch_ntt3(vx_ctx* ctx, PassArgs& a, unsigned gx, size_t n_cols, int inverse) {
    a.n_cols = n_cols;
    constexpr int NC = n3::cols_per_block(MODE, LR);
    const dim3 grid(gx, (unsigned)((n_cols + NC - 1) / NC)), block(256);
    if constexpr (MODE == 1 && LR == 12) {  // the zero-padding first pass of an LDE (forward only: the callers never expand an inverse)
        if (a.expand_bits == 1) { hipLaunchKernelGGL((k_ntt3<1, 0, 12, 1>), grid, block, 0, ctx->stream, a); return; }
        if (a.expand_bits == 2) { hipLaunchKernelGGL((k_ntt3<1, 0, 12, 2>), grid, block, 0, ctx->stream, a); return; }
        if (a.expand_bits == 3) { hipLaunchKernelGGL((k_ntt3<1, 0, 12, 3>), grid, block, 0, ctx->stream, a); return; }
    }
    if (inverse) hipLaunchKernelGGL((k_ntt3<MODE, 1, LR>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((k_ntt3<MODE, 0, LR>), grid, block, 0, ctx->stream, a);
}
static int g_ntt_skip = getenv("VX_NTT_SKIP") ? atoi(getenv("VX_NTT_SKIP")) : 0;  // timing experiments (wrong results!)
// VX_NTT_GROUP=G: run ALL passes of a transform over G columns before moving to the next G (a group of 32 columns of 2^19 is
// 128 MB: the second pass would find in the 256 MB Infinity Cache what the first just wrote).  0 = every pass over all columns.
static size_t g_ntt_group = getenv("VX_NTT_GROUP") ? (size_t)atoi(getenv("VX_NTT_GROUP")) : 0;
template <int MODE>
static int32_t launch_pass(vx_ctx* ctx, PassArgs& a, int log_n, size_t n_cols, int inverse) {
    size_t tiles = (size_t)1 << (log_n - a.log_rows - a.log_T);
    a.dbg_skip = g_ntt_skip;
    const bool expand_ok = a.expand_bits == 0 || (MODE == 1 && !inverse && a.log_rows == 12 && a.expand_bits <= 3);
    if (a.log_rows + a.log_T == 12 && !g_ntt_v1 && !g_ntt_v2 && (a.log_rows == 12 || (a.log_rows >= 4 && a.log_rows <= 8)) && expand_ok) {
        a.n_tiles = tiles;
        const unsigned gx = (unsigned)((tiles + VX_NTT3_TPB - 1) / VX_NTT3_TPB);
        switch (a.log_rows) {
        case 4: launch_ntt3<MODE, 4>(ctx, a, gx, n_cols, inverse); break;
        case 5: launch_ntt3<MODE, 5>(ctx, a, gx, n_cols, inverse); break;
        case 6: launch_ntt3<MODE, 6>(ctx, a, gx, n_cols, inverse); break;
        case 7: launch_ntt3<MODE, 7>(ctx, a, gx, n_cols, inverse); break;
        case 8: launch_ntt3<MODE, 8>(ctx, a, gx, n_cols, inverse); break;
        default: launch_ntt3<MODE, 12>(ctx, a, gx, n_cols, inverse); break;
        }
        VX_HIP(hipGetLastError());
        return VX_OK;
    }
    if (a.log_rows + a.log_T == 12 && !g_ntt_v1) {
        a.n_tiles = tiles;
        const unsigned gx = (unsigned)((tiles + NTT_TPB - 1) / NTT_TPB);
        if (inverse) hipLaunchKernelGGL((k_ntt_tile<MODE, 1>), dim3(gx, (unsigned)n_cols), dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL((k_ntt_tile<MODE, 0>), dim3(gx, (unsigned)n_cols), dim3(256), 0, ctx->stream, a);
        VX_HIP(hipGetLastError());
        return VX_OK;
    }
    hipLaunchKernelGGL(k_ntt_pass<MODE>, dim3((unsigned)tiles, (unsigned)n_cols), dim3(256), lds_bytes(a.log_rows, a.log_T),
                       ctx->stream, a);
    VX_HIP(hipGetLastError());
    return VX_OK;
}

// natural -> bit-reversed positions.  src may equal dst.
static int32_t ntt_dif(vx_ctx* ctx, const uint64_t* src, size_t src_stride, uint64_t* dst, size_t dst_stride, int L,
                       size_t n_cols, int inverse, uint64_t scale) {
    if (g_ntt_group && n_cols > g_ntt_group && L > 12) {
        for (size_t c0 = 0; c0 < n_cols; c0 += g_ntt_group)
            VX_TRY(ntt_dif(ctx, src + c0 * src_stride, src_stride, dst + c0 * dst_stride, dst_stride, L, n_cols - c0 < g_ntt_group ? n_cols - c0 : g_ntt_group, inverse, scale));
        return VX_OK;
    }
    std::vector<int> outer;
    int last;
    plan_chunks(L, outer, last);
    PassArgs a{};
    a.src = src;
    a.src_col_stride = src_stride;
    a.dst = dst;
    a.dst_col_stride = dst_stride;
    a.w12 = inverse ? ctx->w12_inv : ctx->w12_fwd;
    a.tw = inverse ? ctx->tw_inv.d : ctx->tw_fwd.d;
    a.scale = 0;
    int ls = L;
    for (int c : outer) {
        a.log_sub = ls;
        a.log_rows = c;
        a.log_T = (ls - c) < (12 - c) ? (ls - c) : (12 - c);
        {
            Tw2 t2;
            VX_TRY(vx_get_tw2(ctx, ls, inverse, &t2));
            a.tw2_lo = t2.lo, a.tw2_hi = t2.hi, a.tw2_bits = t2.lo_bits, a.tw2_total = ((unsigned)1 << t2.lo_bits) + ((unsigned)1 << (ls - t2.lo_bits));
        }
        VX_TRY(launch_pass<0>(ctx, a, L, n_cols, inverse));
        a.src = dst;
        a.src_col_stride = dst_stride;
        ls -= c;
    }
    a.log_sub = a.log_rows = last;
    a.log_T = 0;
    a.scale = scale;
    return launch_pass<0>(ctx, a, L, n_cols, inverse);
}

// bit-reversed positions -> natural.  With expand_bits = r the source holds 2^(L-r)
// coefficients (bit-reversed positions) that are zero-padded to 2^L on the fly.
static int32_t ntt_dit(vx_ctx* ctx, const uint64_t* src, size_t src_stride, uint64_t* dst, size_t dst_stride, int L,
                       size_t n_cols, int inverse, int expand_bits, const uint64_t* shift_tab) {
    if (g_ntt_group && n_cols > g_ntt_group && L > 12) {
        for (size_t c0 = 0; c0 < n_cols; c0 += g_ntt_group)
            VX_TRY(ntt_dit(ctx, src + c0 * src_stride, src_stride, dst + c0 * dst_stride, dst_stride, L, n_cols - c0 < g_ntt_group ? n_cols - c0 : g_ntt_group, inverse, expand_bits, shift_tab));
        return VX_OK;
    }
    std::vector<int> outer;
    int last;
    plan_chunks(L, outer, last);
    PassArgs a{};
    a.src = src;
    a.src_col_stride = src_stride;
    a.dst = dst;
    a.dst_col_stride = dst_stride;
    a.w12 = inverse ? ctx->w12_inv : ctx->w12_fwd;
    a.tw = inverse ? ctx->tw_inv.d : ctx->tw_fwd.d;
    a.log_sub = a.log_rows = last;
    a.log_T = 0;
    a.expand_bits = expand_bits;
    a.log_coeff = L - expand_bits;
    a.shift_tab = shift_tab;
    VX_TRY(launch_pass<1>(ctx, a, L, n_cols, inverse));
    a.src = dst;
    a.src_col_stride = dst_stride;
    a.expand_bits = 0;
    a.shift_tab = nullptr;
    int ls = last;
    for (int i = (int)outer.size() - 1; i >= 0; --i) {
        int c = outer[i];
        ls += c;
        a.log_sub = ls;
        a.log_rows = c;
        a.log_T = (ls - c) < (12 - c) ? (ls - c) : (12 - c);
        {
            Tw2 t2;
            VX_TRY(vx_get_tw2(ctx, ls, inverse, &t2));
            a.tw2_lo = t2.lo, a.tw2_hi = t2.hi, a.tw2_bits = t2.lo_bits, a.tw2_total = ((unsigned)1 << t2.lo_bits) + ((unsigned)1 << (ls - t2.lo_bits));
        }
        VX_TRY(launch_pass<1>(ctx, a, L, n_cols, inverse));
    }
    return VX_OK;
}

static int32_t bitrev_cols(vx_ctx* ctx, const uint64_t* in, size_t in_stride, uint64_t* out, size_t out_stride, int L,
                           size_t n_cols) {
    size_t n = (size_t)1 << L;
    unsigned gx = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    hipLaunchKernelGGL(k_bitrev_copy, dim3(gx, (unsigned)n_cols), dim3(256), 0, ctx->stream, in, out, L, in_stride,
                       out_stride);
    VX_HIP(hipGetLastError());
    return VX_OK;
}

int32_t vx_ntt_dev(vx_ctx* ctx, uint64_t* d, int log_n, size_t n_cols, size_t col_stride, int inverse, uint64_t shift, int order);

// values/coeffs (column-major, stride n) -> coset evaluations (column-major, stride N, natural order)
int32_t vx_lde_dev(vx_ctx* ctx, const uint64_t* src, int log_n, size_t n_cols, int rate_bits, uint64_t shift, int src_kind,
                   uint64_t* dst, uint64_t* coeffs_out) {
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    PowTab st{nullptr};
    if (shift > 1) VX_TRY(vx_get_shift_tab(ctx, shift, &st));
    uint64_t* co;  // coefficients in bit-reversed positions
    VX_TRY(vx_scratch(ctx, n * n_cols, &co));
    if (src_kind == VX_LDE_SRC_VALUES) {
        uint64_t ninv = glh::inv((uint64_t)n % glh::P);
        if (log_n == 0) VX_HIP(hipMemcpyAsync(co, src, n_cols * 8, hipMemcpyDeviceToDevice, ctx->stream));
        else VX_TRY(ntt_dif(ctx, src, n, co, n, log_n, n_cols, 1, ninv));
        if (coeffs_out) VX_TRY(bitrev_cols(ctx, co, n, coeffs_out, n, log_n, n_cols));
    } else {
        VX_TRY(bitrev_cols(ctx, src, n, co, n, log_n, n_cols));
        if (coeffs_out) VX_HIP(hipMemcpyAsync(coeffs_out, src, n * n_cols * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (log_n + rate_bits == 0) {
        VX_HIP(hipMemcpyAsync(dst, co, n_cols * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return VX_OK;
    }
    return ntt_dit(ctx, co, n, dst, N, log_n + rate_bits, n_cols, 0, rate_bits, st.d);
}

// Same as vx_lde_dev(VX_LDE_SRC_VALUES) but the inverse transform runs IN PLACE on `values`
// (which is left holding the coefficients in bit-reversed positions): no n*c scratch copy, so a
// trace of t bytes needs t + 2^r t instead of 2t + 2^r t -- what lets the MAX_HEADER_SIZE case fit in HBM.
int32_t vx_lde_consume_dev(vx_ctx* ctx, uint64_t* values, int log_n, size_t n_cols, int rate_bits, uint64_t shift, uint64_t* dst) {
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    PowTab st{nullptr};
    if (shift > 1) VX_TRY(vx_get_shift_tab(ctx, shift, &st));
    if (log_n > 0) VX_TRY(ntt_dif(ctx, values, n, values, n, log_n, n_cols, 1, glh::inv((uint64_t)n % glh::P)));
    if (log_n + rate_bits == 0) {
        VX_HIP(hipMemcpyAsync(dst, values, n_cols * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return VX_OK;
    }
    return ntt_dit(ctx, values, n, dst, N, log_n + rate_bits, n_cols, 0, rate_bits, st.d);
}

// Values (kept intact) -> coefficients in bit-reversed positions (coef_brev, n*c) -> coset evaluations (dst, N*c).
// What an AIR with an auxiliary round needs: the trace values are read again after the challenges are known, and the
// coefficient buffer serves the openings at zeta.  The inverse transform's first pass simply writes elsewhere.
int32_t vx_lde_keep_dev(vx_ctx* ctx, const uint64_t* values, int log_n, size_t n_cols, int rate_bits, uint64_t shift, uint64_t* coef_brev,
                        uint64_t* dst) {
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    PowTab st{nullptr};
    if (shift > 1) VX_TRY(vx_get_shift_tab(ctx, shift, &st));
    if (log_n == 0) VX_HIP(hipMemcpyAsync(coef_brev, values, n_cols * 8, hipMemcpyDeviceToDevice, ctx->stream));
    else VX_TRY(ntt_dif(ctx, values, n, coef_brev, n, log_n, n_cols, 1, glh::inv((uint64_t)n % glh::P)));
    if (log_n + rate_bits == 0) {
        VX_HIP(hipMemcpyAsync(dst, coef_brev, n_cols * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return VX_OK;
    }
    return ntt_dit(ctx, coef_brev, n, dst, N, log_n + rate_bits, n_cols, 0, rate_bits, st.d);
}

int32_t vx_gather_rows_dev(vx_ctx* ctx, const uint64_t* lde, int log_N, size_t n_cols, const uint64_t* leaf_idx, size_t n_idx,
                           uint64_t* out) {
    if (n_idx == 0) return VX_OK;
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, n_idx + n_idx * n_cols, &sc));
    VX_HIP(hipMemcpyAsync(sc, leaf_idx, n_idx * 8, hipMemcpyHostToDevice, ctx->stream));
    size_t tot = n_idx * n_cols;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, lde, log_N, n_cols,
                       (const uint64_t*)sc, n_idx, sc + n_idx);
    VX_HIP(hipGetLastError());
    VX_HIP(hipMemcpyAsync(out, sc + n_idx, tot * 8, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}

extern "C" {

int32_t vx_ntt(vx_ctx* ctx, vx_buf* buf, size_t off, int log_n, size_t n_cols, size_t col_stride, int inverse,
               uint64_t shift, int order) {
    if (!ctx || !buf) return VX_ERR_ARG;
    VX_CHECK(log_n >= 0 && log_n <= 28, "vx_ntt: log_n %d out of range [0,28]", log_n);
    VX_CHECK(col_stride >= ((size_t)1 << log_n) && n_cols >= 1 && off + (n_cols - 1) * col_stride + ((size_t)1 << log_n) <= buf->n, "vx_ntt: columns exceed the buffer");
    return vx_ntt_dev(ctx, buf->d + off, log_n, n_cols, col_stride, inverse, shift, order);
}
}  // extern "C"

int32_t vx_ntt_dev(vx_ctx* ctx, uint64_t* d, int log_n, size_t n_cols, size_t col_stride, int inverse, uint64_t shift, int order) {
    if (!ctx || !d) return VX_ERR_ARG;
    VX_CHECK(log_n >= 0 && log_n <= 28, "vx_ntt: log_n %d out of range [0,28]", log_n);
    size_t n = (size_t)1 << log_n;
    VX_CHECK(n_cols >= 1 && n_cols <= 65535, "vx_ntt: n_cols %zu out of range", n_cols);
    VX_CHECK(shift < GL_P, "vx_ntt: shift not canonical");
    VX_CHECK(order == VX_ORDER_NATURAL || order == VX_ORDER_BITREV, "vx_ntt: bad order");
    if (log_n == 0) return VX_OK;
    const bool coset = shift > 1;
    const unsigned gx = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    PowTab st{nullptr};
    if (!inverse) {
        if (coset) VX_TRY(vx_get_shift_tab(ctx, shift, &st));
        if (order == VX_ORDER_BITREV) {
            if (coset) {
                hipLaunchKernelGGL(k_scale_pow, dim3(gx, (unsigned)n_cols), dim3(256), 0, ctx->stream, d, log_n, col_stride,
                                   (const uint64_t*)st.d, (uint64_t)1);
                VX_HIP(hipGetLastError());
            }
            return ntt_dif(ctx, d, col_stride, d, col_stride, log_n, n_cols, 0, 0);
        }
        // natural in -> natural out: permute into scratch (bit-reversed), DIT back into place
        uint64_t* sc;
        VX_TRY(vx_scratch(ctx, n * n_cols, &sc));
        VX_TRY(bitrev_cols(ctx, d, col_stride, sc, n, log_n, n_cols));
        return ntt_dit(ctx, sc, n, d, col_stride, log_n, n_cols, 0, 0, st.d);
    }
    if (coset) VX_TRY(vx_get_shift_tab(ctx, glh::inv(shift), &st));
    const uint64_t ninv = glh::inv((uint64_t)n % glh::P);
    if (order == VX_ORDER_BITREV) {
        // values in bit-reversed positions -> natural coefficients
        VX_TRY(ntt_dit(ctx, d, col_stride, d, col_stride, log_n, n_cols, 1, 0, nullptr));
        hipLaunchKernelGGL(k_scale_pow, dim3(gx, (unsigned)n_cols), dim3(256), 0, ctx->stream, d, log_n, col_stride,
                           (const uint64_t*)st.d, ninv);
        VX_HIP(hipGetLastError());
        return VX_OK;
    }
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, n * n_cols, &sc));
    VX_TRY(ntt_dif(ctx, d, col_stride, sc, n, log_n, n_cols, 1, ninv));
    VX_TRY(bitrev_cols(ctx, sc, n, d, col_stride, log_n, n_cols));
    if (coset) {
        hipLaunchKernelGGL(k_scale_pow, dim3(gx, (unsigned)n_cols), dim3(256), 0, ctx->stream, d, log_n, col_stride,
                           (const uint64_t*)st.d, (uint64_t)1);
        VX_HIP(hipGetLastError());
    }
    return VX_OK;
}

extern "C" {

int32_t vx_lde(vx_ctx* ctx, const vx_buf* src, int log_n, size_t n_cols, int rate_bits, uint64_t shift, int src_kind,
               vx_buf* dst, vx_buf* coeffs_out) {
    if (!ctx || !src || !dst) return VX_ERR_ARG;
    VX_CHECK(log_n >= 0 && rate_bits >= 0 && log_n + rate_bits <= 28, "vx_lde: log_n %d + rate_bits %d out of range", log_n, rate_bits);
    VX_CHECK(n_cols >= 1 && n_cols <= 65535, "vx_lde: n_cols out of range");
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    VX_CHECK(src->n >= n * n_cols, "vx_lde: src holds %zu < %zu elements", src->n, n * n_cols);
    VX_CHECK(dst->n >= N * n_cols, "vx_lde: dst holds %zu < %zu elements", dst->n, N * n_cols);
    VX_CHECK(!coeffs_out || coeffs_out->n >= n * n_cols, "vx_lde: coeffs_out too small");
    VX_CHECK(shift >= 1 && shift < GL_P, "vx_lde: bad shift");
    VX_CHECK(src->d != dst->d, "vx_lde: src and dst alias");
    return vx_lde_dev(ctx, src->d, log_n, n_cols, rate_bits, shift, src_kind, dst->d, coeffs_out ? coeffs_out->d : nullptr);
}

int32_t vx_lde_rows(vx_ctx* ctx, const vx_buf* lde, int log_N, size_t n_cols, const uint64_t* leaf_idx, size_t n_idx,
                    uint64_t* out) {
    if (!ctx || !lde || !leaf_idx || !out) return VX_ERR_ARG;
    size_t N = (size_t)1 << log_N;
    VX_CHECK(lde->n >= N * n_cols, "vx_lde_rows: lde too small");
    for (size_t i = 0; i < n_idx; ++i) VX_CHECK(leaf_idx[i] < N, "vx_lde_rows: leaf index %llu >= %zu", (unsigned long long)leaf_idx[i], N);
    return vx_gather_rows_dev(ctx, lde->d, log_N, n_cols, leaf_idx, n_idx, out);
}
}  // extern "C"
