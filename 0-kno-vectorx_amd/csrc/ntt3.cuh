// k_ntt3<MODE, INV, LR>: the tile pass of vx_ntt.hip with the tile SHAPE fixed at compile time.
//
// A pass works on 4096-element tiles of 2^LR rows x 2^(12-LR) contiguous columns (rows strided by m = 2^log_m elements),
// 16 elements per lane of a 256-thread block, radix-16 rounds in registers, padded LDS exchanges in between -- as
// k_ntt_tile (the run-time-shape kernel it replaces for LR in {4..8, 12}).  What is new:
//   * every address is  (uniform scalar base, per element) + (per-lane offset, computed ONCE per block): the element index
//     of register e in layout F is  lane_part(tid) | e << F  with disjoint bit fields, so row / column / padded LDS slot /
//     bit-reversed row are all sums of a lane term and a compile-time element term.  The run-time-shape kernel spent
//     ~8 VALU instructions of 64-bit index arithmetic per element and access (40 % of its instruction stream);
//   * the round twiddle index of register e is  R(tid) * bitrev4(e)  with R fixed per lane;
//   * the table of w_(2^LR) has 2^(LR-1) entries (64 for the 7-stage pass), which leaves LDS room to stage the two-level
//     table of the between-pass twiddle (w_S^(c * bitrev(r)), S = 2^log_sub) instead of gathering it from L2;
//   * round arithmetic stays in [0, 2^64) + a carry word between reductions (gl96 below).
// Layout names: F = 8 / 4 / 0 is the bit position of the 4-bit register window inside the 12-bit tile index.
#pragma once
#include <utility>

#include "gl.cuh"

namespace n3 {

template <int F>
__device__ __forceinline__ int lane_part(int tid) {
    return (tid & ((1 << F) - 1)) | ((tid >> F) << (F + 4));
}
__host__ __device__ constexpr int pad(int i) { return i + (i >> 4); }
__host__ __device__ constexpr int brev_c(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

template <int LR>
struct Shape {
    static_assert(LR >= 4 && LR <= 12, "k_ntt3 needs 4 <= log_rows <= 12");
    static constexpr int LT = 12 - LR, T = 1 << LT;
    static constexpr int NR = (LR + 3) / 4;
    static constexpr int QA = 4, QB = LR > 4 ? (LR - 4 < 4 ? LR - 4 : 4) : 0, QC = LR > 8 ? LR - 8 : 0;
    static constexpr int WN = LR >= 5 ? 1 << (LR - 1) : 0;  // entries of the staged w_(2^LR) half table
    // layout the DIF pass ends in / the DIT pass starts from when there is no closing exchange
    static constexpr int F_EDGE = NR == 1 ? 8 : (NR == 2 ? 4 : 8);
};

// ---- global addressing: element e of layout F sits at  base + e_off<F>(e) + lane_off<F>(tid)
template <int F, int LT>
__device__ __forceinline__ uint32_t lane_off(int tid, int log_m) {  // in BYTES (a column is below 4 GB: log_n <= 28)
    const int lp = lane_part<F>(tid);
    return (((uint32_t)(lp >> LT) << log_m) + (uint32_t)(lp & ((1 << LT) - 1))) << 3;
}
__device__ __forceinline__ uint64_t ld_at(const uint64_t* p, uint32_t byte_off) { return *(const uint64_t*)((const char*)p + byte_off); }
__device__ __forceinline__ void st_at(uint64_t* p, uint32_t byte_off, uint64_t v) { *(uint64_t*)((char*)p + byte_off) = v; }
// strided DIF passes (they end in the between-pass twiddle) take VX_NTT3_NC columns per block; the DIT side would have to keep the
// 16 products alive through a column's rounds (its kernels sit at the register budget already)
__host__ __device__ constexpr int cols_per_block(int mode, int lr);
template <int F, int LT>
__device__ __forceinline__ size_t e_off(int e, int log_m) {  // uniform
    const int ep = e << F;
    return ((size_t)(ep >> LT) << log_m) + (size_t)(ep & ((1 << LT) - 1));
}

}  // namespace n3

// ---------------------------------------------------------------------------------------------------------------
// The kernel.  Included by vx_ntt.hip after PassArgs and tab3_pow.
#include "gl96.h"

#ifndef VX_NTT3_TPB
#define VX_NTT3_TPB 4
#endif
// waves per SIMD the register allocation aims at.  3 everywhere: the contiguous 12-stage pass would fit 4 blocks per CU
// (37 KB of LDS each), but at 128 VGPRs it spills 7 registers to scratch -- rocprofv3 --pmc counted 0.98 GB of extra HBM
// traffic per launch (8.59 -> 9.57 GB) for no gain in time (4.04 vs 4.09 ms per transform, tools/ntt_ab.py)
#ifndef VX_NTT3_WAVES
#define VX_NTT3_WAVES(LR, MODE) 3
#endif
// columns per block of a strided DIF pass: the between-pass twiddle w = hi * lo of a (tile, lane, element) is the same for
// every column, so the block walks VX_NTT3_NC columns per tile and computes it once (16 products kept in registers)
#ifndef VX_NTT3_NC
#define VX_NTT3_NC 2
#endif
namespace n3 {
__host__ __device__ constexpr int cols_per_block(int mode, int lr) { return (mode == 0 && lr <= 8) ? VX_NTT3_NC : 1; }
}  // namespace n3
#ifndef VX_NTT3_DIRECT0
#define VX_NTT3_DIRECT0 0  // 1: a three-round DIT pass without zero padding also loads straight into layout 0 (A/B aid)
#endif

namespace n3 {

template <int F>
__device__ __forceinline__ void exch_write(const uint64_t* y, uint64_t* lds, int slot) {  // slot = pad(lane_part<F>(tid))
#pragma unroll
    for (int e = 0; e < 16; ++e) lds[slot + pad(e << F)] = y[e];
}
template <int F>
__device__ __forceinline__ void exch_read(uint64_t* y, const uint64_t* lds, int slot) {
#pragma unroll
    for (int e = 0; e < 16; ++e) y[e] = lds[slot + pad(e << F)];
}
__device__ __forceinline__ void widen(gl96::X* x, const uint64_t* y) {
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = gl96::from64(y[e]);
}
// 16 lazy values -> 64-bit representatives in [0, 2^64).  The fast form (two 32-bit additions per value) is exact unless a
// half of some value sits within FOLD_K of a wrap point; the whole wave then takes the exact form (test vectors of all
// p - 1 / 2^32 - 1 words do that on purpose).
__device__ __forceinline__ void fold16(uint64_t* y, const gl96::X* x) {
    uint32_t m = 0xFFFFFFFFu;
#pragma unroll
    for (int e = 0; e < 16; ++e) m = gl96::fold_margin(m, x[e]);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!gl96::fold_ok(m)) != 0, 0)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) y[e] = gl96::fold_exact(x[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) y[e] = gl96::fold_fast(x[e]);
    }
}
// representatives in [0, 2^64) -> canonical: a value >= p has an all-ones high word (probability 2^-32 on random data)
__device__ __forceinline__ void canon16(uint64_t* y) {
    uint32_t mx = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const uint32_t h = (uint32_t)(y[e] >> 32);
        mx = mx > h ? mx : h;
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(mx == 0xFFFFFFFFu) != 0, 0)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) y[e] = gl_canon(y[e]);
    }
}
// y[e] *= w[e - 1]  (e >= 1); any representatives in, [0, 2^64) out
__device__ __forceinline__ void mul15(uint64_t* y, const uint64_t* w) {
#pragma unroll
    for (int e = 1; e < 16; ++e) y[e] = gl_mul_nc(y[e], w[e - 1]);
}

}  // namespace n3

// EB > 0: the first pass of an LDE (MODE 1, LR 12, contiguous tile).  The source holds the 2^log_coeff coefficients in
// bit-reversed positions; position g of the 2^EB times larger transform is coefficient g >> EB when g is a multiple of
// 2^EB and zero otherwise, scaled by shift^bitrev(g >> EB).  The tile is loaded straight into layout 0 (16 consecutive
// positions per lane): a lane reads its 16 >> EB coefficients as one contiguous run, the zeros are compile-time constants
// (the first EB butterfly stages fold away), and the scale factor is split along the bit fields of the index,
//   shift^bitrev(q) = A(tile) * B(lane) * C(element),
// so that a coefficient costs two multiplications instead of three table look-ups and up to three multiplications on
// every lane of the tile, zero or not.
template <int MODE, int INV, int LR, int EB = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VX_NTT3_WAVES(LR, MODE), 4))) void k_ntt3(PassArgs a) {
    static_assert(EB == 0 || (MODE == 1 && LR == 12 && EB <= 3), "zero-padded loads exist for the contiguous DIT pass only");
    using S = n3::Shape<LR>;
    constexpr int LT = S::LT, NR = S::NR;
    constexpr int T2N = LR <= 8 ? 2048 : 0;            // staged two-level table of the between-pass twiddle
    constexpr int WBN = LR > 8 ? 256 : 0;               // staged w_256^j (all of them: no sign fix-up) for the second round twiddle
    // layout of the pass's outer edge: DIF stores from it, DIT loads into it.  A three-round pass ends (starts) in layout 0, 16
    // consecutive elements per lane: with VX_NTT3_DIRECT0 it stores (loads) them from there -- 128 contiguous bytes per lane as
    // 16-byte accesses -- instead of paying one more LDS exchange for 512-byte wave rows
    constexpr int FE = (NR == 3 && (EB > 0 || (VX_NTT3_DIRECT0 && MODE == 1))) ? 0 : S::F_EDGE;
    // strided passes (LR <= 8) look their first-round twiddles up in a full-circle table of w_(2^LR) (at most 256 entries) instead of
    // keeping 15 of them per lane in registers: the registers go to the 16 between-pass products shared by the columns of a block
    constexpr int WAN = (LR > 4 && LR <= 8) ? (1 << LR) : 0;
    __shared__ __attribute__((aligned(16))) uint64_t lds[4096 + 256 + (WBN ? WBN : 1) + (T2N ? T2N : 1) + (WAN ? WAN : 1)];
    uint64_t* const wsB = lds + 4096 + 256;
    uint64_t* const t2s = wsB + (WBN ? WBN : 1);
    uint64_t* const wsA = t2s + (T2N ? T2N : 1);
    const int tid = threadIdx.x;
    const int log_m = a.log_sub - LR;
    const bool strided = log_m > 0;                     // m > 1: this pass carries the between-pass twiddle
    const bool t2_lds = T2N && strided && a.tw2_total <= (unsigned)T2N;
    if (WBN) wsB[tid] = tid < 128 ? a.w12[tid << 4] : GL_P - a.w12[(tid - 128) << 4];
    if (T2N && t2_lds)
        for (unsigned j = tid; j < a.tw2_total; j += 256) t2s[j] = a.tw2_lo[j];

    // round-A twiddles of this lane, w_(2^LR)^(R bitrev4(e)) with R = rho mod 2^(LR-4) = tid >> LT in layout 8: the same for
    // every tile, so they live in registers (the sign of the upper half circle is folded in here, once)
    if (WAN) {  // w_(2^LR)^j for every j (the upper half circle is the negation): no sign fix-up at the look-up
        const int j = tid & (WAN - 1), e4096 = j << (12 - LR);
        if (tid < WAN) wsA[j] = (e4096 & 2048) ? GL_P - a.w12[e4096 & 2047] : a.w12[e4096 & 2047];
    }
    uint64_t wA[WAN ? 1 : 15];
    const int RA = tid >> LT;
    if (LR > 4 && !WAN) {
#pragma unroll
        for (int e = 1; e < 16; ++e) {
            const int idx = (RA * n3::brev_c(e, 4)) << (12 - LR);   // exponent of w_4096, below 4096
            const uint64_t w = a.w12[idx & 2047];
            wA[WAN ? 0 : e - 1] = (idx & 2048) ? GL_P - w : w;
        }
    }
    if (WBN || T2N || WAN) __syncthreads();
    auto twiddle_a = [&](uint64_t* y) {
        if (WAN) {
#pragma unroll
            for (int e = 1; e < 16; ++e) y[e] = gl_mul_nc(y[e], wsA[(RA * n3::brev_c(e, 4)) & (WAN ? WAN - 1 : 0)]);
        } else n3::mul15(y, wA);
    };

    constexpr int NC = n3::cols_per_block(MODE, LR);
    const size_t col_first = (size_t)blockIdx.y * NC;
    const int ncol_here = (int)(a.n_cols - col_first < (size_t)NC ? a.n_cols - col_first : (size_t)NC);
    const int log_tps = a.log_sub - 12;                 // tiles per sub-array
    const size_t tile0 = (size_t)blockIdx.x * VX_NTT3_TPB;
    const int n_here = (int)(a.n_tiles - tile0 < (size_t)VX_NTT3_TPB ? a.n_tiles - tile0 : (size_t)VX_NTT3_TPB);

    // per-lane terms, fixed for the block
    const int s8 = n3::pad(n3::lane_part<8>(tid)), s4 = n3::pad(n3::lane_part<4>(tid)), s0 = n3::pad(n3::lane_part<0>(tid));
    const uint32_t off8 = n3::lane_off<8, LT>(tid, log_m), offE = n3::lane_off<FE, LT>(tid, log_m);
    const int RB = LR > 8 ? ((tid & 15) >> LT) : 0;     // rho mod 2^(LR-8) in layout 4; exponent of w_256 is RB * bitrev4(e) << (12 - LR)
    auto twiddle_b = [&](uint64_t* y) {
#pragma unroll
        for (int e = 1; e < 16; ++e) y[e] = gl_mul_nc(y[e], wsB[((RB * n3::brev_c(e, 4)) << (LR > 8 ? 12 - LR : 0)) & 255]);
    };
    // between-pass twiddle on the strided side (layout FE): exponent (col0 + tau) * bitrev_LR(rho), both sums of a lane and an element term
    const int lpE = n3::lane_part<FE>(tid);
    const uint32_t tauE = lpE & ((1 << LT) - 1), brE = __brev((unsigned)(lpE >> LT)) >> (32 - LR);
    const uint32_t t2_mask = ((uint32_t)1 << a.tw2_bits) - 1;
    auto ip_twiddle = [&](uint32_t col0, int e) -> uint64_t {
        const int ep = e << FE;
        const uint32_t ex = (col0 + tauE + (uint32_t)(ep & ((1 << LT) - 1))) * (brE + (uint32_t)n3::brev_c(ep >> LT, LR));
        if (T2N && t2_lds) return gl_mul_nc(t2s[(t2_mask + 1) + (ex >> a.tw2_bits)], t2s[ex & t2_mask]);
        return gl_mul_nc(a.tw2_hi[ex >> a.tw2_bits], a.tw2_lo[ex & t2_mask]);
    };

    // coset scale factors of the zero-padded load (EB > 0): per lane and per element, fixed for the launch
    uint64_t sB = 1, sC[EB > 0 ? (16 >> EB) : 1];
    if (EB > 0 && a.shift_tab) {
        sB = tab3_pow(a.shift_tab, (uint64_t)(__brev((unsigned)tid) >> 24) << (a.log_coeff - 12 + EB));
#pragma unroll
        for (int j = 0; j < (16 >> EB); ++j) sC[j] = tab3_pow(a.shift_tab, (uint64_t)n3::brev_c(j, 4 - EB) << (a.log_coeff - 4 + EB));
    }

    gl96::X x[16];
    uint64_t y[16], wip[NC > 1 ? 16 : 1];
#pragma unroll 1
    for (int t = 0; t < n_here; ++t) {
        const size_t tile = tile0 + t;
        const size_t sub = tile >> log_tps;
        const uint32_t col0 = (uint32_t)(tile - (sub << log_tps)) << LT;
        const size_t base = (sub << a.log_sub) + col0;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {  // (unrolled: as a rolled loop the compiler hoists per-element address terms across it and spills)
        if (NC > 1 && cc >= ncol_here) break;
        const uint64_t* src = a.src + (col_first + cc) * a.src_col_stride;
        uint64_t* dst = a.dst + (col_first + cc) * a.dst_col_stride;
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = gl96::from64(n3::ld_at(src + base + n3::e_off<8, LT>(e, log_m), off8));
            gl96::dif_round<4, INV>(x);
            n3::fold16(y, x);
            if (LR > 4) twiddle_a(y);
            if (NR >= 2) {
                __syncthreads();
                n3::exch_write<8>(y, lds, s8);
                __syncthreads();
                n3::exch_read<4>(y, lds, s4);
                n3::widen(x, y);
                gl96::dif_round<S::QB ? S::QB : 1, INV>(x);
                n3::fold16(y, x);
                if (LR > 8) twiddle_b(y);
            }
            if (NR == 3) {
                __syncthreads();
                n3::exch_write<4>(y, lds, s4);
                __syncthreads();
                n3::exch_read<0>(y, lds, s0);
                n3::widen(x, y);
                gl96::dif_round<S::QC ? S::QC : 1, INV>(x);
                n3::fold16(y, x);
                if (FE != 0) {
                    __syncthreads();
                    n3::exch_write<0>(y, lds, s0);
                    __syncthreads();
                    n3::exch_read<8>(y, lds, s8);
                }
            }
            if (strided) {
                if (NC > 1) {
                    if (cc == 0) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) wip[NC > 1 ? e : 0] = ip_twiddle(col0, e);
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) y[e] = gl_mul_nc(y[e], wip[NC > 1 ? e : 0]);
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) y[e] = gl_mul_nc(y[e], ip_twiddle(col0, e));
                }
            }
            if (a.scale > 1) {
#pragma unroll
                for (int e = 0; e < 16; ++e) y[e] = gl_mul_nc(y[e], a.scale);
            }
            n3::canon16(y);
#pragma unroll
            for (int e = 0; e < 16; ++e) n3::st_at(dst + base + n3::e_off<FE, LT>(e, log_m), offE, y[e]);
        } else {
            if (EB > 0) {
                constexpr int J = 16 >> EB;                       // coefficients per lane
                const uint64_t* sp = src + (base >> EB) + (size_t)tid * J;
                uint64_t ab = 0;
                if (a.shift_tab) ab = gl_mul_nc(tab3_pow(a.shift_tab, brev32((uint32_t)(tile), a.log_coeff - 12 + EB)), sB);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (e & ((1 << EB) - 1)) y[e] = 0;
                    else {
                        uint64_t v = sp[e >> EB];
                        if (a.shift_tab) v = gl_mul_nc(gl_mul_nc(v, ab), sC[e >> EB]);
                        y[e] = v;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) y[e] = n3::ld_at(src + base + n3::e_off<FE, LT>(e, log_m), offE);
                if (a.shift_tab && !strided) {  // plain coset transform, first pass: c_k *= shift^k
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const size_t g = base + (size_t)(n3::lane_part<FE>(tid) | (e << FE));
                        y[e] = gl_mul_nc(y[e], tab3_pow(a.shift_tab, brev32((uint32_t)g, a.log_coeff)));
                    }
                }
                if (strided) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) y[e] = gl_mul_nc(y[e], ip_twiddle(col0, e));
                }
            }
            if (NR == 3) {
                if (FE != 0) {
                    __syncthreads();
                    n3::exch_write<8>(y, lds, s8);
                    __syncthreads();
                    n3::exch_read<0>(y, lds, s0);
                }
                n3::widen(x, y);
                gl96::dit_round<S::QC ? S::QC : 1, INV>(x);
                n3::fold16(y, x);
                __syncthreads();
                n3::exch_write<0>(y, lds, s0);
                __syncthreads();
                n3::exch_read<4>(y, lds, s4);
            }
            if (NR >= 2) {
                if (LR > 8) twiddle_b(y);
                n3::widen(x, y);
                gl96::dit_round<S::QB ? S::QB : 1, INV>(x);
                n3::fold16(y, x);
                __syncthreads();
                n3::exch_write<4>(y, lds, s4);
                __syncthreads();
                n3::exch_read<8>(y, lds, s8);
            }
            if (LR > 4) twiddle_a(y);
            n3::widen(x, y);
            gl96::dit_round<4, INV>(x);
            n3::fold16(y, x);
            n3::canon16(y);
#pragma unroll
            for (int e = 0; e < 16; ++e) n3::st_at(dst + base + n3::e_off<8, LT>(e, log_m), off8, y[e]);
        }
      }
    }
}
