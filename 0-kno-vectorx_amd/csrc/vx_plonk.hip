// K9 of SURVEY.md section 2.2 / 8(b): the partial products of Plonk's permutation argument as plonky2 computes them
// (plonky2 v0.2.0, Cargo.lock:4848-4869 -- not vendored; plonk/prover.rs wires_permutation_partial_products_and_zs +
// plonk/plonk_common.rs / util/partial_products.rs, reached from every Circuit::prove of the reference, circuits/header_range.rs:167).
// For n = 2^log_n rows x_i = g^i, R routed wires w_j with coset shifts k_j and sigma polynomials s_j, and one challenge pair:
//     q_j(i)     = (w_j(i) + beta k_j x_i + gamma) / (w_j(i) + beta s_j(i) + gamma)
//     chunk_c(i) = prod of q_j(i) over the c-th group of `chunk` consecutive wires            (m = ceil(R / chunk) groups)
//     Z(x_0) = 1,   pp_t(i) = Z(x_i) chunk_0(i) ... chunk_t(i)  for t < m - 1,   Z(x_(i+1)) = Z(x_i) chunk_0(i) ... chunk_(m-1)(i)
// Output columns: Z, pp_0 .. pp_(m-2).  This library's own provers are STARK-only and never call it: it is the primitive a
// plonky2 patched through the C ABI would forward to, HBM-bound: 8 n (2 R + m) algorithmic bytes.
//
// One lane per row.  A row's m chunk quotients need m inversions: one (Montgomery's trick over the m denominators'
// products).  The column Z is an exclusive prefix PRODUCT over the rows (tile products, a scan of those, apply).
#include "gl.cuh"
#include "vx_internal.h"

namespace {
constexpr int PP_MAX_CHUNKS = 32, PP_TILE = 4096, PP_PER_LANE = PP_TILE / 256;

// rows: out[0][i] = prod of all chunks (turned into Z by the scan), out[t + 1][i] = chunk_0 .. chunk_t
__global__ __launch_bounds__(256) void k_pp_rows(const uint64_t* wires, const uint64_t* sigmas, size_t n, int log_n, size_t n_routed, const uint64_t* k_is, uint64_t beta,
                                                 uint64_t gamma, int chunk, int m, uint64_t g, uint64_t* out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t bx = gl_mul(beta, gl_pow(g, i));  // beta x_i
    uint64_t num[PP_MAX_CHUNKS], den[PP_MAX_CHUNKS];
#pragma unroll 1
    for (int c = 0; c < m; ++c) {
        uint64_t a = 1, b = 1;
        const size_t j1 = (size_t)(c + 1) * chunk < n_routed ? (size_t)(c + 1) * chunk : n_routed;
        for (size_t j = (size_t)c * chunk; j < j1; ++j) {
            const uint64_t w = wires[j * n + i];
            a = gl_mul(a, gl_add(gl_add(w, gl_mul(bx, k_is[j])), gamma));
            b = gl_mul(b, gl_add(gl_add(w, gl_mul(beta, sigmas[j * n + i])), gamma));
        }
        num[c] = a, den[c] = b;
    }
    // 1 / den_c for all c with one inversion: prefix products, invert the total, walk back
    uint64_t pre[PP_MAX_CHUNKS];
    uint64_t acc = 1;
#pragma unroll 1
    for (int c = 0; c < m; ++c) pre[c] = acc, acc = gl_mul(acc, den[c]);
    uint64_t inv = gl_inv(acc);  // (a zero denominator -- probability ~R n / p over the challenges -- gives 0 here, as a^(p-2) does)
    uint64_t run = 1;
#pragma unroll 1
    for (int c = m - 1; c >= 0; --c) {
        const uint64_t dinv = gl_mul(inv, pre[c]);
        inv = gl_mul(inv, den[c]);
        num[c] = gl_mul(num[c], dinv);  // chunk_c
    }
#pragma unroll 1
    for (int c = 0; c < m; ++c) {
        run = gl_mul(run, num[c]);
        out[(size_t)(c + 1 < m ? c + 1 : 0) * n + i] = run;
    }
}

__device__ __forceinline__ uint64_t block_exclusive_prod(uint64_t v, uint64_t* lds, uint64_t* total) {
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint64_t x = t >= d ? lds[t - d] : 1;
        __syncthreads();
        if (t >= d) lds[t] = gl_mul(lds[t], x);
        __syncthreads();
    }
    const uint64_t ex = t ? lds[t - 1] : 1;
    if (total) *total = lds[255];
    __syncthreads();
    return ex;
}
__global__ __launch_bounds__(256) void k_pp_tile_prods(const uint64_t* col, size_t n, uint64_t* prods) {
    __shared__ uint64_t lds[256];
    const size_t t0 = (size_t)blockIdx.x * PP_TILE;
    uint64_t s = 1;
    for (int k = 0; k < PP_PER_LANE; ++k) {
        const size_t i = t0 + (size_t)threadIdx.x * PP_PER_LANE + k;
        if (i < n) s = gl_mul(s, col[i]);
    }
    uint64_t total;
    block_exclusive_prod(s, lds, &total);
    if (threadIdx.x == 0) prods[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void k_pp_scan_prods(uint64_t* prods, size_t tiles) {  // one block
    __shared__ uint64_t lds[256];
    uint64_t carry = 1;
    for (size_t base = 0; base < tiles; base += 256) {
        const size_t i = base + threadIdx.x;
        const uint64_t v = i < tiles ? prods[i] : 1;
        uint64_t total;
        const uint64_t ex = block_exclusive_prod(v, lds, &total);
        if (i < tiles) prods[i] = gl_mul(ex, carry);
        carry = gl_mul(carry, total);
    }
}
// column 0: row products -> Z (exclusive prefix products); columns 1 .. m-1: times Z of their row
__global__ __launch_bounds__(256) void k_pp_apply(uint64_t* out, size_t n, int m, const uint64_t* prods) {
    __shared__ uint64_t lds[256];
    const size_t t0 = (size_t)blockIdx.x * PP_TILE;
    uint64_t v[PP_PER_LANE], s = 1;
    for (int k = 0; k < PP_PER_LANE; ++k) {
        const size_t i = t0 + (size_t)threadIdx.x * PP_PER_LANE + k;
        v[k] = i < n ? out[i] : 1;
        s = gl_mul(s, v[k]);
    }
    uint64_t run = gl_mul(block_exclusive_prod(s, lds, nullptr), prods[blockIdx.x]);
    for (int k = 0; k < PP_PER_LANE; ++k) {
        const size_t i = t0 + (size_t)threadIdx.x * PP_PER_LANE + k;
        if (i < n) {
            out[i] = run;
            for (int c = 1; c < m; ++c) out[(size_t)c * n + i] = gl_mul(out[(size_t)c * n + i], run);
        }
        run = gl_mul(run, v[k]);
    }
}
}  // namespace

extern "C" int32_t vx_partial_products(vx_ctx* ctx, const vx_buf* wires, const vx_buf* sigmas, int log_n, size_t n_routed, const uint64_t* k_is, uint64_t beta,
                                       uint64_t gamma, size_t chunk, vx_buf* out) {
    if (!ctx || !wires || !sigmas || !k_is || !out) return VX_ERR_ARG;
    VX_CHECK(log_n >= 0 && log_n <= 28, "partial products: log_n %d out of range [0,28]", log_n);
    VX_CHECK(n_routed >= 1 && n_routed <= 4096 && chunk >= 1, "partial products: %zu routed wires in chunks of %zu", n_routed, chunk);
    const size_t n = (size_t)1 << log_n, m = (n_routed + chunk - 1) / chunk;
    VX_CHECK(m <= (size_t)PP_MAX_CHUNKS, "partial products: %zu chunks (at most %d)", m, PP_MAX_CHUNKS);
    VX_CHECK(wires->n >= n_routed * n && sigmas->n >= n_routed * n, "partial products: wires / sigmas hold fewer than %zu x %zu values", n_routed, n);
    VX_CHECK(out->n >= m * n, "partial products: out holds %zu < %zu values", out->n, m * n);
    VX_CHECK(beta < GL_P && gamma < GL_P, "partial products: challenges not canonical");
    for (size_t j = 0; j < n_routed; ++j) VX_CHECK(k_is[j] < GL_P, "partial products: coset shift %zu not canonical", j);
    const size_t tiles = (n + PP_TILE - 1) / PP_TILE;
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, n_routed + tiles, &sc));
    VX_HIP(hipMemcpyAsync(sc, k_is, n_routed * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_pp_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint64_t*)wires->d, (const uint64_t*)sigmas->d, n, log_n, n_routed,
                       (const uint64_t*)sc, beta, gamma, (int)chunk, (int)m, glh::root(log_n), out->d);
    hipLaunchKernelGGL(k_pp_tile_prods, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, (const uint64_t*)out->d, n, sc + n_routed);
    hipLaunchKernelGGL(k_pp_scan_prods, dim3(1), dim3(256), 0, ctx->stream, sc + n_routed, tiles);
    hipLaunchKernelGGL(k_pp_apply, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, out->d, n, (int)m, (const uint64_t*)(sc + n_routed));
    VX_HIP(hipGetLastError());
    return VX_OK;
}
