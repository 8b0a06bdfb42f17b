// K6: FRI commit-phase folding, layer trees and proof-of-work for gfx950.
// Replaces plonky2::fri::prover::{fri_committed_trees, fri_proof_of_work}
// (v0.2.0).  The CPU prover folds COEFFICIENTS (reduce_with_powers) and re-runs a
// coset FFT per layer; here a layer is folded directly in EVALUATION space:
// an arity-2^a reduction with challenge beta is a chain of a arity-2 folds with
// beta, beta^2, beta^4, ..., each
//     P'(x^2) = (P(x)+P(-x))/2 + beta * (P(x)-P(-x))/(2x),
// all a steps fused in registers (lane i gathers the 2^a values i + k*N/2^a,
// coalesced across lanes).  Field arithmetic is exact, so the values equal the
// coefficient-fold + coset_fft values bit for bit (tests/test_fri.py).
#include "poseidon.cuh"
#include "vx_internal.h"

struct FoldArgs {
    const uint64_t* in;
    uint64_t* out;
    int log_n, arity_bits;
    uint64_t beta_pow[5][2];  // beta^(2^j)
    uint64_t sinv_half[5];    // (shift^(2^j))^-1 / 2
    uint64_t half;            // 1/2
    const uint64_t* tw_inv;   // three-level powers of w_{2^32}^-1
};

__device__ __forceinline__ uint64_t root_pow_inv(const uint64_t* tw, uint64_t e, int log_s) {
    uint64_t E = (e << (32 - log_s)) & 0xFFFFFFFFULL;
    uint64_t r = tw[4096 + (E >> 22)];
    if (log_s > 10) r = gl_mul(r, tw[2048 + ((E >> 11) & 2047)]);
    if (log_s > 21) r = gl_mul(r, tw[E & 2047]);
    return r;
}

template <int A>
__global__ __launch_bounds__(256) void k_fri_fold(FoldArgs a) {
    constexpr int ARITY = 1 << A;
    const size_t M = (size_t)1 << (a.log_n - A);
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= M) return;
    gl2 v[ARITY];
#pragma unroll
    for (int k = 0; k < ARITY; ++k) {
        const uint64_t* p = a.in + 2 * (i + (size_t)k * M);
        v[k] = {p[0], p[1]};
    }
    int cnt = ARITY;
#pragma unroll
    for (int j = 0; j < A; ++j) {
        cnt >>= 1;
        const int log_dom = a.log_n - j;  // current domain size
        gl2 beta{a.beta_pow[j][0], a.beta_pow[j][1]};
#pragma unroll
        for (int k = 0; k < cnt; ++k) {
            // x = shift^(2^j) * w_{2^log_dom}^(i + k*M);  c = 1/(2x)
            uint64_t c = gl_mul(a.sinv_half[j], root_pow_inv(a.tw_inv, i + (size_t)k * M, log_dom));
            gl2 u = v[k], w = v[k + cnt];
            gl2 s = gl2_scale(gl2_add(u, w), a.half);
            gl2 d = gl2_scale(gl2_sub(u, w), c);
            v[k] = gl2_add(s, gl2_mul(beta, d));
        }
    }
    a.out[2 * i] = v[0].a;
    a.out[2 * i + 1] = v[0].b;
}

// Leaf digests of a FRI layer: leaf j = values at natural indices bitrev_N(j*arity + t).
// Lane q (natural position in [0, M)) reads q + bitrev_a(t)*M and writes digest bitrev(q).
template <int A>
__global__ __launch_bounds__(256) void k_fri_leaf_hash(const uint64_t* evals, int log_n, uint64_t* digests) {
    constexpr int ARITY = 1 << A;
    const int log_m = log_n - A;
    const size_t M = (size_t)1 << log_m;
    size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (q >= M) return;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = 0;
    if (ARITY == 1 || ARITY == 2) {
#pragma unroll
        for (int t = 0; t < ARITY; ++t) {
            const uint64_t* p = evals + 2 * (q + (size_t)brev32(t, A) * M);
            s[2 * t] = p[0];
            s[2 * t + 1] = p[1];
        }
    } else {
#pragma unroll
        for (int t0 = 0; t0 < ARITY; t0 += 4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint64_t* p = evals + 2 * (q + (size_t)brev32(t0 + t, A) * M);
                s[2 * t] = p[0];
                s[2 * t + 1] = p[1];
            }
            poseidon_permute(s);
        }
    }
    uint64_t* d = digests + 4 * (size_t)brev32((uint32_t)q, log_m);
    d[0] = s[0];
    d[1] = s[1];
    d[2] = s[2];
    d[3] = s[3];
}
// gather the leaves (arity ext values, flattened) of n_idx leaf indices
__global__ void k_fri_gather_leaves(const uint64_t* evals, int log_n, int arity_bits, const uint64_t* idx, size_t n_idx,
                                    uint64_t* out) {
    const int arity = 1 << arity_bits;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= n_idx * arity) return;
    size_t k = t >> arity_bits;
    int e = (int)(t & (arity - 1));
    const int log_m = log_n - arity_bits;
    size_t q = brev32((uint32_t)idx[k], log_m);
    const uint64_t* p = evals + 2 * (q + ((size_t)brev32(e, arity_bits) << log_m));
    out[2 * t] = p[0];
    out[2 * t + 1] = p[1];
}

// proof-of-work grind: candidates base .. base+count; atomicMin the smallest hit
__global__ __launch_bounds__(256) void k_fri_pow(const uint64_t* state12, int pos, int bits, uint64_t base, uint64_t count,
                                                 unsigned long long* best) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; t < count; t += stride) {
        uint64_t cand = base + t;
        // a smaller hit exists already (the load must not be hoisted out of the loop: it is how late waves stop)
        if (cand >= __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        uint64_t s[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) s[k] = state12[k];
#pragma unroll
        for (int k = 0; k < 12; ++k)
            if (k == pos) s[k] = cand;
        poseidon_permute(s);
        if (bits == 0 || (s[7] >> (64 - bits)) == 0) atomicMin(best, (unsigned long long)cand);
    }
}

extern "C" {

int32_t vx_fri_fold(vx_ctx* ctx, const vx_buf* evals, int log_n, int arity_bits, const uint64_t beta[2], uint64_t shift,
                    vx_buf* out) {
    if (!ctx || !evals || !beta || !out) return VX_ERR_ARG;
    VX_CHECK(arity_bits >= 1 && arity_bits <= 5 && arity_bits <= log_n && log_n <= 30, "fri fold: bad log_n %d / arity_bits %d", log_n, arity_bits);
    size_t N = (size_t)1 << log_n, M = N >> arity_bits;
    VX_CHECK(evals->n >= 2 * N && out->n >= 2 * M, "fri fold: buffers too small");
    VX_CHECK(shift >= 1 && shift < GL_P && beta[0] < GL_P && beta[1] < GL_P, "fri fold: non-canonical input");
    return vx_fri_fold_dev(ctx, evals->d, log_n, arity_bits, beta, shift, out->d);
}
}  // extern "C"

int32_t vx_fri_fold_dev(vx_ctx* ctx, const uint64_t* evals, int log_n, int arity_bits, const uint64_t beta[2], uint64_t shift,
                        uint64_t* out) {
    size_t N = (size_t)1 << log_n, M = N >> arity_bits;
    FoldArgs a{};
    a.in = evals;
    a.out = out;
    a.log_n = log_n;
    a.arity_bits = arity_bits;
    a.half = glh::inv(2);
    a.tw_inv = ctx->tw_inv.d;
    uint64_t b0 = beta[0], b1 = beta[1], s = shift;
    for (int j = 0; j < arity_bits; ++j) {
        a.beta_pow[j][0] = b0;
        a.beta_pow[j][1] = b1;
        a.sinv_half[j] = glh::mul(glh::inv(s), a.half);
        // square beta in the extension (X^2 = 7)
        uint64_t n0 = glh::add(glh::mul(b0, b0), glh::mul(7, glh::mul(b1, b1)));
        uint64_t n1 = glh::mul(2, glh::mul(b0, b1));
        b0 = n0;
        b1 = n1;
        s = glh::mul(s, s);
    }
    dim3 g((unsigned)((M + 255) / 256)), b(256);
    switch (arity_bits) {
    case 1: hipLaunchKernelGGL(k_fri_fold<1>, g, b, 0, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL(k_fri_fold<2>, g, b, 0, ctx->stream, a); break;
    case 3: hipLaunchKernelGGL(k_fri_fold<3>, g, b, 0, ctx->stream, a); break;
    case 4: hipLaunchKernelGGL(k_fri_fold<4>, g, b, 0, ctx->stream, a); break;
    default: hipLaunchKernelGGL(k_fri_fold<5>, g, b, 0, ctx->stream, a); break;
    }
    VX_HIP(hipGetLastError());
    return VX_OK;
}

int32_t vx_fri_layer_tree_dev(vx_ctx* ctx, const uint64_t* evals_d, int log_n, int arity_bits, int cap_height, vx_tree** out) {
    size_t N = (size_t)1 << log_n, M = N >> arity_bits;
    int log_m = log_n - arity_bits;
    VX_CHECK(cap_height >= 0 && cap_height <= log_m, "fri tree: cap_height %d > %d", cap_height, log_m);
    size_t total = 0, cur = M, cap = (size_t)1 << cap_height;
    while (cur > cap) {
        total += 4 * cur;
        cur >>= 1;
    }
    total += 4 * cur;
    vx_tree* t = new vx_tree{nullptr, M, cap_height, total};
    t->levels = (uint64_t*)vx_pool_alloc(ctx, total * 8);
    if (!t->levels) {
        delete t;
        return vx_fail(ctx, VX_ERR_OOM, "fri tree: cannot allocate %zu bytes", total * 8);
    }
    dim3 g((unsigned)((M + 255) / 256)), b(256);
    switch (arity_bits) {
    case 0: hipLaunchKernelGGL(k_fri_leaf_hash<0>, g, b, 0, ctx->stream, evals_d, log_n, t->levels); break;
    case 1: hipLaunchKernelGGL(k_fri_leaf_hash<1>, g, b, 0, ctx->stream, evals_d, log_n, t->levels); break;
    case 2: hipLaunchKernelGGL(k_fri_leaf_hash<2>, g, b, 0, ctx->stream, evals_d, log_n, t->levels); break;
    case 3: hipLaunchKernelGGL(k_fri_leaf_hash<3>, g, b, 0, ctx->stream, evals_d, log_n, t->levels); break;
    case 4: hipLaunchKernelGGL(k_fri_leaf_hash<4>, g, b, 0, ctx->stream, evals_d, log_n, t->levels); break;
    default: hipLaunchKernelGGL(k_fri_leaf_hash<5>, g, b, 0, ctx->stream, evals_d, log_n, t->levels); break;
    }
    vx_merkle_levels_launch(ctx, t->levels, M, cap);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        vx_pool_free(ctx, t->levels);
        delete t;
        return vx_fail(ctx, VX_ERR_DEVICE, "fri tree launch: %s", hipGetErrorString(le));
    }
    *out = t;
    return VX_OK;
}

int32_t vx_fri_leaves_dev(vx_ctx* ctx, const uint64_t* evals, int log_n, int arity_bits, const uint64_t* leaf_idx, size_t n_idx,
                          uint64_t* out);

extern "C" {
int32_t vx_fri_layer_tree(vx_ctx* ctx, const vx_buf* evals, int log_n, int arity_bits, int cap_height, vx_tree** out) {
    if (!ctx || !evals || !out) return VX_ERR_ARG;
    VX_CHECK(arity_bits >= 0 && arity_bits <= 5 && arity_bits <= log_n && log_n <= 30, "fri tree: bad log_n %d / arity_bits %d", log_n, arity_bits);
    VX_CHECK(evals->n >= ((size_t)2 << log_n), "fri tree: evals too small");
    return vx_fri_layer_tree_dev(ctx, evals->d, log_n, arity_bits, cap_height, out);
}

int32_t vx_fri_leaves(vx_ctx* ctx, const vx_buf* evals, int log_n, int arity_bits, const uint64_t* leaf_idx, size_t n_idx,
                      uint64_t* out) {
    if (!ctx || !evals || !leaf_idx || !out) return VX_ERR_ARG;
    VX_CHECK(arity_bits >= 0 && arity_bits <= 5 && arity_bits <= log_n && log_n <= 30, "fri leaves: bad shape");
    size_t N = (size_t)1 << log_n, M = N >> arity_bits;
    VX_CHECK(evals->n >= 2 * N, "fri leaves: evals too small");
    for (size_t i = 0; i < n_idx; ++i) VX_CHECK(leaf_idx[i] < M, "fri leaves: index out of range");
    return vx_fri_leaves_dev(ctx, evals->d, log_n, arity_bits, leaf_idx, n_idx, out);
}
}  // extern "C"

int32_t vx_fri_leaves_dev(vx_ctx* ctx, const uint64_t* evals_d, int log_n, int arity_bits, const uint64_t* leaf_idx, size_t n_idx,
                          uint64_t* out) {
    if (!n_idx) return VX_OK;
    size_t tot = n_idx << arity_bits;
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, n_idx + 2 * tot, &sc));
    VX_HIP(hipMemcpyAsync(sc, leaf_idx, n_idx * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_fri_gather_leaves, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, evals_d, log_n,
                       arity_bits, (const uint64_t*)sc, n_idx, sc + n_idx);
    VX_HIP(hipGetLastError());
    VX_HIP(hipMemcpyAsync(out, sc + n_idx, 2 * tot * 8, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}

extern "C" {
int32_t vx_fri_pow(vx_ctx* ctx, const uint64_t state[12], int pos, int bits, uint64_t* nonce) {
    if (!ctx || !state || !nonce) return VX_ERR_ARG;
    VX_CHECK(pos >= 0 && pos < 8 && bits >= 0 && bits <= 40, "fri pow: bad pos %d / bits %d", pos, bits);
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, 16, &sc));
    unsigned long long* best = (unsigned long long*)(sc + 12);
    unsigned long long init = ~0ULL, h = ~0ULL;
    VX_HIP(hipMemcpyAsync(sc, state, 12 * 8, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemcpyAsync(best, &init, 8, hipMemcpyHostToDevice, ctx->stream));
    // first launch: 2^(bits+2) candidates (misses with probability e^-4), later ones 2^22; the smallest nonce wins
    const uint64_t limit = 1ULL << (bits + 12 > 62 ? 62 : bits + 12);  // far beyond the expected 2^bits tries
    uint64_t chunk = 1ULL << (bits + 2 < 16 ? 16 : (bits + 2 > 22 ? 22 : bits + 2));
    for (uint64_t base = 0; base < limit; base += chunk, chunk = 1ULL << 22) {
        const unsigned blocks = (unsigned)(chunk / 256 < 2048 ? chunk / 256 : 2048);
        hipLaunchKernelGGL(k_fri_pow, dim3(blocks), dim3(256), 0, ctx->stream, sc, pos, bits, base, chunk, best);
        VX_HIP(hipGetLastError());
        VX_HIP(hipMemcpyAsync(&h, best, 8, hipMemcpyDeviceToHost, ctx->stream));
        VX_HIP(hipStreamSynchronize(ctx->stream));
        if (h != ~0ULL) {
            *nonce = h;
            return VX_OK;
        }
    }
    return vx_fail(ctx, VX_ERR_POW, "fri pow: no nonce below 2^%d", bits + 12);
}
}
