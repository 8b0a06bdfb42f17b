// K4: Poseidon permutation batches and Merkle trees with caps on the GPU.
// Replaces plonky2::hash::merkle_tree::MerkleTree::{new,prove} and
// hashing::{hash_n_to_m_no_pad,compress} (v0.2.0).  Leaves are read where the
// LDE left them (column-major, natural row order): lane t hashes row t of every
// column, so each absorb step is one coalesced 512-byte load per wave; the
// plonky2 leaf index bitrev(t) only decides where the 32-byte digest is written.
#include "poseidon.cuh"
#include "vx_internal.h"

__global__ __launch_bounds__(256) void k_poseidon_batch(uint64_t* states, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = states[12 * i + k];
    poseidon_permute(s);
#pragma unroll
    for (int k = 0; k < 12; ++k) states[12 * i + k] = s[k];
}

// hash_or_noop of every leaf.  LAYOUT as in vx.h.
#ifndef VX_HASH_WAVES
#define VX_HASH_WAVES 6  // 80 VGPRs; measured 2^18 x 4337 leaves: 60.5 ms at 4 waves per SIMD, 59.9 at 6
#endif
template <int LAYOUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VX_HASH_WAVES, 8))) void k_hash_leaves(const uint64_t* data, size_t n_leaves, int log_n, size_t leaf_len,
                                                     uint64_t* digests) {
    size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= n_leaves) return;
    const size_t estride = LAYOUT == VX_LEAVES_ROW_MAJOR ? 1 : n_leaves;
    const uint64_t* src = LAYOUT == VX_LEAVES_ROW_MAJOR ? data + p * leaf_len : data + p;
    size_t j = LAYOUT == VX_LEAVES_COLS_BITREV ? brev32((uint32_t)p, log_n) : p;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = 0;
    if (leaf_len <= 4) {
        for (size_t e = 0; e < leaf_len; ++e) s[e] = src[e * estride];
    } else {
        size_t e = 0;
        for (; e + 8 <= leaf_len; e += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] = src[(e + k) * estride];
            poseidon_permute(s);
        }
        if (e < leaf_len) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (e + k < leaf_len) s[k] = src[(e + k) * estride];
            poseidon_permute(s);
        }
    }
    uint64_t* d = digests + 4 * j;
    d[0] = s[0];
    d[1] = s[1];
    d[2] = s[2];
    d[3] = s[3];
}

// Small trees: 16 lanes per leaf (cooperative permutation, poseidon.cuh); block = 16 leaves.
template <int LAYOUT>
__global__ __launch_bounds__(256) void k_hash_leaves_coop(const uint64_t* data, size_t n_leaves, int log_n, size_t leaf_len,
                                                          uint64_t* digests) {
    __shared__ uint64_t lds[16 * 12];
    const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
    uint64_t* g = lds + 12 * grp;
    const size_t t = blockIdx.x * (size_t)16 + grp;
    const size_t p = t < n_leaves ? t : n_leaves - 1;  // surplus groups redo the last leaf and do not write
    const size_t estride = LAYOUT == VX_LEAVES_ROW_MAJOR ? 1 : n_leaves;
    const uint64_t* src = LAYOUT == VX_LEAVES_ROW_MAJOR ? data + p * leaf_len : data + p;
    const size_t j = LAYOUT == VX_LEAVES_COLS_BITREV ? brev32((uint32_t)p, log_n) : p;
    uint64_t s = 0;
    if (leaf_len <= 4) {
        if ((size_t)l < leaf_len) s = src[(size_t)l * estride];
    } else {
        for (size_t e = 0; e < leaf_len; e += 8) {
            if (l < 8 && e + l < leaf_len) s = src[(e + l) * estride];
            s = poseidon_permute_coop(s, l, g);
        }
    }
    if (l < 4 && t < n_leaves) digests[4 * j + l] = s;
}
// parent[i] = compress(child[2i], child[2i+1]), 16 lanes per parent
__global__ __launch_bounds__(256) void k_merkle_level_coop(const uint64_t* child, uint64_t* parent, size_t n_parent) {
    __shared__ uint64_t lds[16 * 12];
    const int l = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const size_t t = blockIdx.x * (size_t)16 + grp;
    const size_t i = t < n_parent ? t : n_parent - 1;
    uint64_t s = l < 8 ? child[8 * i + l] : 0;
    s = poseidon_permute_coop(s, l, lds + 12 * grp);
    if (l < 4 && t < n_parent) parent[4 * i + l] = s;
}
constexpr size_t COOP_MAX_LEAVES = 16384;  // below this the one-lane kernels cannot fill the chip (1024 SIMDs x 64 lanes)

// parent[i] = compress(child[2i], child[2i+1])
__global__ __launch_bounds__(256) void k_merkle_level(const uint64_t* child, uint64_t* parent, size_t n_parent) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n_parent) return;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = child[8 * i + k];
    s[8] = s[9] = s[10] = s[11] = 0;
    poseidon_permute(s);
    parent[4 * i] = s[0];
    parent[4 * i + 1] = s[1];
    parent[4 * i + 2] = s[2];
    parent[4 * i + 3] = s[3];
}

__global__ void k_gather_siblings(const uint64_t* levels, size_t n_leaves, int depth, const uint64_t* idx, size_t n_idx,
                                  uint64_t* out) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= n_idx * depth) return;
    size_t k = t / depth;
    int lvl = (int)(t - k * depth);
    size_t off = 0, cur = n_leaves;
    for (int l = 0; l < lvl; ++l) {
        off += 4 * cur;
        cur >>= 1;
    }
    size_t node = (idx[k] >> lvl) ^ 1;
    for (int e = 0; e < 4; ++e) out[4 * t + e] = levels[off + 4 * node + e];
}


// ---- witness of PoseidonAir (the permutation as a STARK table: air_library.py poseidon_builder; 48 columns, 32 rows per permutation).
// Lane p walks the 30 rounds of permutation p in the PLAIN schedule (constant layer, s-box, MDS: the rows of the table are the
// states entering each round, not the folded form the hashing kernels use) and writes the state, x^2, x^4 and x^7 of every row;
// rows 30 and 31 of a block hold the output.  trace: column-major [48][32 n_perm].
static __constant__ uint64_t POSEIDON_RC_PLAIN[360] = VX_POSEIDON_RC_INIT;
__global__ __launch_bounds__(256) void k_poseidon_air_trace(const uint64_t* in, size_t n_perm, uint64_t* tr) {
    const size_t p = blockIdx.x * (size_t)256 + threadIdx.x;
    if (p >= n_perm) return;
    constexpr uint32_t C[12] = VX_POSEIDON_MDS_CIRC_INIT;
    const size_t n = 32 * n_perm;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl_canon(in[12 * p + i]);
#pragma unroll 1
    for (int r = 0; r < 32; ++r) {
        const size_t row = 32 * p + r;
        const bool full = r < 4 || (r >= 26 && r < 30);
        uint64_t y[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const uint64_t x = r < 30 ? gl_add(s[i], POSEIDON_RC_PLAIN[12 * r + i]) : s[i];
            const uint64_t a = gl_mul(x, x), b = gl_mul(a, a), t = gl_mul(gl_mul(x, a), b);
            tr[(size_t)i * n + row] = s[i];
            tr[(size_t)(12 + i) * n + row] = a;
            tr[(size_t)(24 + i) * n + row] = b;
            tr[(size_t)(36 + i) * n + row] = t;
            y[i] = (full || i == 0) ? t : x;
        }
        if (r < 30) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                unsigned __int128 acc = q == 0 ? (unsigned __int128)y[0] * VX_POSEIDON_MDS_DIAG0 : 0;
#pragma unroll
                for (int i = 0; i < 12; ++i) acc += (unsigned __int128)y[(i + q) % 12] * C[i];
                s[q] = gl_reduce128((uint64_t)(acc >> 64), (uint64_t)acc);
            }
        }
    }
}

// levels above the leaf digests, down to `cap` nodes (levels = digests of level 0 followed by each parent level)
void vx_merkle_levels_launch(vx_ctx* ctx, uint64_t* levels, size_t n_leaves, size_t cap) {
    size_t off = 0, cur = n_leaves;
    while (cur > cap) {
        const size_t np = cur >> 1;
        if (np <= COOP_MAX_LEAVES)
            hipLaunchKernelGGL(k_merkle_level_coop, dim3((unsigned)((np + 15) / 16)), dim3(256), 0, ctx->stream, levels + off, levels + off + 4 * cur, np);
        else
            hipLaunchKernelGGL(k_merkle_level, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, ctx->stream, levels + off, levels + off + 4 * cur, np);
        off += 4 * cur;
        cur = np;
    }
}

int32_t vx_merkle_build_dev(vx_ctx* ctx, const uint64_t* data, size_t n_leaves, size_t leaf_len, int layout,
                            int cap_height, vx_tree** out) {
    int log_n = 0;
    while (((size_t)1 << log_n) < n_leaves) ++log_n;
    VX_CHECK(((size_t)1 << log_n) == n_leaves, "merkle: n_leaves %zu is not a power of two", n_leaves);
    VX_CHECK(cap_height >= 0 && cap_height <= log_n, "merkle: cap_height %d > log2(n_leaves) %d", cap_height, log_n);
    VX_CHECK(leaf_len >= 1, "merkle: empty leaves");
    size_t total = 0, cur = n_leaves, cap = (size_t)1 << cap_height;
    while (cur > cap) {
        total += 4 * cur;
        cur >>= 1;
    }
    total += 4 * cur;
    vx_tree* t = new vx_tree{nullptr, n_leaves, cap_height, total};
    t->levels = (uint64_t*)vx_pool_alloc(ctx, total * 8);
    if (!t->levels) {
        delete t;
        return vx_fail(ctx, VX_ERR_OOM, "merkle: cannot allocate %zu bytes", total * 8);
    }
    unsigned g = (unsigned)((n_leaves + 255) / 256);
    if (n_leaves <= COOP_MAX_LEAVES && leaf_len > 4) {
        const unsigned gc = (unsigned)((n_leaves + 15) / 16);
        if (layout == VX_LEAVES_ROW_MAJOR)
            hipLaunchKernelGGL(k_hash_leaves_coop<VX_LEAVES_ROW_MAJOR>, dim3(gc), dim3(256), 0, ctx->stream, data, n_leaves, log_n, leaf_len, t->levels);
        else if (layout == VX_LEAVES_COLS_BITREV)
            hipLaunchKernelGGL(k_hash_leaves_coop<VX_LEAVES_COLS_BITREV>, dim3(gc), dim3(256), 0, ctx->stream, data, n_leaves, log_n, leaf_len, t->levels);
        else
            hipLaunchKernelGGL(k_hash_leaves_coop<VX_LEAVES_COLS>, dim3(gc), dim3(256), 0, ctx->stream, data, n_leaves, log_n, leaf_len, t->levels);
    } else if (layout == VX_LEAVES_ROW_MAJOR)
        hipLaunchKernelGGL(k_hash_leaves<VX_LEAVES_ROW_MAJOR>, dim3(g), dim3(256), 0, ctx->stream, data, n_leaves, log_n, leaf_len, t->levels);
    else if (layout == VX_LEAVES_COLS_BITREV)
        hipLaunchKernelGGL(k_hash_leaves<VX_LEAVES_COLS_BITREV>, dim3(g), dim3(256), 0, ctx->stream, data, n_leaves, log_n, leaf_len, t->levels);
    else
        hipLaunchKernelGGL(k_hash_leaves<VX_LEAVES_COLS>, dim3(g), dim3(256), 0, ctx->stream, data, n_leaves, log_n, leaf_len, t->levels);
    vx_merkle_levels_launch(ctx, t->levels, n_leaves, cap);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        vx_pool_free(ctx, t->levels);
        delete t;
        return vx_fail(ctx, VX_ERR_DEVICE, "merkle launch: %s", hipGetErrorString(le));
    }
    *out = t;
    return VX_OK;
}

extern "C" {
int32_t vx_poseidon_permute_batch(vx_ctx* ctx, vx_buf* states, size_t n) {
    if (!ctx || !states) return VX_ERR_ARG;
    VX_CHECK(12 * n <= states->n, "poseidon batch: %zu states exceed the buffer", n);
    if (n == 0) return VX_OK;
    hipLaunchKernelGGL(k_poseidon_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, states->d, n);
    VX_HIP(hipGetLastError());
    return VX_OK;
}
int32_t vx_poseidon_air_trace(vx_ctx* ctx, const vx_buf* states, size_t n_perm, vx_buf* trace_out) {
    if (!ctx || !states || !trace_out) return VX_ERR_ARG;
    VX_CHECK(n_perm >= 1 && (n_perm & (n_perm - 1)) == 0 && n_perm <= ((size_t)1 << 22), "poseidon air trace: %zu permutations (a power of two, at most 2^22)", n_perm);
    VX_CHECK(12 * n_perm <= states->n && 48 * 32 * n_perm <= trace_out->n, "poseidon air trace: buffers too small");
    hipLaunchKernelGGL(k_poseidon_air_trace, dim3((unsigned)((n_perm + 255) / 256)), dim3(256), 0, ctx->stream, states->d, n_perm, trace_out->d);
    VX_HIP(hipGetLastError());
    return VX_OK;
}
int32_t vx_merkle_build(vx_ctx* ctx, const vx_buf* data, size_t off, size_t n_leaves, size_t leaf_len, int layout,
                        int cap_height, vx_tree** out) {
    if (!ctx || !data || !out) return VX_ERR_ARG;
    VX_CHECK(layout >= 0 && layout <= 2, "merkle: bad layout %d", layout);
    VX_CHECK(off + n_leaves * leaf_len <= data->n, "merkle: %zu leaves x %zu exceed the buffer", n_leaves, leaf_len);
    return vx_merkle_build_dev(ctx, data->d + off, n_leaves, leaf_len, layout, cap_height, out);
}
int32_t vx_merkle_free(vx_ctx* ctx, vx_tree* tree) {
    if (!ctx || !tree) return VX_ERR_ARG;
    vx_pool_free(ctx, tree->levels);
    delete tree;
    return VX_OK;
}
int32_t vx_merkle_cap(vx_ctx* ctx, const vx_tree* tree, uint64_t* cap_out) {
    if (!ctx || !tree || !cap_out) return VX_ERR_ARG;
    size_t cap = (size_t)4 << tree->cap_height;
    VX_HIP(hipMemcpyAsync(cap_out, tree->levels + tree->total - cap, cap * 8, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}
int32_t vx_merkle_leaf_digests(vx_ctx* ctx, const vx_tree* tree, uint64_t* out) {
    if (!ctx || !tree || !out) return VX_ERR_ARG;
    VX_HIP(hipMemcpyAsync(out, tree->levels, tree->n_leaves * 32, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}
int32_t vx_merkle_open(vx_ctx* ctx, const vx_tree* tree, const uint64_t* leaf_idx, size_t n_idx, uint64_t* siblings_out) {
    if (!ctx || !tree || !leaf_idx || !siblings_out) return VX_ERR_ARG;
    int log_n = 0;
    while (((size_t)1 << log_n) < tree->n_leaves) ++log_n;
    int depth = log_n - tree->cap_height;
    for (size_t i = 0; i < n_idx; ++i) VX_CHECK(leaf_idx[i] < tree->n_leaves, "merkle open: index out of range");
    if (depth == 0 || n_idx == 0) return VX_OK;
    uint64_t* sc;
    size_t tot = n_idx * depth;
    VX_TRY(vx_scratch(ctx, n_idx + 4 * tot, &sc));
    VX_HIP(hipMemcpyAsync(sc, leaf_idx, n_idx * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_gather_siblings, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, tree->levels,
                       tree->n_leaves, depth, sc, n_idx, sc + n_idx);
    VX_HIP(hipGetLastError());
    VX_HIP(hipMemcpyAsync(siblings_out, sc + n_idx, 4 * tot * 8, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}
}
