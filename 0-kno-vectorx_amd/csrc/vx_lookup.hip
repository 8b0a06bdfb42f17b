// Auxiliary-round helpers for lookup arguments (logUp): exclusive prefix sums of field columns (the running-sum
// column Z of an AIR's bus) and the auxiliary-trace generator of the LookupAir test AIR (air.cuh).
// No counterpart in /root/reference: its lookups live inside curta (starkyx v1.0.0, not vendored).
#include "air.cuh"
#include "vx_internal.h"

// ---- exclusive scan (mod p) of n_cols independent columns of n elements, in place ---------------------------------
// pass 1: a block reduces SCAN_TILE consecutive elements of one column; pass 2: one block per column scans the tile
// sums; pass 3: every tile is scanned again with its offset.  n is a power of two.
constexpr int SCAN_TILE = 4096, SCAN_PER_LANE = SCAN_TILE / 256;

__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t* lds, uint64_t* total) {
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint64_t x = t >= d ? lds[t - d] : 0;
        __syncthreads();
        if (t >= d) lds[t] = gl_add(lds[t], x);
        __syncthreads();
    }
    const uint64_t incl = lds[t];
    if (total) *total = lds[255];
    __syncthreads();
    return gl_sub(incl, v);
}
__global__ __launch_bounds__(256) void k_scan_tile_sums(const uint64_t* data, size_t n, size_t tiles_per_col, uint64_t* sums) {
    __shared__ uint64_t lds[256];
    const size_t tile = blockIdx.x, col = tile / tiles_per_col, t0 = (tile % tiles_per_col) * SCAN_TILE;
    const uint64_t* p = data + col * n + t0;
    uint64_t s = 0;
    for (int k = 0; k < SCAN_PER_LANE; ++k) {
        const size_t i = (size_t)threadIdx.x * SCAN_PER_LANE + k;
        if (t0 + i < n) s = gl_add(s, p[i]);
    }
    uint64_t total;
    block_exclusive_scan(s, lds, &total);
    if (threadIdx.x == 0) sums[tile] = total;
}
__global__ __launch_bounds__(256) void k_scan_sums(uint64_t* sums, size_t tiles_per_col) {  // one block per column
    __shared__ uint64_t lds[256];
    uint64_t* p = sums + blockIdx.x * tiles_per_col;
    uint64_t carry = 0;
    for (size_t base = 0; base < tiles_per_col; base += 256) {
        const size_t i = base + threadIdx.x;
        const uint64_t v = i < tiles_per_col ? p[i] : 0;
        uint64_t total;
        const uint64_t ex = block_exclusive_scan(v, lds, &total);
        if (i < tiles_per_col) p[i] = gl_add(ex, carry);
        carry = gl_add(carry, total);
    }
}
__global__ __launch_bounds__(256) void k_scan_apply(uint64_t* data, size_t n, size_t tiles_per_col, const uint64_t* sums) {
    __shared__ uint64_t lds[256];
    const size_t tile = blockIdx.x, col = tile / tiles_per_col, t0 = (tile % tiles_per_col) * SCAN_TILE;
    uint64_t* p = data + col * n + t0;
    uint64_t v[SCAN_PER_LANE], s = 0;
    for (int k = 0; k < SCAN_PER_LANE; ++k) {
        const size_t i = (size_t)threadIdx.x * SCAN_PER_LANE + k;
        v[k] = t0 + i < n ? p[i] : 0;
        s = gl_add(s, v[k]);
    }
    uint64_t run = gl_add(block_exclusive_scan(s, lds, nullptr), sums[tile]);
    for (int k = 0; k < SCAN_PER_LANE; ++k) {
        const size_t i = (size_t)threadIdx.x * SCAN_PER_LANE + k;
        if (t0 + i < n) p[i] = run;
        run = gl_add(run, v[k]);
    }
}
// data: n_cols columns of n = 2^log_n elements (column stride n) -> exclusive prefix sums; totals (optional, device,
// n_cols words) receive each column's full sum.
int32_t vx_scan_cols_dev(vx_ctx* ctx, uint64_t* data, int log_n, size_t n_cols, uint64_t* totals_host) {
    const size_t n = (size_t)1 << log_n, tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    uint64_t* sums;
    VX_TRY(vx_scratch(ctx, tiles * n_cols + 1, &sums));
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)(tiles * n_cols)), dim3(256), 0, ctx->stream, (const uint64_t*)data, n, tiles, sums);
    if (totals_host) {  // column total = sum of its tile sums: take them before the scan overwrites the array
        std::vector<uint64_t> h(tiles * n_cols);
        VX_HIP(hipMemcpyAsync(h.data(), sums, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        VX_HIP(hipStreamSynchronize(ctx->stream));
        for (size_t c = 0; c < n_cols; ++c) {
            uint64_t t = 0;
            for (size_t k = 0; k < tiles; ++k) t = glh::add(t, h[c * tiles + k]);
            totals_host[c] = t;
        }
    }
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)n_cols), dim3(256), 0, ctx->stream, sums, tiles);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)(tiles * n_cols)), dim3(256), 0, ctx->stream, data, n, tiles, (const uint64_t*)sums);
    VX_HIP(hipGetLastError());
    return VX_OK;
}

// ---- LookupAir (AIR 5) auxiliary columns -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lookup_aux(const uint64_t* tr, size_t n, gl2 beta, gl2 gamma, uint64_t* aux) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const gl2 g2 = gl2_mul(gamma, gamma);
    auto fp = [&](uint64_t a, uint64_t b, uint64_t c) {
        gl2 d = gl2_add(beta, gl2_add(gl2_scale(gamma, b), gl2_scale(g2, c)));
        d.a = gl_add(d.a, a);
        return d;
    };
    const uint64_t ti = i & 255, ta = ti & 15, tb = ti >> 4;
    const gl2 d0 = fp(tr[0 * n + i], tr[1 * n + i], tr[2 * n + i]), d1 = fp(tr[3 * n + i], tr[4 * n + i], tr[5 * n + i]);
    const gl2 dt = fp(ta, tb, ta ^ tb);
    const gl2 h = gl2_add(gl2_inv(d0), gl2_inv(d1)), ht = gl2_scale(gl2_inv(dt), tr[6 * n + i]);
    aux[0 * n + i] = h.a, aux[1 * n + i] = h.b, aux[2 * n + i] = ht.a, aux[3 * n + i] = ht.b;
    const gl2 dz = gl2_sub(h, ht);  // Z(next) - Z(this): turned into the running sum by the scan
    aux[4 * n + i] = dz.a, aux[5 * n + i] = dz.b;
}
int32_t vx_lookup_air_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub) {
    (void)aux_pub, (void)pub;
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_lookup_aux, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, trace, n, gl2{chal[0], chal[1]},
                       gl2{chal[2], chal[3]}, aux);
    VX_HIP(hipGetLastError());
    return vx_scan_cols_dev(ctx, aux + 4 * n, log_n, 2, nullptr);
}
