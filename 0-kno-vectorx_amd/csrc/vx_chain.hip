// K8 (witness side) + statement level: the header-chain map/reduce of
// circuits/builder/subchain_verification.rs:56-303 on the GPU.
//   - BLAKE2b-256 of every encoded header (hash_encoded_header,
//     circuits/builder/header.rs:14-19): one lane per header, the <= 280
//     compressions of a header are inherently sequential.
//   - header decoding (decoder.rs:104-157), link checks, 8-leaf SHA-256 roots
//     (map closure, subchain_verification.rs:81-232): one lane per map job.
//   - the reduce tree (subchain_verification.rs:233-289) in LDS, ping-pong buffers.
// Violated in-circuit assertions are reported through a status word; the
// reference would fail witness generation / proof verification instead.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "vx_internal.h"

__device__ __constant__ uint64_t B2B_IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                              0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                              0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
__device__ __constant__ uint8_t B2B_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

__device__ __forceinline__ uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

#define B2B_G(a, b, c, d, x, y)                               \
    v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 24); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 63);

__device__ void blake2b_compress(uint64_t* h, const uint64_t* m, uint64_t t, bool last) {
    uint64_t v[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = h[i];
        v[i + 8] = B2B_IV[i];
    }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
#pragma unroll 1
    for (int r = 0; r < 12; ++r) {
        const uint8_t* s = B2B_SIGMA[r];
        B2B_G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        B2B_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        B2B_G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        B2B_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        B2B_G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        B2B_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        B2B_G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        B2B_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}

// One lane per message.  Messages are zero padded to `stride` (multiple of 8) bytes,
// so the final partial block can be loaded whole and masked.
__global__ __launch_bounds__(64) void k_blake2b_256(const uint8_t* msgs, size_t stride, const uint32_t* sizes, size_t n,
                                                    uint8_t* digests) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t* p = (const uint64_t*)(msgs + i * stride);
    const uint32_t len = sizes[i];
    uint64_t h[8], m[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = B2B_IV[k];
    h[0] ^= 0x01010000ULL ^ 32;
    uint32_t off = 0;
    while (len - off > 128) {
#pragma unroll
        for (int k = 0; k < 16; ++k) m[k] = p[(off >> 3) + k];
        off += 128;
        blake2b_compress(h, m, off, false);
    }
    const uint32_t rem = len - off;  // 0..128 (0 only for the empty message)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        uint32_t b = 8 * k;
        uint64_t w = 0;
        if (b < rem) {
            w = (off + b + 8 <= stride) ? p[(off >> 3) + k] : 0;
            if (rem - b < 8) w &= (1ULL << (8 * (rem - b))) - 1;
        }
        m[k] = w;
    }
    blake2b_compress(h, m, len, true);
    uint64_t* d = (uint64_t*)(digests + 32 * i);
    d[0] = h[0];
    d[1] = h[1];
    d[2] = h[2];
    d[3] = h[3];
}

// ---------------------------------------------------------------- SHA-256
__device__ __constant__ uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
__device__ void sha256_block(uint32_t* h, uint32_t* w) {  // w: 16 big-endian words, clobbered
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
        uint32_t wi;
        if (i < 16) wi = w[i];
        else {
            uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
            uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
            wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
            w[i & 15] = wi;
        }
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + SHA_K[i] + wi;
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + S0 + mj;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
// SHA-256 of the 64-byte message l || r (two compressions: data block + padding block)
__device__ void sha256_pair(const uint8_t* l, const uint8_t* r, uint8_t* out) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint32_t w[16];
    for (int i = 0; i < 8; ++i) {
        w[i] = ((uint32_t)l[4 * i] << 24) | ((uint32_t)l[4 * i + 1] << 16) | ((uint32_t)l[4 * i + 2] << 8) | l[4 * i + 3];
        w[8 + i] = ((uint32_t)r[4 * i] << 24) | ((uint32_t)r[4 * i + 1] << 16) | ((uint32_t)r[4 * i + 2] << 8) | r[4 * i + 3];
    }
    sha256_block(h, w);
    for (int i = 0; i < 16; ++i) w[i] = 0;
    w[0] = 0x80000000u;
    w[15] = 512;
    sha256_block(h, w);
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = (uint8_t)(h[i] >> 24);
        out[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(h[i] >> 8);
        out[4 * i + 3] = (uint8_t)h[i];
    }
}
__global__ void k_sha256_pairs(const uint8_t* in, size_t n, uint8_t* out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    sha256_pair(in + 64 * i, in + 64 * i + 32, out + 32 * i);
}

// ---------------------------------------------------------------- map / reduce
struct MapOut {  // MapReduceSubchainVariable, subchain_verification.rs:43-53
    uint32_t num_blocks, start_block, end_block, pad;
    uint8_t start_header_hash[32], start_parent[32], end_header_hash[32], state_root[32], data_root[32];
};
enum { ST_LINK = 1, ST_FIRST = 2, ST_LAST = 4, ST_REDUCE = 8, ST_COMPACT = 16 };

__device__ bool eq32(const uint8_t* a, const uint8_t* b) {
    bool e = true;
    for (int i = 0; i < 32; ++i) e &= a[i] == b[i];
    return e;
}
__device__ void cp32(uint8_t* d, const uint8_t* s) {
    for (int i = 0; i < 32; ++i) d[i] = s[i];
}
// decoder.rs:39-92
__device__ bool decode_compact(const uint8_t* b, uint32_t* val, uint32_t* mode) {
    uint32_t m = b[0] & 3;
    *mode = m;
    if (m == 0) *val = b[0] >> 2;
    else if (m == 1) *val = ((uint32_t)b[0] | ((uint32_t)b[1] << 8)) >> 2;
    else if (m == 2) *val = ((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24)) >> 2;
    else {
        *val = (uint32_t)b[1] | ((uint32_t)b[2] << 8) | ((uint32_t)b[3] << 16) | ((uint32_t)b[4] << 24);
        return (b[0] >> 2) == 0;
    }
    return true;
}

// decode_header (decoder.rs:104-157) of n encoded headers, one lane each: the same decode_compact / offset select the
// map job below uses.  ok[i] = 0 where the mode-3 assertion (decoder.rs:83-89) fails.
__global__ __launch_bounds__(64) void k_decode_headers(const uint8_t* headers, size_t stride, const uint32_t* sizes, size_t n, uint32_t* numbers,
                                                       uint8_t* modes, uint8_t* ok, uint8_t* parents, uint8_t* state_roots, uint8_t* data_roots) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* hb = headers + i * stride;
    const uint32_t sz = sizes[i];
    uint32_t num = 0, mode = 0;
    ok[i] = decode_compact(hb + 32, &num, &mode) ? 1 : 0;
    numbers[i] = num;
    modes[i] = (uint8_t)mode;
    const int off = mode == 0 ? 33 : mode == 1 ? 34 : mode == 2 ? 36 : 37;
    cp32(parents + 32 * i, hb);
    cp32(state_roots + 32 * i, hb + off);
    cp32(data_roots + 32 * i, hb + (sz == 0 ? 0 : sz - 32));
}
// decode_precommit (decoder.rs:159-200), one lane per 53-byte message
__global__ __launch_bounds__(64) void k_decode_precommits(const uint8_t* pc, size_t n, uint8_t* ok, uint8_t* hashes, uint32_t* numbers, uint64_t* rounds,
                                                          uint64_t* set_ids) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = pc + 53 * i;
    ok[i] = p[0] == 1;  // :165-167
    cp32(hashes + 32 * i, p + 1);
    uint32_t bn = 0;
    uint64_t r = 0, sid = 0;
    for (int k = 0; k < 4; ++k) bn |= (uint32_t)p[33 + k] << (8 * k);
    for (int k = 0; k < 8; ++k) r |= (uint64_t)p[37 + k] << (8 * k), sid |= (uint64_t)p[45 + k] << (8 * k);
    numbers[i] = bn, rounds[i] = r, set_ids[i] = sid;
}

// blockDim.x = J (number of map jobs, power of two <= 1024); LDS: 2 * J * sizeof(MapOut)
__global__ void k_subchain(const uint8_t* headers, size_t stride, const uint32_t* sizes, size_t n_fetched,
                           const uint8_t* digests, uint32_t global_start, uint32_t global_end, MapOut* result,
                           uint32_t* status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    MapOut* buf0 = (MapOut*)lds_raw;
    MapOut* buf1 = buf0 + blockDim.x;
    const uint32_t j = threadIdx.x;
    const uint32_t M = 8;
    uint32_t st = 0;
    {
        const uint32_t batch_start = global_start + 8 * j + 1, batch_end = batch_start + M - 1;
        const bool batch_disabled = global_end < batch_start;
        bool noop = batch_disabled;
        uint8_t zero32[32];
        for (int i = 0; i < 32; ++i) zero32[i] = 0;
        // blake2b-256 of the empty message (zero-size padding header); never used as a link target
        uint8_t prev_hash[32], cur_hash[32], state_leaves[8][32], data_leaves[8][32];
        uint32_t prev_num = 0, end_block = 0, nb_enabled = 0, num_headers = 0, first_num = 0;
        MapOut o;
        for (int i = 0; i < 32; ++i) o.end_header_hash[i] = 0;
        for (uint32_t i = 0; i < M; ++i) {
            const uint32_t blk = batch_start + i;
            const bool present = blk <= global_end && (size_t)(blk - global_start - 1) < n_fetched;
            uint32_t num = 0, mode = 0;
            const uint8_t* hb = nullptr;
            uint8_t parent[32];
            if (present) {
                const size_t idx = blk - global_start - 1;
                hb = headers + idx * stride;
                const uint32_t sz = sizes[idx];
                cp32(cur_hash, digests + 32 * idx);
                cp32(parent, hb);
                if (!decode_compact(hb + 32, &num, &mode) && !noop) st |= ST_COMPACT;
                const int off = mode == 0 ? 33 : mode == 1 ? 34 : mode == 2 ? 36 : 37;
                cp32(state_leaves[i], hb + off);
                cp32(data_leaves[i], hb + (sz == 0 ? 0 : sz - 32));
            } else {
                // (zeros, size 0) padding header: decodes to number 0, zero roots
                for (int k = 0; k < 32; ++k) cur_hash[k] = 0, parent[k] = 0;
                cp32(state_leaves[i], zero32);
                cp32(data_leaves[i], zero32);
            }
            if (i > 0) {
                const bool linked = eq32(parent, prev_hash) && num == prev_num + 1;
                if (!(noop || linked)) st |= ST_LINK;
            } else {
                first_num = num;
                cp32(o.start_header_hash, cur_hash);
                cp32(o.start_parent, parent);
            }
            if (!noop) {
                end_block = num;
                cp32(o.end_header_hash, cur_hash);
                ++num_headers;
                ++nb_enabled;
            }
            if (num == global_end) noop = true;
            cp32(prev_hash, cur_hash);
            prev_num = num;
        }
        if (!(first_num == batch_start || batch_disabled)) st |= ST_FIRST;
        if (!(end_block == batch_end || noop)) st |= ST_LAST;
        for (uint32_t i = nb_enabled; i < M; ++i) {
            cp32(state_leaves[i], zero32);
            cp32(data_leaves[i], zero32);
        }
        for (int w = 8; w > 1; w >>= 1)
            for (int i = 0; i < w / 2; ++i) {
                uint8_t t[32];
                sha256_pair(state_leaves[2 * i], state_leaves[2 * i + 1], t);
                cp32(state_leaves[i], t);
                sha256_pair(data_leaves[2 * i], data_leaves[2 * i + 1], t);
                cp32(data_leaves[i], t);
            }
        o.num_blocks = num_headers;
        o.start_block = first_num;
        o.end_block = end_block;
        o.pad = 0;
        cp32(o.state_root, state_leaves[0]);
        cp32(o.data_root, data_leaves[0]);
        buf0[j] = o;
    }
    __syncthreads();
    MapOut *cur = buf0, *nxt = buf1;
    for (uint32_t w = blockDim.x; w > 1; w >>= 1) {
        if (j < w / 2) {
            const MapOut l = cur[2 * j], r = cur[2 * j + 1];
            const bool linked = eq32(l.end_header_hash, r.start_parent) && l.end_block == r.start_block - 1;
            const bool right_inactive = r.num_blocks == 0;
            if (!(right_inactive || linked)) st |= ST_REDUCE;
            MapOut o = l;
            o.end_block = right_inactive ? l.end_block : r.end_block;
            cp32(o.end_header_hash, right_inactive ? l.end_header_hash : r.end_header_hash);
            sha256_pair(l.state_root, r.state_root, o.state_root);
            sha256_pair(l.data_root, r.data_root, o.data_root);
            o.num_blocks = l.num_blocks + r.num_blocks;
            nxt[j] = o;
        }
        __syncthreads();
        MapOut* t = cur;
        cur = nxt;
        nxt = t;
    }
    if (st) atomicOr(status, st);
    if (j == 0) *result = cur[0];
}

extern "C" {

int32_t vx_blake2b_256_batch(vx_ctx* ctx, const vx_buf* msgs, size_t stride, const uint32_t* sizes, size_t n,
                             uint8_t* digests_out) {
    if (!ctx || !msgs || !sizes || !digests_out) return VX_ERR_ARG;
    VX_CHECK(stride % 8 == 0 && stride > 0, "blake2b batch: stride %zu must be a positive multiple of 8", stride);
    VX_CHECK(n * stride <= msgs->n * 8, "blake2b batch: %zu messages x %zu B exceed the buffer", n, stride);
    for (size_t i = 0; i < n; ++i) VX_CHECK(sizes[i] <= stride, "blake2b batch: message %zu longer than stride", i);
    if (n == 0) return VX_OK;
    uint64_t* sc;
    size_t sz_words = (n * 4 + 7) / 8, dg_words = n * 4;
    VX_TRY(vx_scratch(ctx, sz_words + dg_words, &sc));
    VX_HIP(hipMemcpyAsync(sc, sizes, n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_blake2b_256, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (const uint8_t*)msgs->d, stride,
                       (const uint32_t*)sc, n, (uint8_t*)(sc + sz_words));
    VX_HIP(hipGetLastError());
    VX_HIP(hipMemcpyAsync(digests_out, sc + sz_words, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}

int32_t vx_sha256_pairs(vx_ctx* ctx, const uint8_t* pairs64, size_t n, uint8_t* out32) {
    if (!ctx || !pairs64 || !out32) return VX_ERR_ARG;
    if (n == 0) return VX_OK;
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, n * 8 + n * 4, &sc));
    VX_HIP(hipMemcpyAsync(sc, pairs64, n * 64, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_sha256_pairs, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (const uint8_t*)sc, n,
                       (uint8_t*)(sc + n * 8));
    VX_HIP(hipGetLastError());
    VX_HIP(hipMemcpyAsync(out32, sc + n * 8, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}

int32_t vx_decode_header_batch(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes, size_t n, uint32_t* numbers_out,
                               uint8_t* modes_out, uint8_t* ok_out, uint8_t* parent_out, uint8_t* state_root_out, uint8_t* data_root_out) {
    if (!ctx || !headers || !sizes || !numbers_out || !modes_out || !ok_out || !parent_out || !state_root_out || !data_root_out) return VX_ERR_ARG;
    VX_CHECK(stride >= 72 && n >= 1 && n * stride <= headers->n * 8, "decode_header: %zu headers x %zu B exceed the buffer (stride >= 72)", n, stride);
    for (size_t i = 0; i < n; ++i) VX_CHECK(sizes[i] <= stride && (sizes[i] == 0 || sizes[i] >= 32), "decode_header: header %zu has size %u", i, sizes[i]);
    // device scratch: sizes | numbers | modes | ok | parents | state roots | data roots
    const size_t w4 = (n * 4 + 7) / 8, w1 = (n + 7) / 8, w32 = n * 4;
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, 2 * w4 + 2 * w1 + 3 * w32, &sc));
    uint32_t* d_sizes = (uint32_t*)sc;
    uint32_t* d_num = (uint32_t*)(sc + w4);
    uint8_t* d_mode = (uint8_t*)(sc + 2 * w4);
    uint8_t* d_ok = (uint8_t*)(sc + 2 * w4 + w1);
    uint8_t* d_par = (uint8_t*)(sc + 2 * w4 + 2 * w1);
    uint8_t* d_sr = d_par + 32 * n;
    uint8_t* d_dr = d_sr + 32 * n;
    VX_HIP(hipMemcpyAsync(d_sizes, sizes, n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_decode_headers, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (const uint8_t*)headers->d, stride,
                       (const uint32_t*)d_sizes, n, d_num, d_mode, d_ok, d_par, d_sr, d_dr);
    VX_HIP(hipGetLastError());
    VX_HIP(hipMemcpyAsync(numbers_out, d_num, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipMemcpyAsync(modes_out, d_mode, n, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipMemcpyAsync(ok_out, d_ok, n, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipMemcpyAsync(parent_out, d_par, 32 * n, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipMemcpyAsync(state_root_out, d_sr, 32 * n, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipMemcpyAsync(data_root_out, d_dr, 32 * n, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}

int32_t vx_decode_precommit_batch(vx_ctx* ctx, const uint8_t* precommits, size_t n, uint8_t* ok_out, uint8_t* hash_out, uint32_t* block_number_out,
                                  uint64_t* round_out, uint64_t* set_id_out) {
    if (!ctx || !precommits || !ok_out || !hash_out || !block_number_out || !round_out || !set_id_out) return VX_ERR_ARG;
    VX_CHECK(n >= 1 && n <= (1u << 20), "decode_precommit: n %zu out of range", n);
    const size_t w_in = (53 * n + 7) / 8, w1 = (n + 7) / 8, w4 = (n * 4 + 7) / 8;
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, w_in + w1 + 4 * n + w4 + 2 * n, &sc));
    uint8_t* d_in = (uint8_t*)sc;
    uint8_t* d_ok = (uint8_t*)(sc + w_in);
    uint8_t* d_hash = (uint8_t*)(sc + w_in + w1);
    uint32_t* d_bn = (uint32_t*)(sc + w_in + w1 + 4 * n);
    uint64_t* d_round = sc + w_in + w1 + 4 * n + w4;
    uint64_t* d_sid = d_round + n;
    VX_HIP(hipMemcpyAsync(d_in, precommits, 53 * n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_decode_precommits, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (const uint8_t*)d_in, n, d_ok, d_hash, d_bn, d_round, d_sid);
    VX_HIP(hipGetLastError());
    VX_HIP(hipMemcpyAsync(ok_out, d_ok, n, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipMemcpyAsync(hash_out, d_hash, 32 * n, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipMemcpyAsync(block_number_out, d_bn, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipMemcpyAsync(round_out, d_round, 8 * n, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipMemcpyAsync(set_id_out, d_sid, 8 * n, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}

int32_t vx_verify_subchain(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes, size_t n_fetched,
                           uint32_t max_headers, uint32_t trusted_block, const uint8_t trusted_hash[32],
                           uint32_t target_block, uint8_t out96[96]) {
    if (!ctx || !headers || !sizes || !trusted_hash || !out96) return VX_ERR_ARG;
    VX_CHECK(stride % 8 == 0 && stride >= 72, "verify_subchain: stride %zu must be a multiple of 8 and >= 72", stride);
    VX_CHECK(max_headers >= 8 && max_headers <= 1024, "verify_subchain: max_headers %u out of range", max_headers);
    VX_CHECK(n_fetched >= 1 && n_fetched <= max_headers, "verify_subchain: n_fetched %zu not in [1, %u]", n_fetched, max_headers);
    VX_CHECK(n_fetched * stride <= headers->n * 8, "verify_subchain: headers exceed the buffer");
    VX_CHECK(target_block > trusted_block && target_block - trusted_block == n_fetched,
             "verify_subchain: %zu headers fetched for range (%u, %u]", n_fetched, trusted_block, target_block);
    for (size_t i = 0; i < n_fetched; ++i)
        VX_CHECK(sizes[i] <= stride && (sizes[i] == 0 || sizes[i] >= 72), "verify_subchain: header %zu has size %u", i, sizes[i]);
    uint32_t J = 1;
    while (J < max_headers / 8) J <<= 1;  // (N / HEADERS_PER_MAP).next_power_of_two()
    uint64_t* sc;
    const size_t sz_words = (n_fetched * 4 + 7) / 8, dg_words = n_fetched * 4, res_words = (sizeof(MapOut) + 7) / 8;
    VX_TRY(vx_scratch(ctx, sz_words + dg_words + res_words + 1, &sc));
    uint32_t* d_sizes = (uint32_t*)sc;
    uint8_t* d_dig = (uint8_t*)(sc + sz_words);
    MapOut* d_res = (MapOut*)(sc + sz_words + dg_words);
    uint32_t* d_status = (uint32_t*)(sc + sz_words + dg_words + res_words);
    VX_HIP(hipMemcpyAsync(d_sizes, sizes, n_fetched * 4, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemsetAsync(d_status, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_blake2b_256, dim3((unsigned)((n_fetched + 63) / 64)), dim3(64), 0, ctx->stream,
                       (const uint8_t*)headers->d, stride, (const uint32_t*)d_sizes, n_fetched, d_dig);
    hipLaunchKernelGGL(k_subchain, dim3(1), dim3(J), 2 * J * sizeof(MapOut), ctx->stream, (const uint8_t*)headers->d, stride,
                       (const uint32_t*)d_sizes, n_fetched, (const uint8_t*)d_dig, trusted_block, target_block, d_res, d_status);
    VX_HIP(hipGetLastError());
    MapOut res;
    uint32_t status = 0;
    VX_HIP(hipMemcpyAsync(&res, d_res, sizeof res, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(out96, res.end_header_hash, 32);
    memcpy(out96 + 32, res.state_root, 32);
    memcpy(out96 + 64, res.data_root, 32);
    if (status) return vx_fail(ctx, VX_ERR_STATEMENT, "verify_subchain: chain rule violated (status 0x%x)", status);
    // final asserts, subchain_verification.rs:292-296
    if (memcmp(trusted_hash, res.start_parent, 32) != 0)
        return vx_fail(ctx, VX_ERR_STATEMENT, "verify_subchain: start_parent != trusted_header_hash");
    if (res.end_block != target_block)
        return vx_fail(ctx, VX_ERR_STATEMENT, "verify_subchain: end_block %u != target_block %u", res.end_block, target_block);
    return VX_OK;
}
}
