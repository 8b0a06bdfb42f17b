// Witness/trace generation for ShaChainAir on the GPU: one lane per trace row (block b, round r)
// recomputes the message schedule and the r rounds it needs from the block descriptor the host
// prepared (the 2n-1 compressions of the commitment chain are sequential and tiny: host).
#include <string.h>

#include "air_sha.cuh"
#include "vx_internal.h"

struct ShaBlock {
    uint32_t h_in[8], block[16], dg[8], type, pad[3];
};
enum { SB_FIRST = 0, SB_DATA = 1, SB_PAD = 2, SB_IDLE = 3 };

__host__ __device__ static inline uint32_t s_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void h_compress(const uint32_t* h_in, const uint32_t* block, uint32_t* out) {
    uint32_t w[64], s[8];
    for (int i = 0; i < 16; ++i) w[i] = block[i];
    for (int t = 16; t < 64; ++t)
        w[t] = w[t - 16] + (s_rotr(w[t - 15], 7) ^ s_rotr(w[t - 15], 18) ^ (w[t - 15] >> 3)) + w[t - 7] + (s_rotr(w[t - 2], 17) ^ s_rotr(w[t - 2], 19) ^ (w[t - 2] >> 10));
    memcpy(s, h_in, sizeof s);
    for (int r = 0; r < 64; ++r) {
        const uint32_t t1 = s[7] + (s_rotr(s[4], 6) ^ s_rotr(s[4], 11) ^ s_rotr(s[4], 25)) + ((s[4] & s[5]) ^ (~s[4] & s[6])) + shc::K_H[r] + w[r];
        const uint32_t t2 = (s_rotr(s[0], 2) ^ s_rotr(s[0], 13) ^ s_rotr(s[0], 22)) + ((s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]));
        s[7] = s[6]; s[6] = s[5]; s[5] = s[4]; s[4] = s[3] + t1; s[3] = s[2]; s[2] = s[1]; s[1] = s[0]; s[0] = t1 + t2;
    }
    for (int i = 0; i < 8; ++i) out[i] = h_in[i] + s[i];
}

__device__ __forceinline__ void sbits(uint64_t* tr, size_t n, size_t row, int col0, uint64_t v, int nb = 32) {
    for (int i = 0; i < nb; ++i) tr[(size_t)(col0 + i) * n + row] = (v >> i) & 1;
}
__device__ __forceinline__ void sxor3(uint64_t* tr, size_t n, size_t row, uint32_t x, uint32_t y, uint32_t z, int colr, int colc) {
    for (int i = 0; i < 32; ++i) {
        const uint32_t s = ((x >> i) & 1) + ((y >> i) & 1) + ((z >> i) & 1);
        tr[(size_t)(colr + i) * n + row] = s & 1;
        tr[(size_t)(colc + i) * n + row] = s >> 1;
    }
}

__global__ __launch_bounds__(256) void k_sha_trace(const ShaBlock* blocks, uint64_t* tr, size_t n) {
    using namespace shc;
    const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (row >= n) return;
    const ShaBlock b = blocks[row >> 6];
    const int r = (int)(row & 63);
    uint32_t w[80];
    for (int i = 0; i < 16; ++i) w[i] = b.block[i];
    for (int t = 16; t < 64; ++t)
        w[t] = w[t - 16] + (s_rotr(w[t - 15], 7) ^ s_rotr(w[t - 15], 18) ^ (w[t - 15] >> 3)) + w[t - 7] + (s_rotr(w[t - 2], 17) ^ s_rotr(w[t - 2], 19) ^ (w[t - 2] >> 10));
    for (int t = 64; t < 80; ++t) w[t] = 0;
    uint32_t s[8];
    for (int i = 0; i < 8; ++i) s[i] = b.h_in[i];
    for (int q = 0; q < r; ++q) {
        const uint32_t t1 = s[7] + (s_rotr(s[4], 6) ^ s_rotr(s[4], 11) ^ s_rotr(s[4], 25)) + ((s[4] & s[5]) ^ (~s[4] & s[6])) + K[q] + w[q];
        const uint32_t t2 = (s_rotr(s[0], 2) ^ s_rotr(s[0], 13) ^ s_rotr(s[0], 22)) + ((s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]));
        s[7] = s[6]; s[6] = s[5]; s[5] = s[4]; s[4] = s[3] + t1; s[3] = s[2]; s[2] = s[1]; s[1] = s[0]; s[0] = t1 + t2;
    }
    const uint32_t a = s[0], bb = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
    const uint32_t e1 = s_rotr(e, 6) ^ s_rotr(e, 11) ^ s_rotr(e, 25), a0 = s_rotr(a, 2) ^ s_rotr(a, 13) ^ s_rotr(a, 22);
    const uint32_t ch = (e & f) ^ (~e & g), mj = (a & bb) ^ (a & c) ^ (bb & c);
    const uint64_t t1 = (uint64_t)h + e1 + ch + K[r] + w[r];
    const uint64_t ne_full = (uint64_t)d + t1, na_full = t1 + a0 + mj;
    for (int wd = 0; wd < 8; ++wd) sbits(tr, n, row, ST(wd, 0), s[wd]);
    sbits(tr, n, row, NA0, (uint32_t)na_full);
    sbits(tr, n, row, NE0, (uint32_t)ne_full);
    for (int j = 0; j < 16; ++j) sbits(tr, n, row, WW(j, 0), w[r + j]);
    const uint32_t w1 = w[r + 1], w14 = w[r + 14];
    sxor3(tr, n, row, s_rotr(w1, 7), s_rotr(w1, 18), w1 >> 3, S0R, S0C);
    sxor3(tr, n, row, s_rotr(w14, 17), s_rotr(w14, 19), w14 >> 10, S1R, S1C);
    sxor3(tr, n, row, s_rotr(e, 6), s_rotr(e, 11), s_rotr(e, 25), E1R, E1C);
    sxor3(tr, n, row, s_rotr(a, 2), s_rotr(a, 13), s_rotr(a, 22), A0R, A0C);
    for (int i = 0; i < 32; ++i) {
        const uint32_t sm = ((a >> i) & 1) + ((bb >> i) & 1) + ((c >> i) & 1);
        tr[(size_t)(MAJ + i) * n + row] = sm >> 1;
        tr[(size_t)(PAR + i) * n + row] = sm & 1;
    }
    sbits(tr, n, row, CE0, ne_full >> 32, 3);
    sbits(tr, n, row, CA0, na_full >> 32, 3);
    uint64_t cw = 0;
    if (r <= 47) {
        const uint32_t s0 = s_rotr(w1, 7) ^ s_rotr(w1, 18) ^ (w1 >> 3), s1 = s_rotr(w14, 17) ^ s_rotr(w14, 19) ^ (w14 >> 10);
        cw = ((uint64_t)s1 + w[r + 9] + s0 + w[r]) >> 32;
    }
    sbits(tr, n, row, CW0, cw, 2);
    for (int wd = 0; wd < 8; ++wd) {
        uint64_t tot = 0;
        if (r == 63) {
            const uint32_t s64[8] = {(uint32_t)na_full, a, bb, c, (uint32_t)ne_full, e, f, g};
            tot = (uint64_t)b.h_in[wd] + s64[wd];
        }
        sbits(tr, n, row, FFB(wd, 0), (uint32_t)tot);
        tr[(size_t)(FFC0 + wd) * n + row] = tot >> 32;
        tr[(size_t)(HIN0 + wd) * n + row] = b.h_in[wd];
        tr[(size_t)(DG0 + wd) * n + row] = b.dg[wd];
    }
    tr[(size_t)T_FIRST * n + row] = b.type == SB_FIRST;
    tr[(size_t)T_DATA * n + row] = b.type == SB_DATA;
    tr[(size_t)T_PAD * n + row] = b.type == SB_PAD;
    tr[(size_t)T_IDLE * n + row] = b.type == SB_IDLE;
}

extern "C" {
int32_t vx_sha_chain_trace(vx_ctx* ctx, const uint8_t* pubkeys, size_t n_keys, int log_n, vx_buf* trace_out,
                           uint64_t public_inputs_out[8], uint8_t commitment_out[32]) {
    if (!ctx || !pubkeys || !trace_out || !public_inputs_out) return VX_ERR_ARG;
    VX_CHECK(n_keys >= 1 && log_n >= 6 && log_n <= 24, "sha trace: bad shape");
    const size_t n = (size_t)1 << log_n, n_blocks = n >> 6;
    VX_CHECK(2 * n_keys - 1 <= n_blocks, "sha trace: %zu keys need %zu compressions, 2^%d rows hold %zu", n_keys, 2 * n_keys - 1, log_n, n_blocks);
    VX_CHECK(trace_out->n >= n * (size_t)shc::COLS, "sha trace: trace buffer too small");
    std::vector<ShaBlock> blocks(n_blocks);
    memset(blocks.data(), 0, n_blocks * sizeof(ShaBlock));
    auto be32 = [](const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; };
    uint32_t dig[8];
    size_t bi = 0;
    std::vector<size_t> upd_after;  // index of the block after which the digest register changes, with the new value
    std::vector<std::vector<uint32_t>> dig_hist;
    for (size_t i = 0; i < n_keys; ++i) {
        const uint8_t* pk = pubkeys + 32 * i;
        if (i == 0) {
            ShaBlock& b = blocks[bi++];
            b.type = SB_FIRST;
            for (int j = 0; j < 8; ++j) b.h_in[j] = shc::IV_H[j], b.block[j] = be32(pk + 4 * j), b.block[8 + j] = shc::tail32(j);
            h_compress(b.h_in, b.block, dig);
        } else {
            ShaBlock& d = blocks[bi++];
            d.type = SB_DATA;
            for (int j = 0; j < 8; ++j) d.h_in[j] = shc::IV_H[j], d.block[j] = dig[j], d.block[8 + j] = be32(pk + 4 * j);
            uint32_t mid[8];
            h_compress(d.h_in, d.block, mid);
            ShaBlock& p = blocks[bi++];
            p.type = SB_PAD;
            for (int j = 0; j < 8; ++j) p.h_in[j] = mid[j];
            for (int j = 0; j < 16; ++j) p.block[j] = shc::pad64(j);
            h_compress(p.h_in, p.block, dig);
        }
        upd_after.push_back(bi - 1);
        dig_hist.emplace_back(dig, dig + 8);
    }
    for (; bi < n_blocks; ++bi) {
        blocks[bi].type = SB_IDLE;
        for (int j = 0; j < 8; ++j) blocks[bi].h_in[j] = shc::IV_H[j];
    }
    // digest register per block: the final digest in block 0 (cyclic wrap), then the running value
    {
        std::vector<uint32_t> cur(dig, dig + 8);
        size_t k = 0;
        for (size_t b = 0; b < n_blocks; ++b) {
            for (int j = 0; j < 8; ++j) blocks[b].dg[j] = cur[j];
            if (k < upd_after.size() && upd_after[k] == b) cur = dig_hist[k++];
        }
    }
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, (n_blocks * sizeof(ShaBlock) + 7) / 8, &sc));
    VX_HIP(hipMemcpyAsync(sc, blocks.data(), n_blocks * sizeof(ShaBlock), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_sha_trace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const ShaBlock*)sc, trace_out->d, n);
    VX_HIP(hipGetLastError());
    VX_HIP(hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < 8; ++j) {
        public_inputs_out[j] = dig[j];
        if (commitment_out)
            for (int b = 0; b < 4; ++b) commitment_out[4 * j + b] = (uint8_t)(dig[j] >> (24 - 8 * b));
    }
    return VX_OK;
}
}
