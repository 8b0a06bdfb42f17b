// Witness/trace generation for ShaChainAir on the GPU: one lane per trace row (block b, round r)
// recomputes the message schedule and the r rounds it needs from the block descriptor the host
// prepared (the 2n-1 compressions of the commitment chain are sequential and tiny: host).
#include <string.h>

#include "air_sha_tree.cuh"
#include "vx_internal.h"

struct ShaBlock {
    uint32_t h_in[8], block[16], dg[8], type, sgc, kc, pad;  // sgc / kc: ShaChainAir's signed flag and key counter
};
enum { SB_FIRST = 0, SB_DATA = 1, SB_PAD = 2, SB_IDLE = 3, SB_TREE = 4 /* ShaTreeAir: block kinds are periodic there, no type flags */ };

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
    // state: a, b, c, e, f, g as bits, d and h as values
    sbits(tr, n, row, A_, a), sbits(tr, n, row, B_, bb), sbits(tr, n, row, C_, c), sbits(tr, n, row, E_, e), sbits(tr, n, row, F_, f), sbits(tr, n, row, G_, g);
    tr[(size_t)DV * n + row] = d;
    tr[(size_t)HV * n + row] = h;
    sbits(tr, n, row, NA0, (uint32_t)na_full);
    sbits(tr, n, row, NE0, (uint32_t)ne_full);
    // schedule window w_r .. w_{r+15}: positions 0, 1, 14 as bits, the others as values
    sbits(tr, n, row, W0B, w[r]);
    sbits(tr, n, row, W1B, w[r + 1]);
    sbits(tr, n, row, W14B, w[r + 14]);
    for (int p = 2; p < 14; ++p) tr[(size_t)WV(p) * n + row] = w[r + p];
    tr[(size_t)WV15 * n + row] = w[r + 15];
    const uint32_t w1 = w[r + 1], w14 = w[r + 14];
    tr[(size_t)SV * n + row] = (uint64_t)(s_rotr(w1, 7) ^ s_rotr(w1, 18) ^ (w1 >> 3)) + (s_rotr(w14, 17) ^ s_rotr(w14, 19) ^ (w14 >> 10));
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
        tr[(size_t)(FFV0 + wd) * n + row] = (uint32_t)tot;
        tr[(size_t)(FFC0 + wd) * n + row] = tot >> 32;
        tr[(size_t)(HIN0 + wd) * n + row] = b.h_in[wd];
        tr[(size_t)(DG0 + wd) * n + row] = b.dg[wd];
    }
    tr[(size_t)T_FIRST * n + row] = b.type == SB_FIRST;
    tr[(size_t)T_DATA * n + row] = b.type == SB_DATA;
    tr[(size_t)T_PAD * n + row] = b.type == SB_PAD;
    tr[(size_t)T_IDLE * n + row] = b.type == SB_IDLE;
    if (b.type != SB_TREE) tr[(size_t)SGC * n + row] = b.sgc, tr[(size_t)KC * n + row] = b.kc;  // (the tree table has 412 columns)
}

// auxiliary columns of ShaChainAir: the key sends of signed blocks, one lane per row
__global__ __launch_bounds__(256) void k_sha_chain_aux(const uint64_t* tr, uint64_t* aux, size_t n, gl2 beta, gl2 gamma, uint64_t bus_on) {
    using namespace shc;
    const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (row >= n) return;
    const int r = (int)(row & 63);
    gl2 h{0, 0};
    // bus mode 1: the flagged keys are sent; mode 2: every key is received
    if (bus_on && r < 16 && !(r & 1) && (bus_on == 2 || tr[(size_t)SGC * n + row]) && tr[(size_t)(r < 8 ? T_FIRST : T_DATA) * n + row]) {
        auto limbs = [&](int col0) -> uint64_t {
            uint32_t w = 0;
            for (int i = 0; i < 32; ++i) w |= (uint32_t)tr[(size_t)(col0 + i) * n + row] << i;
            return (uint64_t)((w >> 24) | ((w >> 8) & 0xFF00)) | ((uint64_t)(((w >> 8) & 0xFF) | ((w & 0xFF) << 8)) << 16);
        };
        const gl2 g2 = gl2_mul(gamma, gamma), g4 = gl2_mul(g2, g2);
        gl2 d = gl2_add(beta, gl2_add(gl2_scale(gamma, limbs(W0B)), gl2_add(gl2_scale(g2, limbs(W1B)), gl2_scale(g4, TAG_KEY))));
        d.a = gl_add(d.a, 4 * (tr[(size_t)KC * n + row] - 1) + ((r & 7) >> 1));
        h = gl2_inv(d);
        if (bus_on == 2) h = gl2{gl_neg(h.a), gl_neg(h.b)};
    }
    aux[row] = h.a, aux[n + row] = h.b;
    aux[2 * n + row] = h.a, aux[3 * n + row] = h.b;  // increments; the scan makes them the running sum
}
int32_t vx_sha_chain_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub) {
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_sha_chain_aux, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, trace, aux, n, gl2{chal[0], chal[1]}, gl2{chal[2], chal[3]}, pub[9]);
    VX_HIP(hipGetLastError());
    return vx_bus_close_dev(ctx, aux + 2 * n, log_n, aux_pub);
}

int32_t vx_sha_chain_trace_dev(vx_ctx* ctx, const uint8_t* pubkeys, size_t n_keys, const uint8_t* signed_flags, uint64_t bus_on, int log_n, uint64_t* trace_d,
                               uint64_t public_inputs_out[10], uint8_t commitment_out[32]) {
    VX_CHECK(n_keys >= 1 && log_n >= 6 && log_n <= 24, "sha trace: bad shape");
    const size_t n = (size_t)1 << log_n, n_blocks = n >> 6;
    VX_CHECK(2 * n_keys - 1 <= n_blocks, "sha trace: %zu keys need %zu compressions, 2^%d rows hold %zu", n_keys, 2 * n_keys - 1, log_n, n_blocks);
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
            b.type = SB_FIRST, b.sgc = signed_flags && signed_flags[0], b.kc = 1;
            for (int j = 0; j < 8; ++j) b.h_in[j] = shc::IV_H[j], b.block[j] = be32(pk + 4 * j), b.block[8 + j] = shc::tail32(j);
            h_compress(b.h_in, b.block, dig);
        } else {
            ShaBlock& d = blocks[bi++];
            d.type = SB_DATA, d.sgc = signed_flags && signed_flags[i], d.kc = (uint32_t)(i + 1);
            for (int j = 0; j < 8; ++j) d.h_in[j] = shc::IV_H[j], d.block[j] = dig[j], d.block[8 + j] = be32(pk + 4 * j);
            uint32_t mid[8];
            h_compress(d.h_in, d.block, mid);
            ShaBlock& p = blocks[bi++];
            p.type = SB_PAD, p.kc = (uint32_t)(i + 1);
            for (int j = 0; j < 8; ++j) p.h_in[j] = mid[j];
            for (int j = 0; j < 16; ++j) p.block[j] = shc::pad64(j);
            h_compress(p.h_in, p.block, dig);
        }
        upd_after.push_back(bi - 1);
        dig_hist.emplace_back(dig, dig + 8);
    }
    for (; bi < n_blocks; ++bi) {
        blocks[bi].type = SB_IDLE, blocks[bi].kc = (uint32_t)n_keys;
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
    hipLaunchKernelGGL(k_sha_trace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const ShaBlock*)sc, trace_d, n);
    VX_HIP(hipGetLastError());
    VX_HIP(hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < 8; ++j) {
        public_inputs_out[j] = dig[j];
        if (commitment_out)
            for (int b = 0; b < 4; ++b) commitment_out[4 * j + b] = (uint8_t)(dig[j] >> (24 - 8 * b));
    }
    public_inputs_out[8] = n_keys, public_inputs_out[9] = bus_on;
    return VX_OK;
}
extern "C" int32_t vx_sha_chain_trace(vx_ctx* ctx, const uint8_t* pubkeys, size_t n_keys, const uint8_t* signed_flags, uint32_t bus_on, int log_n, vx_buf* trace_out,
                                      uint64_t public_inputs_out[10], uint8_t commitment_out[32]) {
    if (!ctx || !pubkeys || !trace_out || !public_inputs_out) return VX_ERR_ARG;
    VX_CHECK(log_n >= 6 && log_n <= 24 && trace_out->n >= ((size_t)shc::CHAIN_COLS << log_n), "sha trace: trace buffer too small");
    VX_CHECK(bus_on <= 2, "sha trace: bus mode %u (0 off, 1 send the flagged keys, 2 receive every key)", bus_on);
    return vx_sha_chain_trace_dev(ctx, pubkeys, n_keys, signed_flags, bus_on, log_n, trace_out->d, public_inputs_out, commitment_out);
}

// ---- ShaTreeAir: the two SHA-256 Merkle trees over state roots and data roots ------------------------------------------
// The trees are tiny (2 (N - 1) nodes): node values on the host, then the same row kernel as the chain AIR.  Block
// descriptor of node g of tree t: DATA block (IV, l || r) and PAD block (DATA's output, the constant second block of a
// 64-byte message); dg[0] / dg[1] carry the leaf-enable flags ENL / ENR of a bottom-level node, dg[2] their running count.
int32_t vx_sha_tree_trace_dev(vx_ctx* ctx, const uint8_t* state_roots, const uint8_t* data_roots, size_t n_leaves, int log_tree, uint64_t* trace_d,
                              uint64_t pub_out[17]) {
    const size_t N = (size_t)1 << log_tree, n = 256 * N, n_blocks = n >> 6;
    VX_CHECK(n_leaves >= 1 && n_leaves <= N, "sha tree: %zu leaves do not fit a tree of %zu", n_leaves, N);
    std::vector<ShaBlock> blocks(n_blocks);
    memset(blocks.data(), 0, n_blocks * sizeof(ShaBlock));
    auto be32 = [](const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; };
    std::vector<uint32_t> node(2 * N * 8);  // heap array of digests as big-endian words
    for (int t = 0; t < 2; ++t) {
        const uint8_t* leaves = t == 0 ? state_roots : data_roots;
        std::fill(node.begin(), node.end(), 0u);
        for (size_t i = 0; i < n_leaves; ++i)
            for (int j = 0; j < 8; ++j) node[(N + i) * 8 + j] = be32(leaves + 32 * i + 4 * j);
        for (size_t g = N - 1; g >= 1; --g) {
            ShaBlock& d = blocks[2 * ((size_t)t * N + g)];
            ShaBlock& p = blocks[2 * ((size_t)t * N + g) + 1];
            d.type = p.type = SB_TREE;
            for (int j = 0; j < 8; ++j) d.h_in[j] = shc::IV_H[j], d.block[j] = node[2 * g * 8 + j], d.block[8 + j] = node[(2 * g + 1) * 8 + j];
            uint32_t mid[8];
            h_compress(d.h_in, d.block, mid);
            for (int j = 0; j < 8; ++j) p.h_in[j] = mid[j];
            for (int j = 0; j < 16; ++j) p.block[j] = shc::pad64(j);
            h_compress(p.h_in, p.block, &node[g * 8]);
            if (g >= N / 2) {
                d.dg[0] = p.dg[0] = 2 * g - N < n_leaves, d.dg[1] = p.dg[1] = 2 * g - N + 1 < n_leaves;
                const size_t cnt = 2 * g - N + 2 < n_leaves ? 2 * g - N + 2 : n_leaves;  // enabled leaves up to and including this node
                d.dg[2] = p.dg[2] = (uint32_t)cnt;
            }
        }
        {  // slot 0: a dummy node (zero message), never on the bus
            ShaBlock& d = blocks[2 * (size_t)t * N];
            ShaBlock& p = blocks[2 * (size_t)t * N + 1];
            d.type = p.type = SB_TREE;
            for (int j = 0; j < 8; ++j) d.h_in[j] = shc::IV_H[j];
            uint32_t mid[8];
            h_compress(d.h_in, d.block, mid);
            for (int j = 0; j < 8; ++j) p.h_in[j] = mid[j];
            for (int j = 0; j < 16; ++j) p.block[j] = shc::pad64(j);
        }
        for (int j = 0; j < 8; ++j) pub_out[8 * t + j] = node[8 + j];
    }
    pub_out[16] = n_leaves;
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, (n_blocks * sizeof(ShaBlock) + 7) / 8, &sc));
    VX_HIP(hipMemcpyAsync(sc, blocks.data(), n_blocks * sizeof(ShaBlock), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_sha_trace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const ShaBlock*)sc, trace_d, n);
    VX_HIP(hipGetLastError());
    VX_HIP(hipStreamSynchronize(ctx->stream));  // blocks (host vector) must outlive the copy
    return VX_OK;
}

// auxiliary columns of ShaTreeAir: one lane per row; the positional (periodic) quantities are recomputed from the row index
__global__ __launch_bounds__(256) void k_sha_tree_aux(const uint64_t* tr, uint64_t* aux, size_t n, size_t N, gl2 beta, gl2 gamma) {
    using namespace shc;
    const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (row >= n) return;
    const size_t r = row & 63, bk = (row >> 6) & 1, pair = row >> 7, tree = pair / N, g = pair % N;
    const bool msg = bk == 0 && r < 16, bottom = g >= N / 2, inner = g >= 1 && g < N / 2, send = bk == 1 && r == 63 && g >= 2;
    const uint64_t en = msg ? tr[(size_t)(r < 8 ? sht::ENL : sht::ENR) * n + row] : 0;
    const uint64_t m_word = msg && inner ? 1 : 0, m_byte = msg && bottom && en ? 1 : 0;  // inner nodes take words, the bottom level of both trees bytes
    auto word = [&](int col0, int nb) -> uint64_t {
        uint64_t v = 0;
        for (int i = 0; i < nb; ++i) v |= tr[(size_t)(col0 + i) * n + row] << i;
        return v;
    };
    gl2 h[7], hsum{0, 0};
    for (int e = 0; e < 7; ++e) h[e] = gl2{0, 0};
    if (m_word | m_byte | (send ? 1 : 0)) {
        const gl2 g2 = gl2_mul(gamma, gamma), g3 = gl2_mul(g2, gamma), g4 = gl2_mul(g2, g2);
        const gl2 tag_w = gl2_scale(g4, blk::TAG_WORD), tag_b = gl2_scale(g4, blk::TAG_BYTE);
        const uint64_t c = r >= 8 ? 1 : 0, jj = r & 7, w0 = (msg ? word(W0B, 32) : 0);
        const uint64_t cid = bottom ? 2 * g - N + c : 2 * g + c;  // a leaf's index / an inner child's node id
        gl2 d[13];
        uint64_t m[13];  // 1 = receive (-1), 2 = send (+1), 0 = inactive
        d[0] = gl2_add(gl2_add(beta, gl2{(uint64_t)tree, 0}), gl2_add(gl2_add(gl2_scale(gamma, cid), gl2_scale(g2, jj)), gl2_add(gl2_scale(g3, w0), tag_w)));
        m[0] = m_word;
        for (int q = 0; q < 4; ++q) {
            const uint64_t byte = (w0 >> (24 - 8 * q)) & 0xFF;
            d[1 + q] = gl2_add(gl2_add(beta, gl2{cid, 0}), gl2_add(gl2_add(gl2_scale(gamma, 4 * jj + q), gl2_scale(g2, byte)), gl2_add(gl2_scale(g3, tree), tag_b)));
            m[1 + q] = m_byte;
        }
        for (int j = 0; j < 8; ++j) {
            const uint64_t ff = send ? tr[(size_t)(FFV0 + j) * n + row] : 0;
            d[5 + j] = gl2_add(gl2_add(beta, gl2{(uint64_t)tree, 0}), gl2_add(gl2_add(gl2_scale(gamma, g), gl2_scale(g2, j)), gl2_add(gl2_scale(g3, ff), tag_w)));
            m[5 + j] = send ? 2 : 0;
        }
        auto term = [&](int q) -> gl2 {  // m / D with m in {-1, 0, +1}
            if (!m[q]) return gl2{0, 0};
            const gl2 iv = gl2_inv(d[q]);
            return m[q] == 2 ? iv : gl2{gl_neg(iv.a), gl_neg(iv.b)};
        };
        for (int e = 0; e < 7; ++e) {
            h[e] = term(2 * e);
            if (e < 6) h[e] = gl2_add(h[e], term(2 * e + 1));
            hsum = gl2_add(hsum, h[e]);
        }
    }
    for (int e = 0; e < 7; ++e) aux[(size_t)(2 * e) * n + row] = h[e].a, aux[(size_t)(2 * e + 1) * n + row] = h[e].b;
    aux[(size_t)14 * n + row] = hsum.a, aux[(size_t)15 * n + row] = hsum.b;  // increments; the scan makes them the running sum
}
static int32_t sha_tree_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, size_t N, const uint64_t* chal, uint64_t* aux, uint64_t* aux_pub) {
    const size_t n = (size_t)1 << log_n;
    VX_CHECK(n == 256 * N, "sha tree aux: a tree of %zu leaves has %zu rows, not 2^%d", N, 256 * N, log_n);
    hipLaunchKernelGGL(k_sha_tree_aux, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, trace, aux, n, N, gl2{chal[0], chal[1]}, gl2{chal[2], chal[3]});
    VX_HIP(hipGetLastError());
    return vx_bus_close_dev(ctx, aux + 14 * n, log_n, aux_pub);
}
int32_t vx_sha_tree_gen_aux_16(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t*, uint64_t* aux, uint64_t* aux_pub) {
    return sha_tree_gen_aux(ctx, trace, log_n, 16, chal, aux, aux_pub);
}
int32_t vx_sha_tree_gen_aux_256(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t*, uint64_t* aux, uint64_t* aux_pub) {
    return sha_tree_gen_aux(ctx, trace, log_n, 256, chal, aux, aux_pub);
}
int32_t vx_sha_tree_gen_aux_512(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t*, uint64_t* aux, uint64_t* aux_pub) {
    return sha_tree_gen_aux(ctx, trace, log_n, 512, chal, aux, aux_pub);
}
