// K8: witness/trace generation for BlakeChainAir on the GPU, and the header_range prove entry.
//   kernel A  k_blake_chain: one lane per header -- digest + the chaining value before every
//             128-byte chunk (the only sequential part of BLAKE2b);
//   kernel B  k_blake_trace: one lane per TRACE ROW (block b, r = row mod 16): recomputes
//             the <= 12 rounds it needs from the chunk's chaining value and writes its 4822
//             cells; lanes of a wave write 64 consecutive rows of a column, so every store
//             instruction is a coalesced 512-byte segment of the column-major trace.
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

__device__ __forceinline__ void put_bits(uint64_t* tr, size_t n, size_t row, int col0, uint64_t val, int nbits = 64) {
    for (int i = 0; i < nbits; ++i) tr[(size_t)(col0 + i) * n + row] = (val >> i) & 1;
}

// The 4096 bit columns of the G area and the 64 of MB0 are not stored bit by bit from the row lane: a lane would
// walk 4337 columns 8n bytes apart, one 512-byte store each, and the address translation of that walk -- not the
// bytes -- set the kernel's time (22 ms for 18 GB).  k_blake_trace writes the 65 WORDS per row instead
// (words[wc * n + row]) and k_expand_bits turns each word column into its 64 bit columns, a block writing 16 KB
// runs of 64 columns only.
constexpr int N_WORD_COLS = 65, EXP_RPL = 8;
__global__ __launch_bounds__(256) void k_expand_bits(const uint64_t* __restrict__ words, uint64_t* __restrict__ tr, size_t n) {
    const int wc = blockIdx.y;
    const size_t col0 = wc < 64 ? (size_t)wc * 64 : (size_t)blk::MB0;
    const size_t row0 = (size_t)blockIdx.x * (256 * EXP_RPL) + threadIdx.x;
    uint64_t w[EXP_RPL];
#pragma unroll
    for (int rr = 0; rr < EXP_RPL; ++rr) {
        const size_t row = row0 + 256 * (size_t)rr;
        w[rr] = row < n ? words[(size_t)wc * n + row] : 0;
    }
    for (int i = 0; i < 64; ++i) {
        uint64_t* c = tr + (col0 + i) * n;
#pragma unroll
        for (int rr = 0; rr < EXP_RPL; ++rr) {
            const size_t row = row0 + 256 * (size_t)rr;
            if (row < n) c[row] = (w[rr] >> i) & 1;
        }
    }
}

// one lane per trace row
__global__ __launch_bounds__(256) void k_blake_trace(const uint8_t* msgs, const BlockDesc* descs, const uint64_t* hchain, size_t n_real,
                                                     uint64_t* tr, uint64_t* __restrict__ words, size_t n) {
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
    } else {  // padding block: the 32-byte message D
        for (int k = 0; k < 8; ++k) h[k] = IV[k];
        h[0] ^= 0x01010020ULL;
        for (int k = 0; k < 16; ++k) m[k] = k < 4 ? ((uint64_t)d.D[2 * k] | ((uint64_t)d.D[2 * k + 1] << 32)) : 0;
        m[4] = 4ULL * d.num + 2;  // bytes 32..36: SCALE compact (4-byte mode) of the last block number
    }
    // ---- G area (columns 0 .. 4159): zero unless this row uses it
    GRec rec[8];
    bool have_rec = false;
    uint64_t vfin[16];
    blake_init_v(v, h, d.t, d.fin);
    uint64_t v0[16];
    for (int k = 0; k < 16; ++k) v0[k] = v[k];
    if (r >= 1) {
        const int last = r <= 12 ? r - 1 : 11;  // rounds 0 .. last
        for (int q = 0; q <= last; ++q) blake_round(v, m, q, q == last && r <= 12 ? rec : nullptr);
        have_rec = r <= 12;
        for (int k = 0; k < 16; ++k) vfin[k] = v[k];
    }
    uint64_t h_out[8];
    if (r >= 13)
        for (int k = 0; k < 8; ++k) h_out[k] = h[k] ^ vfin[k] ^ vfin[k + 8];
    auto put_word = [&](int bit_col0, uint64_t val) { words[(size_t)(bit_col0 >> 6) * n + row] = val; };  // G-area cells are word aligned
    if (have_rec) {
        for (int k = 0; k < 8; ++k) {
            for (int w = 0; w < 8; ++w) put_word(GB(k, w, 0), rec[k].w[w]);
            for (int j = 0; j < 8; ++j) tr[(size_t)CAR(k, j) * n + row] = rec[k].car[j];
        }
    } else {
        for (int wc = 0; wc < 64; ++wc) words[(size_t)wc * n + row] = 0;
        for (int col = 4096; col < 4160; ++col) tr[(size_t)col * n + row] = 0;
        if (r == 0) {  // (a lane's later store to the same address wins)
            for (int w = 0; w < 16; ++w) put_word(OUT(w), v0[w]);
        } else if (r == 13) {
            for (int w = 0; w < 8; ++w) {
                put_word(FT(w, 0), h[w] ^ vfin[w]);
                put_word(FV(w, 0), vfin[8 + w]);
                put_word(FH(w, 0), h[w]);
            }
        } else if (r == 14) {
            for (int w = 0; w < 8; ++w) put_word(FT(w, 0), h_out[w]);
        }
    }
    // ---- message schedule + range check of natural word r
    for (int s = 0; s < 16; ++s) {
        const uint64_t w = m[ORDER[r][s]];
        tr[(size_t)MS(s, 0) * n + row] = w & 0xFFFFFFFFULL;
        tr[(size_t)MS(s, 1) * n + row] = w >> 32;
    }
    words[(size_t)64 * n + row] = m[r];  // MB0 .. MB0+63
    for (int b = 0; b < 8; ++b) tr[(size_t)(MK0 + b) * n + row] = (uint32_t)(8 * r + b) < d.inc ? 1 : 0;
    tr[(size_t)CNT * n + row] = d.inc < (uint32_t)(8 * (r + 1)) ? d.inc : (uint32_t)(8 * (r + 1));
    // ---- H register
    for (int w = 0; w < 8; ++w) {
        const uint64_t hv = r <= 13 ? h[w] : (r == 14 ? h_out[w] : (d.fin ? (w == 0 ? IV[0] ^ 0x01010020ULL : IV[w]) : h_out[w]));
        tr[(size_t)HL(w, 0) * n + row] = hv & 0xFFFFFFFFULL;
        tr[(size_t)HL(w, 1) * n + row] = hv >> 32;
    }
    // ---- digest register, flags, counters
    const bool cap = d.act && d.fin;
    for (int j = 0; j < 8; ++j) {
        uint32_t dv = d.D[j];
        if (r == 15 && cap) dv = (uint32_t)(h_out[j / 2] >> (32 * (j & 1)));
        tr[(size_t)(D0 + j) * n + row] = dv;
    }
    tr[(size_t)ACT * n + row] = d.act;
    tr[(size_t)FIN * n + row] = d.fin;
    tr[(size_t)FIRST * n + row] = d.first;
    tr[(size_t)CAP * n + row] = cap ? 1 : 0;
    tr[(size_t)T * n + row] = d.t;
    tr[(size_t)INC * n + row] = d.inc;
    tr[(size_t)NUM * n + row] = d.num;
    tr[(size_t)FA * n + row] = (d.first && d.act) ? 1 : 0;
    put_bits(tr, n, row, TB0, d.t, 32);
    put_bits(tr, n, row, IB0, d.inc, 8);
}

extern "C" {

int32_t vx_blake_chain_trace(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes, size_t n_headers,
                             const uint8_t trusted_hash[32], uint32_t first_block_number, int log_n, vx_buf* trace_out,
                             uint64_t public_inputs_out[18], uint8_t* digests_out) {
    if (!ctx || !headers || !sizes || !trusted_hash || !trace_out || !public_inputs_out) return VX_ERR_ARG;
    VX_CHECK(stride % 128 == 0 && stride > 0, "blake trace: stride %zu must be a positive multiple of 128", stride);
    VX_CHECK(n_headers >= 1 && n_headers * stride <= headers->n * 8, "blake trace: headers exceed the buffer");
    VX_CHECK(log_n >= 4 && log_n <= 24, "blake trace: log_n %d out of range", log_n);
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
    const size_t w_words = (size_t)N_WORD_COLS * n;
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, 2 * w_sizes + w_dig + w_hc + w_desc + w_words, &sc));
    uint32_t* d_sizes = (uint32_t*)sc;
    uint32_t* d_base = (uint32_t*)(sc + w_sizes);
    uint8_t* d_dig = (uint8_t*)(sc + 2 * w_sizes);
    uint64_t* d_hc = sc + 2 * w_sizes + w_dig;
    BlockDesc* d_desc = (BlockDesc*)(d_hc + w_hc);
    uint64_t* d_words = d_hc + w_hc + w_desc;
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
    hipLaunchKernelGGL(k_blake_trace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint8_t*)headers->d,
                       (const BlockDesc*)d_desc, (const uint64_t*)d_hc, n_real, trace_out->d, d_words, n);
    VX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_expand_bits, dim3((unsigned)((n + 256 * EXP_RPL - 1) / (256 * EXP_RPL)), N_WORD_COLS), dim3(256), 0, ctx->stream,
                       (const uint64_t*)d_words, trace_out->d, n);
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

static const uint64_t VX_HR_MAGIC = 0x3245474e41525248ULL;  // "HRRANGE2"
static const size_t VX_HR_HDR = 18;  // magic, max_headers, trusted, target, out96 (12), len(blake proof), len(sha proof)

static int sha_log_n(size_t n_keys) {
    int log_n = 6;
    while (((size_t)1 << log_n) < 64 * (2 * n_keys - 1)) ++log_n;
    return log_n;
}

int32_t vx_header_range_proof_bound(const vx_stark_config* cfg, size_t n_chunks, size_t n_authorities, size_t* n_words) {
    if (!cfg || !n_words || n_chunks == 0) return VX_ERR_ARG;
    int log_n = 4;
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
    int log_n = 4;
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
