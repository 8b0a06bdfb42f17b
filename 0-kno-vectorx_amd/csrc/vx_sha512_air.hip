// Witness generation for Sha512Air (air_sha512.cuh): H_i = SHA-512(R_i || A_i || M), the hash half of the conditional EdDSA
// verifications (/root/reference circuits/builder/justification.rs:229-243).
//   k_s512_slots  one lane per slot: both compressions of the slot, natively (mid state, digest)
//   k_s512_rows   one lane per trace row: recomputes the message schedule and the rounds it needs from the slot record and
//                 writes its 1055 cells (column-major: the lanes of a wave write neighbouring rows)
//   k_s512_aux    one lane per row: the bus helper of receive / send rows, the running-sum increments
#include <string.h>

#include "air_sha512.cuh"
#include "vx_internal.h"

namespace {
using namespace s5;
struct S5Slot {
    uint64_t blk1[16], mid[8], out[8];
    uint32_t sg, pad;
};
__device__ const uint64_t K5[80] = SHA512_K_INIT;
__device__ __forceinline__ uint64_t rr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
__device__ __forceinline__ uint64_t sg0(uint64_t x) { return rr(x, 1) ^ rr(x, 8) ^ (x >> 7); }
__device__ __forceinline__ uint64_t sg1(uint64_t x) { return rr(x, 19) ^ rr(x, 61) ^ (x >> 6); }
__device__ void compress(const uint64_t* h_in, const uint64_t* block, uint64_t* out) {
    uint64_t w[80], s[8];
    for (int i = 0; i < 16; ++i) w[i] = block[i];
    for (int t = 16; t < 80; ++t) w[t] = w[t - 16] + sg0(w[t - 15]) + w[t - 7] + sg1(w[t - 2]);
    for (int i = 0; i < 8; ++i) s[i] = h_in[i];
    for (int r = 0; r < 80; ++r) {
        const uint64_t t1 = s[7] + (rr(s[4], 14) ^ rr(s[4], 18) ^ rr(s[4], 41)) + ((s[4] & s[5]) ^ (~s[4] & s[6])) + K5[r] + w[r];
        const uint64_t t2 = (rr(s[0], 28) ^ rr(s[0], 34) ^ rr(s[0], 39)) + ((s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]));
        s[7] = s[6], s[6] = s[5], s[5] = s[4], s[4] = s[3] + t1, s[3] = s[2], s[2] = s[1], s[1] = s[0], s[0] = t1 + t2;
    }
    for (int i = 0; i < 8; ++i) out[i] = h_in[i] + s[i];
}
// slot s < k hashes for authority idx[s] (the slots are compact, in the order of EdAir's); the others hash an all-zero R || A
__global__ __launch_bounds__(64) void k_s512_slots(const uint8_t* pubkeys, const uint8_t* sigs, const uint8_t* msg, const uint32_t* idx, size_t k_active, size_t m, S5Slot* out) {
    const size_t s = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (s >= m) return;
    S5Slot& o = out[s];
    const bool on = s < k_active;
    uint8_t buf[128];
    for (int j = 0; j < 128; ++j) buf[j] = 0;
    if (on)
        for (int j = 0; j < 32; ++j) buf[j] = sigs[64 * (size_t)idx[s] + j], buf[32 + j] = pubkeys[32 * (size_t)idx[s] + j];
    for (int j = 0; j < MSG_LEN; ++j) buf[64 + j] = msg[j];
    buf[64 + MSG_LEN] = 0x80;
    for (int i = 0; i < 16; ++i) {
        uint64_t v = 0;
        for (int b = 0; b < 8; ++b) v = (v << 8) | buf[8 * i + b];
        o.blk1[i] = v;
    }
    uint64_t pad[16];
    for (int p = 0; p < 16; ++p) pad[p] = pad2(p);
    compress(IV, o.blk1, o.mid);
    compress(o.mid, pad, o.out);
    o.sg = on, o.pad = 0;
}
__device__ __forceinline__ void put_bits(uint64_t* tr, size_t n, size_t row, int col0, uint64_t v, int nb = 64) {
    for (int i = 0; i < nb; ++i) tr[(size_t)(col0 + i) * n + row] = (v >> i) & 1;
}
__device__ __forceinline__ void put_halves(uint64_t* tr, size_t n, size_t row, int col, uint64_t v) {
    tr[(size_t)col * n + row] = v & 0xFFFFFFFFULL, tr[(size_t)(col + 1) * n + row] = v >> 32;
}
// sum of the 32-bit halves of `cnt` 64-bit terms: (low-half carry, high-half carry, the 64-bit result)
__device__ __forceinline__ void add_halves(const uint64_t* t, int cnt, uint64_t& c_lo, uint64_t& c_hi, uint64_t& res) {
    uint64_t lo = 0, hi = 0;
    for (int i = 0; i < cnt; ++i) lo += t[i] & 0xFFFFFFFFULL, hi += t[i] >> 32;
    c_lo = lo >> 32, hi += c_lo, c_hi = hi >> 32;
    res = (lo & 0xFFFFFFFFULL) | (hi << 32);
}
__global__ __launch_bounds__(256) void k_s512_rows(const S5Slot* slots, size_t m, uint64_t* tr, size_t n) {
    const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (row >= n) return;
    const size_t slot = row / SLOT_ROWS;
    const int q = (int)(row % SLOT_ROWS);
    if (slot >= m) {
        for (int col = 0; col < COLS; ++col) tr[(size_t)col * n + row] = 0;
        return;
    }
    const S5Slot& sl = slots[slot];
    tr[(size_t)SGF * n + row] = sl.sg;
    const int blk = q / 80, r = q % 80;
    uint64_t w[96], s[8], h_in[8];
    for (int i = 0; i < 16; ++i) w[i] = blk ? pad2(i) : sl.blk1[i];
    for (int t = 16; t < 80; ++t) w[t] = w[t - 16] + sg0(w[t - 15]) + w[t - 7] + sg1(w[t - 2]);
    for (int t = 80; t < 96; ++t) w[t] = 0;
    for (int i = 0; i < 8; ++i) h_in[i] = s[i] = blk ? sl.mid[i] : IV[i];
    for (int t = 0; t < r; ++t) {
        const uint64_t t1 = s[7] + (rr(s[4], 14) ^ rr(s[4], 18) ^ rr(s[4], 41)) + ((s[4] & s[5]) ^ (~s[4] & s[6])) + K5[t] + w[t];
        const uint64_t t2 = (rr(s[0], 28) ^ rr(s[0], 34) ^ rr(s[0], 39)) + ((s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]));
        s[7] = s[6], s[6] = s[5], s[5] = s[4], s[4] = s[3] + t1, s[3] = s[2], s[2] = s[1], s[1] = s[0], s[0] = t1 + t2;
    }
    const uint64_t a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
    const uint64_t e1 = rr(e, 14) ^ rr(e, 18) ^ rr(e, 41), a0 = rr(a, 28) ^ rr(a, 34) ^ rr(a, 39), ch = (e & f) ^ (~e & g), mj = (a & b) ^ (a & c) ^ (b & c);
    uint64_t ne, na, ce_lo, ce_hi, ca_lo, ca_hi;
    {
        const uint64_t te[6] = {d, h, e1, ch, K5[r], w[r]}, ta[7] = {h, e1, ch, K5[r], w[r], a0, mj};
        add_halves(te, 6, ce_lo, ce_hi, ne);
        add_halves(ta, 7, ca_lo, ca_hi, na);
    }
    put_bits(tr, n, row, A_, a), put_bits(tr, n, row, B_, b), put_bits(tr, n, row, C_, c), put_bits(tr, n, row, E_, e), put_bits(tr, n, row, F_, f), put_bits(tr, n, row, G_, g);
    put_halves(tr, n, row, DV, d), put_halves(tr, n, row, HV, h);
    put_bits(tr, n, row, NA0, na), put_bits(tr, n, row, NE0, ne);
    put_bits(tr, n, row, W0B, w[r]), put_bits(tr, n, row, W1B, w[r + 1]), put_bits(tr, n, row, W14B, w[r + 14]);
    for (int p = 2; p < 14; ++p) put_halves(tr, n, row, WV(p), w[r + p]);
    put_halves(tr, n, row, WV15, w[r + 15]);
    const uint64_t w1 = w[r + 1], w14 = w[r + 14];
    {
        const uint64_t s0 = sg0(w1), s1 = sg1(w14);
        tr[(size_t)SV * n + row] = (s0 & 0xFFFFFFFFULL) + (s1 & 0xFFFFFFFFULL), tr[(size_t)(SV + 1) * n + row] = (s0 >> 32) + (s1 >> 32);
    }
    put_bits(tr, n, row, CE0, ce_lo, 3), put_bits(tr, n, row, CE0 + 3, ce_hi, 3), put_bits(tr, n, row, CA0, ca_lo, 3), put_bits(tr, n, row, CA0 + 3, ca_hi, 3);
    uint64_t cw_lo = 0, cw_hi = 0;
    if (r <= 63) {
        const uint64_t tw[4] = {sg1(w14), w[r + 9], sg0(w1), w[r]};
        uint64_t res;
        add_halves(tw, 4, cw_lo, cw_hi, res);
    }
    put_bits(tr, n, row, CW0, cw_lo, 2), put_bits(tr, n, row, CW0 + 2, cw_hi, 2);
    const uint64_t s80[8] = {na, a, b, c, ne, e, f, g};
    for (int wd = 0; wd < 8; ++wd) {
        uint64_t ff = 0, c_lo = 0, c_hi = 0;
        if (r == 79) {
            const uint64_t t2[2] = {h_in[wd], s80[wd]};
            add_halves(t2, 2, c_lo, c_hi, ff);
        } else if (blk && r >= 74) ff = sl.out[wd];  // the digest is held (and sent) from row 74 of block 2 on
        put_halves(tr, n, row, FFV0 + 2 * wd, ff);
        tr[(size_t)(FFC0 + 2 * wd) * n + row] = c_lo, tr[(size_t)(FFC0 + 2 * wd + 1) * n + row] = c_hi;
        put_halves(tr, n, row, HIN0 + 2 * wd, h_in[wd]);
    }
}

__global__ __launch_bounds__(256) void k_s512_aux(const uint64_t* tr, uint64_t* aux, size_t n, size_t m, gl2 beta, gl2 gamma, uint64_t bus_on) {
    const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (row >= n) return;
    const size_t slot = row / SLOT_ROWS;
    const int q = (int)(row % SLOT_ROWS);
    const bool rcv = slot < m && q < 8 && !(q & 1), snd = slot < m && q >= SEND0;
    gl2 h{0, 0};
    if ((rcv || snd) && bus_on && tr[(size_t)SGF * n + row]) {
        uint64_t t0, t1, t2, t3;
        if (rcv) {
            auto word = [&](int col0) -> uint64_t {
                uint64_t v = 0;
                for (int i = 0; i < 64; ++i) v |= tr[(size_t)(col0 + i) * n + row] << i;
                return v;
            };
            const uint64_t w[2] = {word(W0B), word(W1B)};
            uint64_t l[8];
            for (int j = 0; j < 8; ++j) {  // limb j of a word: bytes 2j, 2j+1 of its big-endian byte string, little-endian
                const uint64_t x = w[j >> 2];
                const int k = j & 3;
                l[j] = ((x >> (56 - 16 * k)) & 0xFF) | (((x >> (48 - 16 * k)) & 0xFF) << 8);
            }
            t0 = 4 * slot + q / 2, t1 = l[0] | (l[1] << 16) | (l[2] << 32), t2 = l[3] | (l[4] << 16) | (l[5] << 32), t3 = l[6] | (l[7] << 16);
        } else {
            const int j = q - SEND0;
            auto ff = [&](int k) -> uint64_t { return k < 16 ? tr[(size_t)(FFV0 + k) * n + row] : 0; };
            t0 = 8 * slot + j, t1 = ff(3 * j), t2 = ff(3 * j + 1), t3 = ff(3 * j + 2);
        }
        const gl2 g2 = gl2_mul(gamma, gamma), g3 = gl2_mul(g2, gamma), g4 = gl2_mul(g2, g2);
        gl2 d = gl2_add(beta, gl2_add(gl2_scale(gamma, t1), gl2_add(gl2_scale(g2, t2), gl2_add(gl2_scale(g3, t3), gl2_scale(g4, snd ? edc::TAG_EDH : edc::TAG_EDMSG)))));
        d.a = gl_add(d.a, t0);
        h = gl2_inv(d);
        if (rcv) h = gl2{gl_neg(h.a), gl_neg(h.b)};
    }
    aux[row] = h.a, aux[n + row] = h.b;
    aux[2 * n + row] = h.a, aux[3 * n + row] = h.b;  // increments; the scan makes them the running sum
}
}  // namespace

int32_t vx_sha512_air_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub) {
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_s512_aux, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, trace, aux, n, n / SLOT_ROWS, gl2{chal[0], chal[1]}, gl2{chal[2], chal[3]}, pub[14]);
    VX_HIP(hipGetLastError());
    return vx_bus_close_dev(ctx, aux + 2 * n, log_n, aux_pub);
}

int32_t vx_sha512_trace_dev(vx_ctx* ctx, const uint8_t* pubkeys, const uint8_t* sigs, const uint8_t* msg, const uint8_t* flags, size_t n_sigs, int log_n, uint64_t bus_on,
                            uint64_t* trace_d, uint64_t pub_out[15]) {
    const size_t n = (size_t)1 << log_n, m = n / SLOT_ROWS;
    VX_CHECK(log_n >= 8 && log_n <= 20, "sha512 trace: log_n %d out of range [8, 20]", log_n);
    std::vector<uint32_t> idx;  // compact slots, in EdAir's order: slot s = the s-th flagged authority
    for (size_t s = 0; s < n_sigs; ++s)
        if (flags[s]) idx.push_back((uint32_t)s);
    const size_t k = idx.size();
    VX_CHECK(k <= m, "sha512 trace: %zu signatures do not fit the %zu slots of 2^%d rows", k, m, log_n);
    const size_t w_keys = 4 * n_sigs + 1, w_sigs = 8 * n_sigs + 1, w_msg = 8, w_idx = k / 2 + 1, w_slots = (m * sizeof(S5Slot) + 7) / 8;
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, w_keys + w_sigs + w_msg + w_idx + w_slots, &sc));
    uint8_t* d_keys = (uint8_t*)sc;
    uint8_t* d_sigs = (uint8_t*)(sc + w_keys);
    uint8_t* d_msg = (uint8_t*)(sc + w_keys + w_sigs);
    uint32_t* d_idx = (uint32_t*)(sc + w_keys + w_sigs + w_msg);
    S5Slot* d_slots = (S5Slot*)(sc + w_keys + w_sigs + w_msg + w_idx);
    if (n_sigs) {
        VX_HIP(hipMemcpyAsync(d_keys, pubkeys, 32 * n_sigs, hipMemcpyHostToDevice, ctx->stream));
        VX_HIP(hipMemcpyAsync(d_sigs, sigs, 64 * n_sigs, hipMemcpyHostToDevice, ctx->stream));
    }
    if (k) VX_HIP(hipMemcpyAsync(d_idx, idx.data(), k * 4, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemcpyAsync(d_msg, msg, MSG_LEN, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_s512_slots, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, ctx->stream, d_keys, d_sigs, d_msg, (const uint32_t*)d_idx, k, m, d_slots);
    hipLaunchKernelGGL(k_s512_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const S5Slot*)d_slots, m, trace_d, n);
    VX_HIP(hipGetLastError());
    VX_HIP(hipStreamSynchronize(ctx->stream));  // idx (host vector) must outlive the copy
    uint8_t tail[64];
    memset(tail, 0, sizeof tail);
    memcpy(tail, msg, MSG_LEN);
    tail[MSG_LEN] = 0x80;
    for (int j = 0; j < 7; ++j) {
        uint64_t v = 0;
        for (int b = 0; b < 8; ++b) v = (v << 8) | tail[8 * j + b];
        pub_out[2 * j] = v & 0xFFFFFFFFULL, pub_out[2 * j + 1] = v >> 32;
    }
    pub_out[14] = bus_on;
    return VX_OK;
}

extern "C" int32_t vx_sha512_trace(vx_ctx* ctx, const uint8_t* pubkeys, const uint8_t* signatures, const uint8_t* message, uint32_t message_len, const uint8_t* signed_flags,
                                   size_t n_signatures, int log_n, uint32_t bus_on, vx_buf* trace_out, uint64_t public_inputs_out[15]) {
    if (!ctx || !message || !trace_out || !public_inputs_out || (n_signatures && (!pubkeys || !signatures || !signed_flags))) return VX_ERR_ARG;
    VX_CHECK(message_len == MSG_LEN, "sha512 trace: the message must be the 53-byte precommit, not %u bytes", message_len);
    VX_CHECK(log_n >= 8 && log_n <= 20 && trace_out->n >= ((size_t)COLS << log_n), "sha512 trace: trace buffer holds %zu elements, 2^%d rows need %zu", trace_out->n, log_n,
             (size_t)COLS << log_n);
    return vx_sha512_trace_dev(ctx, pubkeys, signatures, message, signed_flags, n_signatures, log_n, bus_on ? 1 : 0, trace_out->d, public_inputs_out);
}
