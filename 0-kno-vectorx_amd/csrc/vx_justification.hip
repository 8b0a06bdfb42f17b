// verify_simple_justification on the GPU (statement level): the native counterpart of
// /root/reference circuits/builder/justification.rs:195-257 and of the hint's own checks
// (justification.rs:29-83 -> circuits/input/mod.rs:241-260):
//   1. authority-set commitment  = chained SHA-256 of the public keys      (justification.rs:127-162)
//   2. precommit decoding        = byte 0 == 1, hash / number / set id     (decoder.rs:159-200)
//   3. Ed25519 verification of every signed vote over the SAME 53-byte precommit
//      (justification.rs:229-243; ed25519-dalek `verify`: [s]B - [k]A compressed == R)
//   4. threshold                 = signed * 3 > n * 2                       (justification.rs:164-186)
// One lane per signature; GF(2^255-19) as 8 x 32-bit limbs with plain schoolbook products and the
// 2^256 = 38 fold -- 300 signatures per proof are far off the hot path, so the code is written
// for obviousness, not speed.  (The in-circuit EdDSA / SHA-256 AIRs are future work: DESIGN.md.)
#include <string.h>

#include "ed25519_constants.h"
#include "vx_internal.h"

#include "ed25519.cuh"
using namespace ed;

// ok[i] = 1 iff signature i verifies (ed25519-dalek `verify`: s canonical, A decodes,
// compress([s]B - [k]A) == R bytes), k = SHA-512(R || A || M) mod L
__global__ __launch_bounds__(64) void k_ed25519_verify(const uint8_t* pubkeys, const uint8_t* sigs, const uint8_t* msg, uint32_t msg_len,
                                                       const uint8_t* enabled, size_t n, uint8_t* ok) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!enabled[i]) {
        ok[i] = 2;  // not signed: skipped (justification.rs:59-62)
        return;
    }
    const uint8_t* pk = pubkeys + 32 * i;
    const uint8_t* sg = sigs + 64 * i;
    uint32_t s[8];
    for (int j = 0; j < 8; ++j) s[j] = (uint32_t)sg[32 + 4 * j] | ((uint32_t)sg[33 + 4 * j] << 8) | ((uint32_t)sg[34 + 4 * j] << 16) | ((uint32_t)sg[35 + 4 * j] << 24);
    Pt A;
    if (u256_geq(s, SC_L) || !pt_decode(pk, &A)) {
        ok[i] = 0;
        return;
    }
    uint8_t buf[64 + 128], dig[64];
    for (int j = 0; j < 32; ++j) buf[j] = sg[j], buf[32 + j] = pk[j];
    for (uint32_t j = 0; j < msg_len; ++j) buf[64 + j] = msg[j];
    sha512(buf, 64 + msg_len, dig);
    uint32_t k[8];
    sc_reduce512(dig, k);
    // -A
    U256 zero;
    for (int j = 0; j < 8; ++j) zero.w[j] = 0;
    Pt nA = {fe_sub(zero, A.X), A.Y, A.Z, fe_sub(zero, A.T)};
    U256 one;
    for (int j = 0; j < 8; ++j) one.w[j] = j == 0;
    const U256 bx = fe_from(ED_BX), by = fe_from(ED_BY);
    const Pt B = {bx, by, one, fe_mul(bx, by)};
    const Pt R = pt_double_scalar_mul(B, s, nA, k);
    uint8_t enc[32];
    pt_encode(R, enc);
    bool eq = true;
    for (int j = 0; j < 32; ++j) eq &= enc[j] == sg[j];
    ok[i] = eq ? 1 : 0;
}
// host SHA-256 for the authority-set commitment: 300 dependent 64-byte hashes are ~30 us of scalar
// work on the host and ~30 ms on a single GPU lane, so this one stays on the host
static void h_sha256(const uint8_t* msg, size_t len, uint8_t* out32) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    auto rr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t buf[128];
    const size_t total = (len + 9 <= 64) ? 64 : 128;
    memset(buf, 0, sizeof buf);
    memcpy(buf, msg, len);
    buf[len] = 0x80;
    const uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; ++i) buf[total - 1 - i] = (uint8_t)(bits >> (8 * i));
    for (size_t off = 0; off < total; off += 64) {
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)buf[off + 4 * i] << 24) | ((uint32_t)buf[off + 4 * i + 1] << 16) | ((uint32_t)buf[off + 4 * i + 2] << 8) | buf[off + 4 * i + 3];
        for (int i = 16; i < 64; ++i) w[i] = w[i - 16] + (rr(w[i - 15], 7) ^ rr(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (rr(w[i - 2], 17) ^ rr(w[i - 2], 19) ^ (w[i - 2] >> 10));
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            const uint32_t t1 = hh + (rr(e, 6) ^ rr(e, 11) ^ rr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rr(a, 2) ^ rr(a, 13) ^ rr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    for (int i = 0; i < 8; ++i)
        for (int b = 0; b < 4; ++b) out32[4 * i + b] = (uint8_t)(h[i] >> (24 - 8 * b));
}

extern "C" {

int32_t vx_ed25519_verify_batch(vx_ctx* ctx, const uint8_t* pubkeys, const uint8_t* sigs, const uint8_t* msg, uint32_t msg_len,
                                const uint8_t* enabled, size_t n, uint8_t* ok_out) {
    if (!ctx || !pubkeys || !sigs || !msg || !enabled || !ok_out) return VX_ERR_ARG;
    VX_CHECK(msg_len <= 128 && n >= 1 && n <= 65536, "ed25519 batch: bad sizes");
    uint64_t* sc;
    const size_t bytes = 32 * n + 64 * n + 128 + n + n;
    VX_TRY(vx_scratch(ctx, (bytes + 7) / 8 + 8, &sc));
    uint8_t* d = (uint8_t*)sc;
    uint8_t *d_pk = d, *d_sg = d + 32 * n, *d_msg = d_sg + 64 * n, *d_en = d_msg + 128, *d_ok = d_en + n;
    VX_HIP(hipMemcpyAsync(d_pk, pubkeys, 32 * n, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemcpyAsync(d_sg, sigs, 64 * n, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemcpyAsync(d_msg, msg, msg_len, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemcpyAsync(d_en, enabled, n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_ed25519_verify, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (const uint8_t*)d_pk, (const uint8_t*)d_sg,
                       (const uint8_t*)d_msg, msg_len, (const uint8_t*)d_en, n, d_ok);
    VX_HIP(hipGetLastError());
    VX_HIP(hipMemcpyAsync(ok_out, d_ok, n, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}

int32_t vx_verify_simple_justification(vx_ctx* ctx, uint32_t block_number, const uint8_t block_hash[32], uint64_t authority_set_id,
                                       const uint8_t authority_set_hash[32], const uint8_t precommit[53], const uint8_t* pubkeys,
                                       const uint8_t* signatures, const uint8_t* validator_signed, uint32_t num_authorities,
                                       uint32_t max_authorities) {
    if (!ctx || !block_hash || !authority_set_hash || !precommit || !pubkeys || !signatures || !validator_signed) return VX_ERR_ARG;
    VX_CHECK(max_authorities >= 1 && max_authorities <= 4096 && num_authorities <= max_authorities, "justification: %u authorities of max %u", num_authorities, max_authorities);
    // justification.rs:134-137: at least one authority
    if (num_authorities == 0) return vx_fail(ctx, VX_ERR_STATEMENT, "justification: no authorities");
    // 1. authority-set commitment over the first num_authorities keys (sequential chain: host)
    uint8_t commit[32], buf[64];
    h_sha256(pubkeys, 32, commit);
    for (uint32_t i = 1; i < num_authorities; ++i) {
        memcpy(buf, commit, 32);
        memcpy(buf + 32, pubkeys + 32 * (size_t)i, 32);
        h_sha256(buf, 64, commit);
    }
    if (memcmp(commit, authority_set_hash, 32) != 0) return vx_fail(ctx, VX_ERR_STATEMENT, "justification: authority set commitment mismatch");
    // 2. precommit (decoder.rs:159-200)
    if (precommit[0] != 1) return vx_fail(ctx, VX_ERR_STATEMENT, "justification: precommit type byte is %u", precommit[0]);
    uint32_t bn = 0;
    uint64_t sid = 0;
    for (int i = 0; i < 4; ++i) bn |= (uint32_t)precommit[33 + i] << (8 * i);
    for (int i = 0; i < 8; ++i) sid |= (uint64_t)precommit[45 + i] << (8 * i);
    if (bn != block_number || sid != authority_set_id || memcmp(precommit + 1, block_hash, 32) != 0)
        return vx_fail(ctx, VX_ERR_STATEMENT, "justification: precommit does not match block number / set id / block hash");
    // 3. signatures of the validators marked as signed (only the first num_authorities can count)
    std::vector<uint8_t> en(max_authorities), ok(max_authorities);
    uint32_t n_signed = 0;
    for (uint32_t i = 0; i < max_authorities; ++i) {
        en[i] = validator_signed[i] ? 1 : 0;
        n_signed += en[i];
    }
    VX_TRY(vx_ed25519_verify_batch(ctx, pubkeys, signatures, precommit, 53, en.data(), max_authorities, ok.data()));
    for (uint32_t i = 0; i < max_authorities; ++i)
        if (en[i] && ok[i] != 1) return vx_fail(ctx, VX_ERR_STATEMENT, "justification: signature %u is invalid", i);
    // 4. threshold: signed * 3 > n * 2
    if (!((uint64_t)n_signed * 3 > (uint64_t)num_authorities * 2))
        return vx_fail(ctx, VX_ERR_STATEMENT, "justification: %u of %u signed, not more than 2/3", n_signed, num_authorities);
    return VX_OK;
}
}
