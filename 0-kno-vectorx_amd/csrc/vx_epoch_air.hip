// Witness of EpochEndAir (air_epoch.cuh): the ScheduledChange log of the epoch-end header, one row per authority record.
//   k_epoch_rows  one lane per row (512): copies the row's bytes out of the header on the device, sets the row flags and the
//                 bits of the first length byte
//   k_epoch_aux   one lane per row: the 22 bus helpers (40 byte receives, 4 key sends, two lookups each) and the running-sum
//                 increments
// Parity: tests/test_gpu_epoch_air.py compares trace, auxiliary columns and public inputs with oracle/epoch_air.py.
#include <string.h>

#include "air_epoch.cuh"
#include "vx_internal.h"

namespace {
using namespace epo;

__global__ __launch_bounds__(64) void k_epoch_rows(const uint8_t* __restrict__ header, uint32_t log_pos, uint32_t plen, uint32_t n_auth, uint64_t* __restrict__ tr) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << LOG_N;
    if (row >= n) return;
    const uint8_t* p = header + log_pos;  // the consensus flag
    const bool pre = row == 0, val = row >= 1 && row <= n_auth, dly = row == n_auth + 1;
    const uint32_t base = pre ? 0 : plen + 40 * (row - 1), cnt = pre ? plen : val ? 40 : dly ? 4 : 0;
    for (uint32_t j = 0; j < NB; ++j) tr[(size_t)j * n + row] = j < cnt ? p[base + j] : 0;
    tr[(size_t)V * n + row] = val, tr[(size_t)DL * n + row] = dly;
    for (int i = 0; i < 6; ++i) tr[(size_t)(Q0 + i) * n + row] = pre ? ((p[5] >> 2) >> i) & 1 : 0;
}

__global__ __launch_bounds__(64) void k_epoch_aux(const uint64_t* __restrict__ tr, uint64_t* __restrict__ aux, uint32_t plen, gl2 beta, gl2 gamma, uint64_t bus_on) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << LOG_N;
    if (row >= n) return;
    const bool pre = row == 0, val = tr[(size_t)V * n + row] != 0, dly = tr[(size_t)DL * n + row] != 0;
    const gl2 g2 = gl2_mul(gamma, gamma), g3 = gl2_mul(g2, gamma), g4 = gl2_mul(g2, g2);
    const gl2 bbase = gl2_add(beta, gl2_add(g3, gl2_scale(g4, blk::TAG_BYTE)));
    const uint64_t kbase = pre ? 0 : (uint64_t)plen + 40 * (uint64_t)(row - 1);
    auto cell = [&](int j) -> uint64_t { return tr[(size_t)j * n + row]; };
    gl2 hsum{0, 0};
    auto pair = [&](int e, bool mu, bool mv, const gl2& du, const gl2& dv, bool neg) {
        gl2 h{0, 0};
        if (bus_on && (mu || mv)) {
            h = gl2_mul(gl2_add(mu ? dv : gl2{0, 0}, mv ? du : gl2{0, 0}), gl2_inv(gl2_mul(du, dv)));
            if (neg) h = gl2{gl_neg(h.a), gl_neg(h.b)};
        }
        aux[(size_t)(2 * e) * n + row] = h.a, aux[(size_t)(2 * e + 1) * n + row] = h.b;
        hsum = gl2_add(hsum, h);
    };
    auto d_byte = [&](int j) -> gl2 { return gl2_add(bbase, gl2_add(gl2_scale(gamma, gl_add(kbase, (uint64_t)j)), gl2_scale(g2, cell(j)))); };
    auto m_byte = [&](int j) -> bool { return val || (pre && (uint32_t)j < plen) || (dly && j < 4); };
#pragma unroll 1
    for (int e = 0; e < 20; ++e) pair(e, m_byte(2 * e), m_byte(2 * e + 1), d_byte(2 * e), d_byte(2 * e + 1), true);
    auto d_key = [&](int q) -> gl2 {
        const uint64_t la = cell(8 * q) | (cell(8 * q + 1) << 8) | (cell(8 * q + 2) << 16) | (cell(8 * q + 3) << 24);
        const uint64_t lb = cell(8 * q + 4) | (cell(8 * q + 5) << 8) | (cell(8 * q + 6) << 16) | (cell(8 * q + 7) << 24);
        gl2 d = gl2_add(beta, gl2_add(gl2_scale(gamma, la), gl2_add(gl2_scale(g2, lb), gl2_scale(g4, edc::TAG_KEY))));
        d.a = gl_add(d.a, gl_add(gl_mul(gl_sub((uint64_t)row, 1), 4), (uint64_t)q));  // (row - 1) * 4 + q in the field (row 0: multiplicity 0)
        return d;
    };
#pragma unroll 1
    for (int e = 0; e < 2; ++e) pair(20 + e, val, val, d_key(2 * e), d_key(2 * e + 1), false);
    aux[(size_t)(2 * (N_HELP - 1)) * n + row] = hsum.a, aux[(size_t)(2 * (N_HELP - 1) + 1) * n + row] = hsum.b;  // increments; the scan makes them the running sum
}
}  // namespace

int32_t vx_epoch_air_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub) {
    const size_t n = (size_t)1 << log_n;
    uint32_t plen = 6;
    for (int a = 0; a < 4; ++a) plen += (uint32_t)(pub[2 + a] + pub[6 + a]) * (uint32_t)len_of(a);
    hipLaunchKernelGGL(k_epoch_aux, dim3((unsigned)(n / 64)), dim3(64), 0, ctx->stream, trace, aux, plen, gl2{chal[0], chal[1]}, gl2{chal[2], chal[3]}, pub[1]);
    VX_HIP(hipGetLastError());
    return vx_bus_close_dev(ctx, aux + 2 * (N_HELP - 1) * n, log_n, aux_pub);
}

// header_d: the header bytes on the device (header_bytes of them readable).  The prefix is parsed here (rotate.rs:74-174: its two
// compact lengths are public inputs of the table); the validator records are copied as they are -- a header whose records do not
// satisfy rotate.rs:243-274 yields a trace no proof exists for (vx_verify_epoch_end_header names the failing rule).
int32_t vx_epoch_end_trace_dev(vx_ctx* ctx, const uint8_t* header_d, size_t header_bytes, uint32_t start_position, uint32_t num_authorities, uint64_t bus_on, uint64_t* trace_d,
                               uint64_t pub_out[10], uint32_t* window_length_out) {
    const uint32_t n = 1u << LOG_N;
    VX_CHECK(num_authorities >= 1 && num_authorities <= n - 2, "epoch-end trace: %u authorities (the table holds %u)", num_authorities, n - 2);
    VX_CHECK((uint64_t)start_position + 1 + 17 <= header_bytes, "epoch-end trace: start position %u leaves no room for the 17-byte prefix", start_position);
    uint8_t p[17];
    VX_HIP(hipMemcpyAsync(p, header_d + start_position + 1, 17, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    if (p[0] != 4 || memcmp(p + 1, "FRNK", 4) != 0) return vx_fail(ctx, VX_ERR_STATEMENT, "epoch-end trace: consensus flag / engine id");
    const int m1 = p[5] & 3, len1 = len_of(m1);
    if (m1 == 3 && p[5] != 3) return vx_fail(ctx, VX_ERR_STATEMENT, "epoch-end trace: compact int (length)");
    if (p[5 + len1] != 1) return vx_fail(ctx, VX_ERR_STATEMENT, "epoch-end trace: scheduled change flag");
    const uint8_t* e = p + 6 + len1;
    const int m2 = e[0] & 3, len2 = len_of(m2);
    const uint32_t w2 = (uint32_t)e[0] | ((uint32_t)e[1] << 8), w4 = w2 | ((uint32_t)e[2] << 16) | ((uint32_t)e[3] << 24);
    const uint32_t val = m2 == 0 ? e[0] >> 2 : m2 == 1 ? w2 >> 2 : m2 == 2 ? w4 >> 2 : ((uint32_t)e[1] | ((uint32_t)e[2] << 8) | ((uint32_t)e[3] << 16) | ((uint32_t)e[4] << 24));
    if (val != num_authorities || (m2 == 3 && e[0] != 3)) return vx_fail(ctx, VX_ERR_STATEMENT, "epoch-end trace: authority count");
    const uint32_t plen = 6 + len1 + len2, wlen = plen + 40 * num_authorities + 4;
    VX_CHECK((uint64_t)start_position + 1 + wlen <= header_bytes, "epoch-end trace: the log of %u authorities runs past the header buffer", num_authorities);
    hipLaunchKernelGGL(k_epoch_rows, dim3(n / 64), dim3(64), 0, ctx->stream, header_d, start_position + 1, plen, num_authorities, trace_d);
    VX_HIP(hipGetLastError());
    pub_out[0] = num_authorities, pub_out[1] = bus_on ? 1 : 0;
    for (int a = 0; a < 4; ++a) pub_out[2 + a] = a == m1, pub_out[6 + a] = a == m2;
    if (window_length_out) *window_length_out = wlen;
    return VX_OK;
}

extern "C" int32_t vx_epoch_end_trace(vx_ctx* ctx, const vx_buf* header, uint32_t start_position, uint32_t num_authorities, uint32_t bus_on, vx_buf* trace_out,
                                      uint64_t public_inputs_out[10], uint32_t* window_length_out) {
    if (!ctx || !header || !trace_out || !public_inputs_out) return VX_ERR_ARG;
    VX_CHECK(trace_out->n >= ((size_t)COLS << LOG_N), "epoch-end trace: trace buffer holds %zu elements, 2^%d rows need %zu", trace_out->n, LOG_N, (size_t)COLS << LOG_N);
    return vx_epoch_end_trace_dev(ctx, (const uint8_t*)header->d, header->n * 8, start_position, num_authorities, bus_on, trace_out->d, public_inputs_out, window_length_out);
}
