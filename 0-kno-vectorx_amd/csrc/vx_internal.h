// libvxprove internals: context, device buffers, host-side field helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

#pragma GCC visibility push(default)
#include "../../include/vx.h"
#pragma GCC visibility pop

// ---- host-side Goldilocks (table generation, transcript) ----------------------------
namespace glh {
static const uint64_t P = 0xFFFFFFFF00000001ULL;
static inline uint64_t add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    return (s < a || s >= P) ? s - P : s;
}
static inline uint64_t sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (P - b); }
// x mod p for any 128-bit x, without the 128-bit division `% P` compiles to (__umodti3, ~30 ns -- the host transcript absorbs
// thousands of elements per proof): x = lo + hl 2^64 + hh 2^96 = lo - hh + hl (2^32 - 1) (mod p), 2^64 = 2^32 - 1, 2^96 = -1.
static inline uint64_t reduce128(unsigned __int128 x) {
    const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64), hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
    uint64_t t0, t2;
    if (__builtin_sub_overflow(lo, hh, &t0)) t0 -= 0xFFFFFFFFULL;  // + p (mod 2^64); t0 >= 2^64 - 2^32 here, cannot wrap again
    if (__builtin_add_overflow(t0, hl * 0xFFFFFFFFULL, &t2)) t2 += 0xFFFFFFFFULL;  // - p (mod 2^64); the wrapped sum is < 2^64 - 2^32
    return t2 >= P ? t2 - P : t2;
}
static inline uint64_t mul(uint64_t a, uint64_t b) { return reduce128((unsigned __int128)a * b); }
static inline uint64_t pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mul(r, a);
        a = mul(a, a);
        e >>= 1;
    }
    return r;
}
static inline uint64_t inv(uint64_t a) { return pow(a, P - 2); }
static const uint64_t ROOT_2_32 = 1753635133440165772ULL;  // 7^((p-1)/2^32)
static inline uint64_t root(int log_n) {
    uint64_t r = ROOT_2_32;
    for (int i = 32; i > log_n; --i) r = mul(r, r);
    return r;
}
}  // namespace glh

struct vx_buf {
    uint64_t* d;
    size_t n;
};

// three-level power table of a base b: lvl[0][j] = b^j, lvl[1][j] = b^(j<<11), lvl[2][j] = b^(j<<22), j < 2048
struct PowTab {
    uint64_t* d;  // 3*2048 on device
};

struct vx_ctx {
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    std::string err;
    PowTab tw_fwd, tw_inv;       // base = omega_{2^32}, omega_{2^32}^-1
    uint64_t *w12_fwd, *w12_inv;  // omega_4096^e, e < 2048
    std::map<uint64_t, PowTab> shift_tabs;
    std::map<int, uint64_t*> tw2;  // key = log_s * 2 + inverse: two-level table of w_{2^log_s}
    uint64_t* scratch;
    size_t scratch_n;
    void* pinned;  // small pinned staging area
    size_t pinned_n;
    // device memory pool: blocks are recycled by exact (rounded) size instead of hipFree'd -- a
    // proof allocates tens of GB and hipMalloc/hipFree of such blocks costs far more than the kernels.
    // All work is on ctx->stream, so a recycled block is safe to hand out again immediately.
    std::map<size_t, std::vector<void*>> pool_free;
    std::map<void*, size_t> pool_live;
    // a second context on the same device (own stream, scratch and pool), made on first use: independent small proofs
    // (the authority-set commitment STARKs) run on it from a host thread while this context proves the hash chain
    vx_ctx* side = nullptr;
    // periodic columns of an AIR on the LDE coset, by (air id, degree bits, rate bits): identical for every proof of a shape
    std::map<uint64_t, uint64_t*> periodic_cache;
};
vx_ctx* vx_side_ctx(vx_ctx* ctx);  // nullptr if it cannot be created
void* vx_pool_alloc(vx_ctx* ctx, size_t bytes);
void vx_pool_free(vx_ctx* ctx, void* p);
void vx_pool_trim(vx_ctx* ctx);

struct vx_tree {
    uint64_t* levels;  // level 0 (leaf digests, 4*n) followed by each parent level up to the cap
    size_t n_leaves;
    int cap_height;
    size_t total;  // uint64 count
};

int32_t vx_fail(vx_ctx* ctx, int32_t code, const char* fmt, ...);
#define VX_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return vx_fail(ctx, e_ == hipErrorOutOfMemory ? VX_ERR_OOM : VX_ERR_DEVICE, "%s: %s", \
                           #call, hipGetErrorString(e_));                                         \
    } while (0)
#define VX_CHECK(cond, ...)                                   \
    do {                                                      \
        if (!(cond)) return vx_fail(ctx, VX_ERR_ARG, __VA_ARGS__); \
    } while (0)
#define VX_TRY(call)            \
    do {                        \
        int32_t r_ = (call);    \
        if (r_ != VX_OK) return r_; \
    } while (0)

int32_t vx_get_shift_tab(vx_ctx* ctx, uint64_t base, PowTab* out);
int32_t vx_scratch(vx_ctx* ctx, size_t n_u64, uint64_t** out);
int32_t vx_merkle_build_dev(vx_ctx* ctx, const uint64_t* data, size_t n_leaves, size_t leaf_len, int layout,
                            int cap_height, vx_tree** out);

int32_t vx_ntt_dev(vx_ctx* ctx, uint64_t* d, int log_n, size_t n_cols, size_t col_stride, int inverse, uint64_t shift, int order);
int32_t vx_lde_dev(vx_ctx* ctx, const uint64_t* src, int log_n, size_t n_cols, int rate_bits, uint64_t shift, int src_kind,
                   uint64_t* dst, uint64_t* coeffs_out);
int32_t vx_gather_rows_dev(vx_ctx* ctx, const uint64_t* lde, int log_N, size_t n_cols, const uint64_t* leaf_idx, size_t n_idx,
                           uint64_t* out);
int32_t vx_fri_fold_dev(vx_ctx* ctx, const uint64_t* evals, int log_n, int arity_bits, const uint64_t beta[2], uint64_t shift,
                        uint64_t* out);
int32_t vx_fri_layer_tree_dev(vx_ctx* ctx, const uint64_t* evals, int log_n, int arity_bits, int cap_height, vx_tree** out);
int32_t vx_fri_leaves_dev(vx_ctx* ctx, const uint64_t* evals, int log_n, int arity_bits, const uint64_t* leaf_idx, size_t n_idx,
                          uint64_t* out);

// two-level root table for one sub-transform size: lo[j] = w_S^j (j < 2^lo_bits), hi[j] = w_S^(j << lo_bits)
struct Tw2 {
    const uint64_t* lo;
    const uint64_t* hi;
    int lo_bits;
};
int32_t vx_get_tw2(vx_ctx* ctx, int log_s, int inverse, Tw2* out);
int32_t vx_lde_consume_dev(vx_ctx* ctx, uint64_t* values, int log_n, size_t n_cols, int rate_bits, uint64_t shift, uint64_t* dst);
// Called by the prover once the trace cap of an AIR with an auxiliary round is known; fills `chal` (n_chal values) with
// lookup challenges that depend on EVERY table sharing the bus (typically by proving the other table from inside).
struct vx_chal_hook {
    int32_t (*fn)(void* user, const uint64_t* pub, size_t n_pub, const uint64_t* cap, size_t cap_words, uint64_t* chal, size_t n_chal);
    void* user;
};
int32_t vx_stark_prove_impl(vx_ctx* ctx, int air_id, const vx_stark_config* cfg, uint64_t* trace_d, size_t trace_len, int consume_trace,
                            int log_n, const uint64_t* public_inputs, size_t n_public, uint64_t* proof_out, size_t proof_cap,
                            size_t* proof_len, const vx_chal_hook* hook = nullptr);
void vx_shared_challenges_n(const uint64_t* const* pubs, const size_t* n_pubs, const uint64_t* const* caps, size_t k, size_t cap_words, uint64_t* out, size_t n_out);
void vx_shared_challenges(const uint64_t* pub_a, size_t n_a, const uint64_t* cap_a, const uint64_t* pub_b, size_t n_b, const uint64_t* cap_b,
                          size_t cap_words, uint64_t* out, size_t n_out);
// verifier with externally derived lookup challenges (nullptr = drawn from the proof's own transcript); apub_out (optional)
// receives a pointer to the values published with the auxiliary cap, and log_n_out the degree bits
int32_t vx_stark_verify_ext(const vx_stark_config* cfg, const uint64_t* proof, size_t proof_len, int expect_air, const uint64_t* expect_public,
                            size_t n_expect_public, const uint64_t* ext_chal, const uint64_t** apub_out, int* log_n_out, char* err, size_t errlen);
// (public inputs, trace cap) of a serialised proof, for deriving shared challenges; false if the proof is too short
bool vx_stark_proof_peek(const uint64_t* proof, size_t len, int cap_height, const uint64_t** pub, size_t* n_pub, const uint64_t** cap);
int32_t vx_lde_keep_dev(vx_ctx* ctx, const uint64_t* values, int log_n, size_t n_cols, int rate_bits, uint64_t shift, uint64_t* coef_brev,
                        uint64_t* dst);
int32_t vx_scan_cols_dev(vx_ctx* ctx, uint64_t* data, int log_n, size_t n_cols, uint64_t* totals_host);
int32_t vx_blake_air_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub);
int32_t vx_lookup_air_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub);
int32_t vx_sha_tree_trace_dev(vx_ctx* ctx, const uint8_t* state_roots, const uint8_t* data_roots, size_t n_leaves, int log_tree, uint64_t* trace_d,
                              uint64_t pub_out[17]);
int32_t vx_bus_close_dev(vx_ctx* ctx, uint64_t* z_cols, int log_n, uint64_t aux_pub[2]);
int32_t vx_sha_tree_gen_aux_16(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub);
int32_t vx_sha_tree_gen_aux_256(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub);
int32_t vx_sha_tree_gen_aux_512(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub);
int32_t vx_ed_air_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub);
int32_t vx_ed_trace_dev(vx_ctx* ctx, const uint8_t* pubkeys, const uint8_t* sigs, const uint8_t* msg, uint32_t msg_len, const uint8_t* signed_flags, size_t n_sigs,
                        int log_n, uint64_t bus_on, uint64_t* trace_d, uint64_t pub_out[2]);
int32_t vx_sha512_air_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub);
int32_t vx_sha512_trace_dev(vx_ctx* ctx, const uint8_t* pubkeys, const uint8_t* sigs, const uint8_t* msg, const uint8_t* flags, size_t n_sigs, int log_n, uint64_t bus_on,
                            uint64_t* trace_d, uint64_t pub_out[15]);
int32_t vx_epoch_air_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub);
int32_t vx_epoch_end_trace_dev(vx_ctx* ctx, const uint8_t* header_d, size_t header_bytes, uint32_t start_position, uint32_t num_authorities, uint64_t bus_on, uint64_t* trace_d,
                               uint64_t pub_out[10], uint32_t* window_length_out);
int32_t vx_sha_chain_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub);
int32_t vx_sha_chain_trace_dev(vx_ctx* ctx, const uint8_t* pubkeys, size_t n_keys, const uint8_t* signed_flags, uint64_t bus_on, int log_n, uint64_t* trace_d,
                               uint64_t public_inputs_out[10], uint8_t commitment_out[32]);
void vx_merkle_levels_launch(vx_ctx* ctx, uint64_t* levels, size_t n_leaves, size_t cap);
