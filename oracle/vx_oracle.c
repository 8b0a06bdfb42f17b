/*
 * oracle/vx_oracle.c -- TEST INFRASTRUCTURE: CPU restatement ("oracle") of the
 * header_range proving path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (0-kno-vectorx_amd/) never links or calls it.
 *
 * What it restates and where that is specified:
 *   - Statement-level semantics (in-tree, /root/reference):
 *       circuits/builder/subchain_verification.rs:81-303  map/reduce chain rules
 *       circuits/builder/decoder.rs:39-200                SCALE / header / precommit layouts
 *       circuits/builder/justification.rs:127-186         authority-set commitment, threshold
 *       circuits/input/mod.rs:250-290, 464-528            native mirrors (set hash, precommit, Merkle root)
 *       circuits/dummy_header_range.rs:11-52              96-byte public output
 *   - STARK backend primitives (NOT in tree; third-party plonky2 v0.2.0,
 *     git 0xPolygonZero/plonky2 #7445ec91, Cargo.lock:4848-4905): restated from
 *     the published algorithms of that crate -- field/src/fft.rs (radix-2 FFT,
 *     natural order in/out), plonky2/src/fri/oracle.rs (LDE = zero-pad + coset
 *     FFT with shift g=7, transposed, bit-reversed leaves), hash/poseidon.rs,
 *     hash/hashing.rs (sponge, rate 8, overwrite mode), hash/merkle_tree.rs,
 *     iop/challenger.rs, fri/{prover,verifier}.rs.
 *   Pins: Poseidon by upstream's known-answer vectors + constant regeneration
 *   (tools/gen_poseidon_constants.py); NTT by an O(n^2) DFT; BLAKE2b-256 /
 *   SHA-256 by RFC 7693 / FIPS 180-4 via Python hashlib; decoders by the
 *   literal vectors of decoder.rs:238-249, 388-395.  Proof-byte parity against
 *   the Rust prover is UNPINNED (no reference test holds proof bytes).
 */
#include "goldilocks.h"
#include "poseidon_constants.h"
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ field batch (K1) */
EXPORT void vxo_batch_add(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
    for (size_t i = 0; i < n; ++i) o[i] = gl_add(a[i], b[i]);
}
EXPORT void vxo_batch_sub(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
    for (size_t i = 0; i < n; ++i) o[i] = gl_sub(a[i], b[i]);
}
EXPORT void vxo_batch_mul(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
    for (size_t i = 0; i < n; ++i) o[i] = gl_mul(a[i], b[i]);
}
/* plonky2 Field::batch_multiplicative_inverse semantics: inverse of every
 * element; zero is not allowed upstream (panics) -- here 0 maps to 0. */
EXPORT void vxo_batch_inv(const uint64_t* a, uint64_t* o, size_t n) {
    /* Montgomery's trick in blocks of 256 (one exponentiation per block instead of per element; the same values) */
    enum { B = 256 };
    uint64_t pre[B];
    for (size_t s = 0; s < n; s += B) {
        const size_t m = n - s < B ? n - s : B;
        uint64_t acc = 1;
        for (size_t i = 0; i < m; ++i) {
            pre[i] = acc;
            if (a[s + i]) acc = gl_mul(acc, a[s + i]);
        }
        uint64_t inv = gl_inv(acc);
        for (size_t i = m; i-- > 0;) {
            const uint64_t x = a[s + i];  /* (o may alias a) */
            if (x) {
                o[s + i] = gl_mul(inv, pre[i]);
                inv = gl_mul(inv, x);
            } else o[s + i] = 0;
        }
    }
}
EXPORT uint64_t vxo_pow(uint64_t a, uint64_t e) { return gl_pow(a, e); }
EXPORT uint64_t vxo_root(int log_n) { return gl_root(log_n); }
EXPORT void vxo_ext_mul(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        gl2_t x = {{a[2 * i], a[2 * i + 1]}}, y = {{b[2 * i], b[2 * i + 1]}};
        gl2_t r = gl2_mul(x, y);
        o[2 * i] = r.c[0];
        o[2 * i + 1] = r.c[1];
    }
}
EXPORT void vxo_ext_inv(const uint64_t* a, uint64_t* o, size_t n) {
    /* the same trick in the quadratic extension; 0 maps to 0 as gl2_inv does */
    enum { B = 256 };
    gl2_t pre[B];
    for (size_t s = 0; s < n; s += B) {
        const size_t m = n - s < B ? n - s : B;
        gl2_t acc = {{1, 0}};
        for (size_t i = 0; i < m; ++i) {
            pre[i] = acc;
            gl2_t x = {{a[2 * (s + i)], a[2 * (s + i) + 1]}};
            if (x.c[0] | x.c[1]) acc = gl2_mul(acc, x);
        }
        gl2_t inv = gl2_inv(acc);
        for (size_t i = m; i-- > 0;) {
            gl2_t x = {{a[2 * (s + i)], a[2 * (s + i) + 1]}};
            if (x.c[0] | x.c[1]) {
                gl2_t r = gl2_mul(inv, pre[i]);
                inv = gl2_mul(inv, x);
                o[2 * (s + i)] = r.c[0], o[2 * (s + i) + 1] = r.c[1];
            } else o[2 * (s + i)] = 0, o[2 * (s + i) + 1] = 0;
        }
    }
}

/* ------------------------------------------------------------------ NTT (K2) */
static inline size_t bitrev(size_t x, int bits) {
    size_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
static void bitrev_permute(uint64_t* a, int log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; ++i) {
        size_t j = bitrev(i, log_n);
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}
/* values[i] = sum_k coeffs[k] * w^(i k), w = primitive_root_of_unity(log_n);
 * natural order in and out (plonky2_field::fft::fft). */
static void ntt_root(uint64_t* a, int log_n, uint64_t w) {
    size_t n = (size_t)1 << log_n;
    bitrev_permute(a, log_n);
    for (int s = 1; s <= log_n; ++s) {
        size_t m = (size_t)1 << s, h = m >> 1;
        uint64_t wm = w;
        for (int i = s; i < log_n; ++i) wm = gl_mul(wm, wm);
        uint64_t* tw = (uint64_t*)malloc(h * sizeof(uint64_t));
        tw[0] = 1;
        for (size_t j = 1; j < h; ++j) tw[j] = gl_mul(tw[j - 1], wm);
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < h; ++j) {
                uint64_t t = gl_mul(tw[j], a[k + j + h]);
                uint64_t u = a[k + j];
                a[k + j] = gl_add(u, t);
                a[k + j + h] = gl_sub(u, t);
            }
        free(tw);
    }
}
EXPORT void vxo_ntt(uint64_t* a, int log_n) { ntt_root(a, log_n, gl_root(log_n)); }
/* plonky2_field::fft::ifft: inverse root, then scale by n^-1. */
EXPORT void vxo_intt(uint64_t* a, int log_n) {
    size_t n = (size_t)1 << log_n;
    ntt_root(a, log_n, gl_inv(gl_root(log_n)));
    uint64_t ni = gl_inv((uint64_t)n % GL_P);
    for (size_t i = 0; i < n; ++i) a[i] = gl_mul(a[i], ni);
}
/* PolynomialCoeffs::coset_fft(shift): c_k *= shift^k, then fft. */
EXPORT void vxo_coset_ntt(uint64_t* a, int log_n, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
    uint64_t s = 1;
    for (size_t i = 0; i < n; ++i) { a[i] = gl_mul(a[i], s); s = gl_mul(s, shift); }
    vxo_ntt(a, log_n);
}
/* PolynomialValues::coset_ifft(shift): ifft, then c_k *= shift^-k. */
EXPORT void vxo_coset_intt(uint64_t* a, int log_n, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
    vxo_intt(a, log_n);
    uint64_t si = gl_inv(shift), s = 1;
    for (size_t i = 0; i < n; ++i) { a[i] = gl_mul(a[i], s); s = gl_mul(s, si); }
}
EXPORT void vxo_ntt_batch(uint64_t* a, int log_n, size_t n_cols, int inverse, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < n_cols; ++c) {
        uint64_t* col = a + c * n;
        if (!inverse) { if (shift > 1) vxo_coset_ntt(col, log_n, shift); else vxo_ntt(col, log_n); }
        else { if (shift > 1) vxo_coset_intt(col, log_n, shift); else vxo_intt(col, log_n); }
    }
}

/* ------------------------------------------------------------------ LDE (K3)
 * PolynomialBatch::from_coeffs (plonky2/src/fri/oracle.rs): for every
 * polynomial, lde(rate_bits) = zero-pad to n<<rate_bits, coset_fft(shift);
 * transpose to one leaf per evaluation point; reverse_index_bits_in_place.
 * coeffs: column-major [n_cols][n].  leaves: row-major [N][n_cols], row i =
 * evaluations at shift * w_N^bitrev(i). */
EXPORT void vxo_lde_from_coeffs(const uint64_t* coeffs, int log_n, size_t n_cols, int rate_bits,
                                uint64_t shift, uint64_t* leaves) {
    size_t n = (size_t)1 << log_n;
    int log_N = log_n + rate_bits;
    size_t N = (size_t)1 << log_N;
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < n_cols; ++c) {
        uint64_t* tmp = (uint64_t*)calloc(N, sizeof(uint64_t));
        memcpy(tmp, coeffs + c * n, n * sizeof(uint64_t));
        vxo_coset_ntt(tmp, log_N, shift);
        for (size_t i = 0; i < N; ++i) leaves[bitrev(i, log_N) * n_cols + c] = tmp[i];
        free(tmp);
    }
}
/* PolynomialBatch::from_values: ifft every column first.  coeffs_out (may be
 * NULL) receives the column-major coefficients. */
EXPORT void vxo_lde_from_values(const uint64_t* values, int log_n, size_t n_cols, int rate_bits,
                                uint64_t shift, uint64_t* leaves, uint64_t* coeffs_out) {
    size_t n = (size_t)1 << log_n;
    uint64_t* co = coeffs_out ? coeffs_out : (uint64_t*)malloc(n * n_cols * sizeof(uint64_t));
    memcpy(co, values, n * n_cols * sizeof(uint64_t));
    vxo_ntt_batch(co, log_n, n_cols, 1, 0);
    vxo_lde_from_coeffs(co, log_n, n_cols, rate_bits, shift, leaves);
    if (!coeffs_out) free(co);
}

/* ------------------------------------------------------------------ Poseidon (K4) */
static const uint64_t RC[360] = VX_POSEIDON_RC_INIT;
static const uint64_t MDS_CIRC[12] = VX_POSEIDON_MDS_CIRC_INIT;

static inline uint64_t sbox7(uint64_t x) {
    uint64_t x2 = gl_mul(x, x), x4 = gl_mul(x2, x2), x3 = gl_mul(x2, x);
    return gl_mul(x4, x3);
}
static void mds_layer(uint64_t* s) {
    uint64_t o[12];
    for (int r = 0; r < 12; ++r) {
        unsigned __int128 acc = 0; /* 12 * 2^64 * 41 < 2^128 */
        for (int i = 0; i < 12; ++i) acc += (unsigned __int128)s[(i + r) % 12] * MDS_CIRC[i];
        if (r == 0) acc += (unsigned __int128)s[0] * VX_POSEIDON_MDS_DIAG0;
        o[r] = (uint64_t)(acc % GL_P);
    }
    memcpy(s, o, sizeof o);
}
/* plonky2 Poseidon::poseidon_naive ordering: constant layer, s-box, MDS. */
EXPORT void vxo_poseidon(uint64_t* s) {
    for (int r = 0; r < 30; ++r) {
        for (int i = 0; i < 12; ++i) s[i] = gl_add(s[i], RC[12 * r + i]);
        if (r < 4 || r >= 26) { for (int i = 0; i < 12; ++i) s[i] = sbox7(s[i]); }
        else s[0] = sbox7(s[0]);
        mds_layer(s);
    }
}
EXPORT void vxo_poseidon_batch(uint64_t* states, size_t n) {
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i) vxo_poseidon(states + 12 * i);
}
/* hash_n_to_m_no_pad with m = 4 (hash/hashing.rs): overwrite-mode sponge, rate 8. */
EXPORT void vxo_hash_no_pad(const uint64_t* in, size_t len, uint64_t* out4) {
    uint64_t s[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t k = len - off < 8 ? len - off : 8;
        memcpy(s, in + off, k * sizeof(uint64_t));
        vxo_poseidon(s);
    }
    memcpy(out4, s, 4 * sizeof(uint64_t));
}
/* Hasher::hash_or_noop: inputs of <= 4 elements are zero-padded, not hashed. */
EXPORT void vxo_hash_or_noop(const uint64_t* in, size_t len, uint64_t* out4) {
    if (len <= 4) {
        memset(out4, 0, 4 * sizeof(uint64_t));
        memcpy(out4, in, len * sizeof(uint64_t));
    } else vxo_hash_no_pad(in, len, out4);
}
/* hashing::compress = PoseidonHash::two_to_one */
EXPORT void vxo_two_to_one(const uint64_t* l, const uint64_t* r, uint64_t* out4) {
    uint64_t s[12] = {0};
    memcpy(s, l, 32);
    memcpy(s + 4, r, 32);
    vxo_poseidon(s);
    memcpy(out4, s, 32);
}

/* ------------------------------------------------------------------ Merkle tree with cap
 * MerkleTree::new(leaves, cap_height) (hash/merkle_tree.rs).  The digest
 * storage order upstream is an implementation detail; cap and authentication
 * paths are not.  Here: level 0 = leaf digests (n*4), level k = n>>k nodes,
 * stored back to back up to and including the cap level (2^cap_height nodes).
 * Returns number of uint64 written to `levels`. */
EXPORT size_t vxo_merkle_build(const uint64_t* leaves, size_t n_leaves, size_t leaf_len,
                               int cap_height, uint64_t* levels) {
    size_t off = 0;
#pragma omp parallel for
    for (size_t i = 0; i < n_leaves; ++i) vxo_hash_or_noop(leaves + i * leaf_len, leaf_len, levels + 4 * i);
    size_t cur = n_leaves, cap = (size_t)1 << cap_height;
    while (cur > cap) {
        uint64_t* src = levels + off;
        uint64_t* dst = src + 4 * cur;
#pragma omp parallel for
        for (size_t i = 0; i < cur / 2; ++i) vxo_two_to_one(src + 8 * i, src + 8 * i + 4, dst + 4 * i);
        off += 4 * cur;
        cur >>= 1;
    }
    return off + 4 * cur;
}
EXPORT size_t vxo_merkle_levels_len(size_t n_leaves, int cap_height) {
    size_t t = 0, cur = n_leaves, cap = (size_t)1 << cap_height;
    while (cur > cap) { t += 4 * cur; cur >>= 1; }
    return t + 4 * cur;
}
EXPORT void vxo_merkle_cap(const uint64_t* levels, size_t n_leaves, int cap_height, uint64_t* cap_out) {
    size_t len = vxo_merkle_levels_len(n_leaves, cap_height);
    size_t cap = (size_t)1 << cap_height;
    memcpy(cap_out, levels + len - 4 * cap, 4 * cap * sizeof(uint64_t));
}
/* MerkleTree::prove(leaf_index): siblings bottom-up, log2(n) - cap_height of them. */
EXPORT size_t vxo_merkle_prove(const uint64_t* levels, size_t n_leaves, int cap_height, size_t idx,
                               uint64_t* siblings) {
    size_t off = 0, cur = n_leaves, cap = (size_t)1 << cap_height, k = 0;
    while (cur > cap) {
        memcpy(siblings + 4 * k, levels + off + 4 * (idx ^ 1), 32);
        off += 4 * cur;
        cur >>= 1;
        idx >>= 1;
        ++k;
    }
    return k;
}
/* verify_merkle_proof_to_cap */
EXPORT int vxo_merkle_verify(const uint64_t* leaf, size_t leaf_len, size_t idx, const uint64_t* siblings,
                             size_t n_sib, const uint64_t* cap) {
    uint64_t cur[4], nxt[4];
    vxo_hash_or_noop(leaf, leaf_len, cur);
    for (size_t k = 0; k < n_sib; ++k) {
        if (idx & 1) vxo_two_to_one(siblings + 4 * k, cur, nxt);
        else vxo_two_to_one(cur, siblings + 4 * k, nxt);
        memcpy(cur, nxt, 32);
        idx >>= 1;
    }
    return memcmp(cur, cap + 4 * idx, 32) == 0;
}

/* ------------------------------------------------------------------ Challenger (K7)
 * iop/challenger.rs: duplex sponge over the Poseidon permutation, rate 8,
 * overwrite mode; challenges are popped from the END of the output buffer. */
typedef struct {
    uint64_t state[12];
    uint64_t in[8];
    int n_in;
    uint64_t out[8];
    int n_out;
} vxo_challenger;
EXPORT void vxo_ch_init(vxo_challenger* c) { memset(c, 0, sizeof *c); }
static void ch_duplex(vxo_challenger* c) {
    for (int i = 0; i < c->n_in; ++i) c->state[i] = c->in[i];
    c->n_in = 0;
    vxo_poseidon(c->state);
    memcpy(c->out, c->state, 8 * sizeof(uint64_t));
    c->n_out = 8;
}
EXPORT void vxo_ch_observe(vxo_challenger* c, const uint64_t* v, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        c->n_out = 0;
        c->in[c->n_in++] = v[i];
        if (c->n_in == 8) ch_duplex(c);
    }
}
EXPORT uint64_t vxo_ch_challenge(vxo_challenger* c) {
    if (c->n_in > 0 || c->n_out == 0) ch_duplex(c);
    return c->out[--c->n_out];
}
EXPORT size_t vxo_ch_sizeof(void) { return sizeof(vxo_challenger); }

/* ------------------------------------------------------------------ FRI (K6)
 * fri/prover.rs fri_committed_trees, one reduction step in COEFFICIENT space
 * (the reference prover's way): coefficients are extension elements
 * (interleaved c0,c1), chunks of `arity` are reduced with powers of beta
 * (Horner from the top).  Output has n/arity coefficients. */
EXPORT void vxo_fri_fold_coeffs(const uint64_t* coeffs, size_t n, int arity_bits, const uint64_t* beta,
                                uint64_t* out) {
    size_t arity = (size_t)1 << arity_bits;
    gl2_t b = {{beta[0], beta[1]}};
    for (size_t i = 0; i < n / arity; ++i) {
        gl2_t acc = gl2_from(0);
        for (size_t j = arity; j-- > 0;) {
            gl2_t c = {{coeffs[2 * (i * arity + j)], coeffs[2 * (i * arity + j) + 1]}};
            acc = gl2_add(gl2_mul(acc, b), c);
        }
        out[2 * i] = acc.c[0];
        out[2 * i + 1] = acc.c[1];
    }
}
/* extension-coefficient coset FFT: both components transformed independently
 * (a base-field shift and base-field twiddles act componentwise). In/out
 * interleaved, natural order. */
EXPORT void vxo_ext_coset_ntt(uint64_t* a, int log_n, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
    uint64_t* t = (uint64_t*)malloc(n * sizeof(uint64_t));
    for (int comp = 0; comp < 2; ++comp) {
        for (size_t i = 0; i < n; ++i) t[i] = a[2 * i + comp];
        vxo_coset_ntt(t, log_n, shift);
        for (size_t i = 0; i < n; ++i) a[2 * i + comp] = t[i];
    }
    free(t);
}
EXPORT void vxo_ext_coset_intt(uint64_t* a, int log_n, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
    uint64_t* t = (uint64_t*)malloc(n * sizeof(uint64_t));
    for (int comp = 0; comp < 2; ++comp) {
        for (size_t i = 0; i < n; ++i) t[i] = a[2 * i + comp];
        vxo_coset_intt(t, log_n, shift);
        for (size_t i = 0; i < n; ++i) a[2 * i + comp] = t[i];
    }
    free(t);
}
/* fri/verifier.rs compute_evaluation: given the `arity` evaluations of one
 * coset (in the bit-reversed order they sit in a committed leaf), the coset
 * member x (a base-field point) and its index within the coset, interpolate
 * and evaluate at beta.  Straight Lagrange (the upstream barycentric form
 * computes the same polynomial value). */
EXPORT void vxo_fri_compute_evaluation(uint64_t x, size_t x_index_within_coset, int arity_bits,
                                       const uint64_t* evals_ext, const uint64_t* beta, uint64_t* out) {
    size_t arity = (size_t)1 << arity_bits;
    uint64_t g = gl_root(arity_bits);
    gl2_t ev[64];
    uint64_t pts[64];
    for (size_t i = 0; i < arity; ++i) {
        size_t j = bitrev(i, arity_bits);
        ev[j].c[0] = evals_ext[2 * i];
        ev[j].c[1] = evals_ext[2 * i + 1];
    }
    size_t rev = bitrev(x_index_within_coset, arity_bits);
    uint64_t start = gl_mul(x, gl_pow(g, arity - rev));
    uint64_t gp = 1;
    for (size_t i = 0; i < arity; ++i) { pts[i] = gl_mul(start, gp); gp = gl_mul(gp, g); }
    gl2_t b = {{beta[0], beta[1]}}, acc = gl2_from(0);
    for (size_t i = 0; i < arity; ++i) {
        gl2_t num = gl2_from(1);
        uint64_t den = 1;
        for (size_t j = 0; j < arity; ++j) {
            if (j == i) continue;
            num = gl2_mul(num, gl2_sub(b, gl2_from(pts[j])));
            den = gl_mul(den, gl_sub(pts[i], pts[j]));
        }
        acc = gl2_add(acc, gl2_mul(ev[i], gl2_scale(num, gl_inv(den))));
    }
    out[0] = acc.c[0];
    out[1] = acc.c[1];
}
/* fri/prover.rs fri_proof_of_work: the duplex that follows observing the
 * witness is split; candidate goes at input position `pos`; the response is
 * the LAST squeezed element (state[7]); valid when leading_zeros >= bits + 0
 * ((64 - order.bits()) = 0 for Goldilocks).  Upstream uses a parallel
 * find_any (scheduling-dependent); policy here and in the product: SMALLEST
 * valid nonce, which is always one of find_any's admissible answers. */
EXPORT uint64_t vxo_fri_pow(const uint64_t* state12, int pos, int bits, uint64_t start, uint64_t max_iter) {
    for (uint64_t cand = start; cand < start + max_iter; ++cand) {
        uint64_t s[12];
        memcpy(s, state12, sizeof s);
        s[pos] = cand;
        vxo_poseidon(s);
        if (bits == 0 || (s[7] >> (64 - bits)) == 0) return cand;
    }
    return UINT64_MAX;
}

/* ------------------------------------------------------------------ BLAKE2b-256 (RFC 7693)
 * circuits/builder/header.rs:14-19 hash_encoded_header == blake2b-256 of the
 * first `size` bytes (native mirror: Blake2Hasher::hash, header.rs:215). */
static const uint64_t B2B_IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                   0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                   0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
static const uint8_t B2B_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
static inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
EXPORT void vxo_blake2b_compress(uint64_t* h, const uint8_t* block, uint64_t t, int last) {
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; ++i) memcpy(&m[i], block + 8 * i, 8);
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = B2B_IV[i]; }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
#define G(a, b, c, d, x, y)                                   \
    v[a] = v[a] + v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 24); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); \
    v[c] = v[c] + v[d];       v[b] = rotr64(v[b] ^ v[c], 63);
    for (int r = 0; r < 12; ++r) {
        const uint8_t* s = B2B_SIGMA[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef G
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}
EXPORT void vxo_blake2b_256(const uint8_t* msg, size_t len, uint8_t* out32) {
    uint64_t h[8];
    memcpy(h, B2B_IV, sizeof h);
    h[0] ^= 0x01010000ULL ^ 32;
    size_t off = 0;
    uint8_t blk[128];
    while (len - off > 128) {
        vxo_blake2b_compress(h, msg + off, off + 128, 0);
        off += 128;
    }
    memset(blk, 0, 128);
    memcpy(blk, msg + off, len - off);
    vxo_blake2b_compress(h, blk, len, 1);
    memcpy(out32, h, 32);
}

/* ------------------------------------------------------------------ SHA-256 (FIPS 180-4) */
static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_block(uint32_t* h, const uint8_t* p) {
    uint32_t w[64], a, b, c, d, e, f, g, hh;
    for (int i = 0; i < 16; ++i)
        w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
        uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    a = h[0]; b = h[1]; c = h[2]; d = h[3]; e = h[4]; f = h[5]; g = h[6]; hh = h[7];
    for (int i = 0; i < 64; ++i) {
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + SHA_K[i] + w[i];
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
EXPORT void vxo_sha256(const uint8_t* msg, size_t len, uint8_t* out32) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t off = 0;
    for (; off + 64 <= len; off += 64) sha256_block(h, msg + off);
    uint8_t blk[128];
    size_t rem = len - off;
    memset(blk, 0, sizeof blk);
    memcpy(blk, msg + off, rem);
    blk[rem] = 0x80;
    size_t tot = rem + 1 + 8 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; ++i) blk[tot - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_block(h, blk);
    if (tot == 128) sha256_block(h, blk + 64);
    for (int i = 0; i < 8; ++i) {
        out32[4 * i] = (uint8_t)(h[i] >> 24); out32[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out32[4 * i + 2] = (uint8_t)(h[i] >> 8); out32[4 * i + 3] = (uint8_t)h[i];
    }
}

/* ------------------------------------------------------------------ SCALE / header decoding
 * decoder.rs:39-92 decode_compact_int over 5 bytes -> (value, mode); returns
 * -1 when the in-circuit assertion (mode 3 => upper 6 bits zero, :83-89) fails. */
EXPORT int vxo_decode_compact_int(const uint8_t* b5, uint32_t* value, uint32_t* mode) {
    uint32_t m = b5[0] & 3;
    *mode = m;
    switch (m) {
    case 0: *value = b5[0] >> 2; break;
    case 1: *value = ((uint32_t)b5[0] | ((uint32_t)b5[1] << 8)) >> 2; break;
    case 2: *value = ((uint32_t)b5[0] | ((uint32_t)b5[1] << 8) | ((uint32_t)b5[2] << 16) | ((uint32_t)b5[3] << 24)) >> 2; break;
    default:
        *value = (uint32_t)b5[1] | ((uint32_t)b5[2] << 8) | ((uint32_t)b5[3] << 16) | ((uint32_t)b5[4] << 24);
        if ((b5[0] >> 2) != 0) return -1;
    }
    return 0;
}
/* decoder.rs:94-103 */
EXPORT uint32_t vxo_compact_int_byte_length(uint32_t mode) {
    static const uint32_t L[4] = {1, 2, 4, 5};
    return L[mode & 3];
}
typedef struct {
    uint32_t block_number;
    uint8_t parent_hash[32];
    uint8_t state_root[32];
    uint8_t data_root[32];
} vxo_header;
/* decoder.rs:104-157 decode_header over the zero-padded MAX_HEADER_SIZE buffer. */
EXPORT int vxo_decode_header(const uint8_t* bytes, uint32_t size, vxo_header* out) {
    uint32_t mode;
    memcpy(out->parent_hash, bytes, 32);
    int rc = vxo_decode_compact_int(bytes + 32, &out->block_number, &mode);
    static const int OFF[4] = {33, 34, 36, 37};
    memcpy(out->state_root, bytes + OFF[mode], 32);
    uint32_t start = size == 0 ? 0 : size - 32;
    memcpy(out->data_root, bytes + start, 32);
    return rc;
}
/* decoder.rs:159-200 / input/mod.rs:262-290 */
EXPORT int vxo_decode_precommit(const uint8_t* p53, uint8_t* block_hash, uint32_t* block_number,
                                uint64_t* round, uint64_t* set_id) {
    if (p53[0] != 1) return -1;
    memcpy(block_hash, p53 + 1, 32);
    uint32_t bn = 0; uint64_t r = 0, s = 0;
    for (int i = 0; i < 4; ++i) bn |= (uint32_t)p53[33 + i] << (8 * i);
    for (int i = 0; i < 8; ++i) { r |= (uint64_t)p53[37 + i] << (8 * i); s |= (uint64_t)p53[45 + i] << (8 * i); }
    *block_number = bn; *round = r; *set_id = s;
    return 0;
}
/* input/mod.rs:464-489 get_merkle_root: leaves NOT hashed, zero leaves pad to
 * a power of two, node = SHA256(l || r). n must be a power of two here. */
EXPORT void vxo_simple_merkle_root(const uint8_t* leaves32, size_t n, uint8_t* out32) {
    uint8_t* cur = (uint8_t*)malloc(32 * n);
    memcpy(cur, leaves32, 32 * n);
    while (n > 1) {
        for (size_t i = 0; i < n / 2; ++i) {
            uint8_t tmp[32];
            vxo_sha256(cur + 64 * i, 64, tmp);
            memcpy(cur + 32 * i, tmp, 32);
        }
        n >>= 1;
    }
    memcpy(out32, cur, 32);
    free(cur);
}
/* input/mod.rs:250-260 compute_authority_set_hash == justification.rs:127-162 */
EXPORT void vxo_authority_set_hash(const uint8_t* pubkeys32, size_t n, uint8_t* out32) {
    uint8_t buf[64];
    vxo_sha256(pubkeys32, 32, out32);
    for (size_t i = 1; i < n; ++i) {
        memcpy(buf, out32, 32);
        memcpy(buf + 32, pubkeys32 + 32 * i, 32);
        vxo_sha256(buf, 64, out32);
    }
}

/* ------------------------------------------------------------------ verify_subchain restated
 * subchain_verification.rs:56-303.  `headers`: n_fetched encoded headers for
 * blocks trusted+1 .. target, each zero-padded to `stride` bytes, with sizes.
 * N = MAX_NUM_HEADERS (256/512).  Runs the MAP closure per batch of 8
 * (with the hint's zero-header padding, :366-372), the REDUCE tree and the
 * final assertions; any violated in-circuit assertion returns a negative code.
 * out96 = target_header_hash || state_root_merkle_root || data_root_merkle_root
 * (header_range.rs:56-58). */
typedef struct {
    uint64_t num_blocks;
    uint32_t start_block, end_block;
    uint8_t start_header_hash[32], start_parent[32], end_header_hash[32];
    uint8_t state_root[32], data_root[32];
} vxo_mr;
enum { VXO_E_LINK = -2, VXO_E_FIRST = -3, VXO_E_LAST = -4, VXO_E_REDUCE = -5, VXO_E_TRUSTED = -6, VXO_E_TARGET = -7, VXO_E_COMPACT = -8 };
static int map_job(const uint8_t* headers, const uint32_t* sizes, size_t stride, size_t n_fetched,
                   uint32_t global_start, uint32_t global_end, uint32_t rel0, vxo_mr* out) {
    const int M = 8;
    uint32_t batch_start = global_start + rel0, batch_end = global_start + rel0 + M - 1;
    int batch_disabled = global_end < batch_start;
    int noop = batch_disabled;
    uint8_t* zero_hdr = (uint8_t*)calloc(stride, 1);
    uint8_t hashes[8][32];
    vxo_header hv[8];
    uint8_t state_leaves[8 * 32], data_leaves[8 * 32];
    uint32_t end_block = 0, nb_enabled = 0;
    uint8_t end_hash[32] = {0};
    uint64_t num_headers = 0;
    int rc = 0;
    for (int i = 0; i < M; ++i) {
        /* HeaderRangeFetcherHint: fetch [batch_start, min(batch_end, global_end)], pad with (zeros, 0). */
        uint32_t blk = batch_start + i;
        const uint8_t* hb = zero_hdr;
        uint32_t sz = 0;
        if (blk <= global_end && blk > global_start && (size_t)(blk - global_start - 1) < n_fetched) {
            hb = headers + (size_t)(blk - global_start - 1) * stride;
            sz = sizes[blk - global_start - 1];
        }
        vxo_blake2b_256(hb, sz, hashes[i]);
        if (vxo_decode_header(hb, sz, &hv[i]) != 0 && !noop) rc = VXO_E_COMPACT;
        if (i > 0) {
            int linked = memcmp(hv[i].parent_hash, hashes[i - 1], 32) == 0 && hv[i].block_number == hv[i - 1].block_number + 1;
            if (!(noop || linked)) rc = rc ? rc : VXO_E_LINK;
        }
        if (!noop) { end_block = hv[i].block_number; memcpy(end_hash, hashes[i], 32); num_headers++; nb_enabled++; }
        if (hv[i].block_number == global_end) noop = 1;
    }
    if (!(hv[0].block_number == batch_start || batch_disabled)) rc = rc ? rc : VXO_E_FIRST;
    if (!(end_block == batch_end || noop)) rc = rc ? rc : VXO_E_LAST;
    /* get_root_from_hashed_leaves(leaves, nb_enabled): disabled leaves are zero leaves (mirror input/mod.rs:518-521). */
    for (int i = 0; i < M; ++i) {
        if ((uint32_t)i < nb_enabled) { memcpy(state_leaves + 32 * i, hv[i].state_root, 32); memcpy(data_leaves + 32 * i, hv[i].data_root, 32); }
        else { memset(state_leaves + 32 * i, 0, 32); memset(data_leaves + 32 * i, 0, 32); }
    }
    out->num_blocks = num_headers;
    out->start_block = hv[0].block_number;
    memcpy(out->start_header_hash, hashes[0], 32);
    memcpy(out->start_parent, hv[0].parent_hash, 32);
    out->end_block = end_block;
    memcpy(out->end_header_hash, end_hash, 32);
    vxo_simple_merkle_root(state_leaves, M, out->state_root);
    vxo_simple_merkle_root(data_leaves, M, out->data_root);
    free(zero_hdr);
    return rc;
}
static int reduce_job(const vxo_mr* l, const vxo_mr* r, vxo_mr* o) {
    int linked = memcmp(l->end_header_hash, r->start_parent, 32) == 0 && l->end_block == r->start_block - 1;
    int right_inactive = r->num_blocks == 0;
    int rc = (right_inactive || linked) ? 0 : VXO_E_REDUCE;
    uint8_t buf[64];
    vxo_mr t = *l;
    t.end_block = right_inactive ? l->end_block : r->end_block;
    memcpy(t.end_header_hash, right_inactive ? l->end_header_hash : r->end_header_hash, 32);
    memcpy(buf, l->state_root, 32); memcpy(buf + 32, r->state_root, 32); vxo_sha256(buf, 64, t.state_root);
    memcpy(buf, l->data_root, 32); memcpy(buf + 32, r->data_root, 32); vxo_sha256(buf, 64, t.data_root);
    t.num_blocks = l->num_blocks + r->num_blocks;
    *o = t;
    return rc;
}
EXPORT int vxo_verify_subchain(const uint8_t* headers, const uint32_t* sizes, size_t stride, size_t n_fetched,
                               uint32_t N, uint32_t trusted_block, const uint8_t* trusted_hash, uint32_t target_block,
                               uint8_t* out96) {
    size_t J = 1;
    while (J < N / 8) J <<= 1;
    vxo_mr* nodes = (vxo_mr*)malloc(J * sizeof(vxo_mr));
    int rc = 0;
    for (size_t j = 0; j < J; ++j) {
        int r = map_job(headers, sizes, stride, n_fetched, trusted_block, target_block, (uint32_t)(8 * j + 1), &nodes[j]);
        if (r && !rc) rc = r;
    }
    for (size_t w = J; w > 1; w >>= 1)
        for (size_t i = 0; i < w / 2; ++i) {
            vxo_mr o;
            int r = reduce_job(&nodes[2 * i], &nodes[2 * i + 1], &o);
            if (r && !rc) rc = r;
            nodes[i] = o;
        }
    if (memcmp(trusted_hash, nodes[0].start_parent, 32) != 0 && !rc) rc = VXO_E_TRUSTED;
    if (target_block != nodes[0].end_block && !rc) rc = VXO_E_TARGET;
    memcpy(out96, nodes[0].end_header_hash, 32);
    memcpy(out96 + 32, nodes[0].state_root, 32);
    memcpy(out96 + 64, nodes[0].data_root, 32);
    free(nodes);
    return rc;
}
/* dummy_header_range.rs:11-52 + input/mod.rs:493-528: the native public-output
 * computation (no chain checks): target hash, and Merkle roots over N leaves. */
EXPORT void vxo_dummy_header_range(const uint8_t* headers, const uint32_t* sizes, size_t stride, size_t n_fetched,
                                   uint32_t N, uint8_t* out96) {
    uint8_t* st = (uint8_t*)calloc(N, 32);
    uint8_t* dt = (uint8_t*)calloc(N, 32);
    for (size_t i = 0; i < n_fetched; ++i) {
        vxo_header h;
        vxo_decode_header(headers + i * stride, sizes[i], &h);
        memcpy(st + 32 * i, h.state_root, 32);
        memcpy(dt + 32 * i, h.data_root, 32);
    }
    vxo_blake2b_256(headers + (n_fetched - 1) * stride, sizes[n_fetched - 1], out96);
    vxo_simple_merkle_root(st, N, out96 + 32);
    vxo_simple_merkle_root(dt, N, out96 + 64);
    free(st);
    free(dt);
}
