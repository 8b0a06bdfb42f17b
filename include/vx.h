/*
 * vx.h -- C ABI of libvxprove, the MI355X-native (gfx950) backend for the
 * header_range proving path of VectorX (reference: AsherBond/0-kno-vectorx).
 *
 * The reference has no FFI of its own: its seams are the
 * `Circuit::prove` library call (circuits/header_range.rs:167-170) and cargo's
 * dependency-override mechanism (Cargo.toml:101-106).  Each group below names
 * the upstream routine (plonky2 v0.2.0 / starkyx v1.0.0 / plonky2x v1.1.0,
 * pinned at Cargo.lock:4848-4910, 7232-7249) that a `[patch]`-ed Rust shim
 * would forward to this library, and the in-tree call site that reaches it.
 * INTEGRATION.md shows the Rust `extern "C"` stubs.
 *
 * Conventions
 *   - plain C, no exceptions; every call returns int32_t: 0 = VX_OK, <0 = VX_ERR_*;
 *     vx_last_error(ctx) returns a message for the last failure on that ctx.
 *   - field elements: canonical (< p = 2^64-2^32+1) little-endian uint64_t;
 *     extension elements (D = 2, X^2 = 7): two consecutive uint64_t (c0, c1).
 *   - matrices are COLUMN-MAJOR: one polynomial / trace column is contiguous.
 *   - the caller owns host buffers; vx_buf are device (HBM) buffers owned by
 *     the ctx that allocated them.
 *   - a vx_ctx is bound to one device and one HIP stream; calls are
 *     asynchronous on that stream; vx_sync / vx_download block.  Not
 *     thread-safe: one ctx per host thread.
 */
#ifndef VX_H
#define VX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vx_ctx vx_ctx;
typedef struct vx_buf vx_buf;
typedef struct vx_tree vx_tree;

enum {
    VX_OK = 0,
    VX_ERR_ARG = -1,      /* bad argument (shape, range, null) */
    VX_ERR_DEVICE = -2,   /* HIP runtime failure / no gfx950 device */
    VX_ERR_OOM = -3,      /* device allocation failed */
    VX_ERR_BUFSZ = -4,    /* caller buffer too small; needed size reported */
    VX_ERR_STATEMENT = -5,/* the header_range statement does not hold for the witness
                             (the reference panics / fails an in-circuit assert) */
    VX_ERR_POW = -6       /* proof-of-work search exhausted */
};

/* ---- lifecycle ------------------------------------------------------------ */
int32_t vx_ctx_create(int device, vx_ctx** out);
int32_t vx_ctx_destroy(vx_ctx* ctx);
int32_t vx_sync(vx_ctx* ctx);
const char* vx_last_error(const vx_ctx* ctx);
const char* vx_backend_name(void); /* "hip-gfx950" */
/* elapsed milliseconds between two points of the ctx stream (HIP events):
 * vx_timer_start records, vx_timer_stop records + synchronises and returns ms. */
int32_t vx_timer_start(vx_ctx* ctx);
int32_t vx_timer_stop(vx_ctx* ctx, float* ms);

/* ---- memory --------------------------------------------------------------- */
int32_t vx_alloc(vx_ctx* ctx, size_t n_u64, vx_buf** out);
int32_t vx_free(vx_ctx* ctx, vx_buf* buf);
int32_t vx_upload(vx_ctx* ctx, vx_buf* dst, size_t dst_off, const uint64_t* src, size_t n_u64);
int32_t vx_download(vx_ctx* ctx, const vx_buf* src, size_t src_off, uint64_t* dst, size_t n_u64);
int32_t vx_copy(vx_ctx* ctx, vx_buf* dst, size_t dst_off, const vx_buf* src, size_t src_off, size_t n_u64);
/* fill with canonical pseudo-random field elements (SplitMix64 of seed+index, reduced mod p) */
int32_t vx_fill_random(vx_ctx* ctx, vx_buf* dst, size_t off, size_t n_u64, uint64_t seed);
void* vx_buf_devptr(const vx_buf* buf);
size_t vx_buf_len(const vx_buf* buf);

/* ---- K1: Goldilocks batch arithmetic (plonky2_field::goldilocks_field; test surface) */
int32_t vx_field_batch_add(vx_ctx* ctx, const vx_buf* a, const vx_buf* b, vx_buf* out, size_t n);
int32_t vx_field_batch_sub(vx_ctx* ctx, const vx_buf* a, const vx_buf* b, vx_buf* out, size_t n);
int32_t vx_field_batch_mul(vx_ctx* ctx, const vx_buf* a, const vx_buf* b, vx_buf* out, size_t n);
/* Field::batch_multiplicative_inverse; 0 maps to 0 */
int32_t vx_field_batch_inv(vx_ctx* ctx, const vx_buf* a, vx_buf* out, size_t n);
int32_t vx_ext_batch_mul(vx_ctx* ctx, const vx_buf* a, const vx_buf* b, vx_buf* out, size_t n_ext);

/* ---- K2: batched radix-2 NTT (plonky2_field::fft::{fft,ifft}, polynomial::{coset_fft,coset_ifft})
 * buf holds n_cols columns of n = 2^log_n elements at column stride `col_stride`
 * (elements).  Natural order in and out, in place.
 *   inverse = 0: values[i] = sum_k coeff[k] * (shift * w^i)^k        (coset_fft; shift 0/1 = plain fft)
 *   inverse = 1: the inverse map, including the 1/n factor           (coset_ifft / ifft)
 * order: VX_ORDER_NATURAL, or VX_ORDER_BITREV to leave a FORWARD result (or take an
 * INVERSE input) in bit-reversed positions and skip the permutation pass. */
enum { VX_ORDER_NATURAL = 0, VX_ORDER_BITREV = 1 };
int32_t vx_ntt(vx_ctx* ctx, vx_buf* buf, size_t off, int log_n, size_t n_cols, size_t col_stride,
               int inverse, uint64_t shift, int order);

/* ---- K3: low-degree extension (plonky2::fri::oracle::PolynomialBatch::{from_values,from_coeffs})
 * src: n_cols columns of n values (natural order) or, with VX_LDE_SRC_COEFFS, coefficients.
 * dst: n_cols columns of N = n << rate_bits evaluations on the coset shift*<w_N>, column-major,
 *      NATURAL order within a column: dst[c*N + i] = P_c(shift * w_N^i).  plonky2's leaf j
 *      (after its transpose + reverse_index_bits) is row i = bitrev_N(j); vx_merkle_build
 *      and vx_lde_rows take that mapping into account, no transposed copy is ever made.
 * coeffs_out (optional, may be NULL): n_cols columns of n coefficients (natural order).
 * src is preserved. */
enum { VX_LDE_SRC_VALUES = 0, VX_LDE_SRC_COEFFS = 1 };
int32_t vx_lde(vx_ctx* ctx, const vx_buf* src, int log_n, size_t n_cols, int rate_bits, uint64_t shift,
               int src_kind, vx_buf* dst, vx_buf* coeffs_out);
/* gather plonky2 leaves: for each of n_idx leaf indices j, the n_cols values of row bitrev(j)
 * -> out[k*n_cols + c] on the HOST (PolynomialBatch::get_lde_values) */
int32_t vx_lde_rows(vx_ctx* ctx, const vx_buf* lde, int log_N, size_t n_cols, const uint64_t* leaf_idx,
                    size_t n_idx, uint64_t* out);

/* ---- K4: Poseidon-Goldilocks + Merkle caps (plonky2::hash::{poseidon,hashing,merkle_tree}) */
/* in-place permutation of n states of 12 elements (state-major: 12 consecutive per state) */
int32_t vx_poseidon_permute_batch(vx_ctx* ctx, vx_buf* states, size_t n);
/* MerkleTree::new(leaves, cap_height).  Layouts of `data`:
 *   VX_LEAVES_ROW_MAJOR: leaf j = data[j*leaf_len .. +leaf_len)
 *   VX_LEAVES_COLS_BITREV: data is column-major [leaf_len][n_leaves]; leaf j = row bitrev(j)
 *                          of every column (the layout vx_lde produces)
 *   VX_LEAVES_COLS: column-major, leaf j = row j. */
enum { VX_LEAVES_ROW_MAJOR = 0, VX_LEAVES_COLS_BITREV = 1, VX_LEAVES_COLS = 2 };
int32_t vx_merkle_build(vx_ctx* ctx, const vx_buf* data, size_t off, size_t n_leaves, size_t leaf_len,
                        int layout, int cap_height, vx_tree** out);
int32_t vx_merkle_free(vx_ctx* ctx, vx_tree* tree);
/* cap_out: 4 * 2^cap_height elements (host) */
int32_t vx_merkle_cap(vx_ctx* ctx, const vx_tree* tree, uint64_t* cap_out);
/* MerkleTree::prove for n_idx leaves: siblings_out[k][log2(n_leaves)-cap_height][4] (host) */
int32_t vx_merkle_open(vx_ctx* ctx, const vx_tree* tree, const uint64_t* leaf_idx, size_t n_idx,
                       uint64_t* siblings_out);
/* leaf digests (4*n_leaves, device->host), test surface */
int32_t vx_merkle_leaf_digests(vx_ctx* ctx, const vx_tree* tree, uint64_t* out);

/* ---- K6: FRI (plonky2::fri::{prover,verifier,reduction_strategies}) */
/* One reduction step.  evals: N = 2^log_n extension values of a polynomial on the coset
 * shift*<w_N>, NATURAL order (evals[i] = P(shift*w_N^i), interleaved c0,c1).  Writes the
 * N >> arity_bits values of P'(y) = sum_i beta^i P_i(y) (P(x) = sum_i x^i P_i(x^arity)) on
 * the coset shift^arity * <w_{N/arity}>, natural order -- the values
 * fri_committed_trees obtains by folding coefficients and re-running coset_fft. */
int32_t vx_fri_fold(vx_ctx* ctx, const vx_buf* evals, int log_n, int arity_bits, const uint64_t beta[2],
                    uint64_t shift, vx_buf* out);
/* Merkle tree over a FRI layer: leaf j = the `arity` extension values whose natural indices
 * are bitrev(j*arity + t), t < arity (reverse_index_bits + chunk + flatten of prover.rs). */
int32_t vx_fri_layer_tree(vx_ctx* ctx, const vx_buf* evals, int log_n, int arity_bits, int cap_height,
                          vx_tree** out);
/* FriInitialTreeProof / FriQueryStep evals: the flattened `arity` extension values of each of
 * n_idx leaves of a FRI layer (MerkleTree::get on the layer tree) -> out[k][2*arity] (host) */
int32_t vx_fri_leaves(vx_ctx* ctx, const vx_buf* evals, int log_n, int arity_bits, const uint64_t* leaf_idx,
                      size_t n_idx, uint64_t* out);
/* fri_proof_of_work: smallest nonce w such that Poseidon(state with state[pos] = w)[7] has
 * `bits` leading zero bits.  (Upstream's parallel find_any returns an arbitrary valid nonce;
 * see DESIGN.md section 6.) */
int32_t vx_fri_pow(vx_ctx* ctx, const uint64_t state[12], int pos, int bits, uint64_t* nonce);

/* ---- K5/K7 + K6 query phase: generic STARK prover (starky v0.2.0 `prove_with_commitment` over
 * plonky2 v0.2.0 PolynomialBatch / FRI; reached in the reference through curta's hash/EdDSA
 * STARKs -- circuits/builder/header.rs:18, justification.rs:140,156,237 -- and, for the
 * Plonk side, through every `circuit.prove`, circuits/header_range.rs:167).
 * trace: n_cols(air) columns of 2^log_n rows, column-major, resident in HBM.
 * proof_out: caller buffer of proof_cap uint64 words; *proof_len receives the length; returns
 * VX_ERR_BUFSZ (with *proof_len set) when the buffer is too small.  Proof layout: DESIGN.md. */
typedef struct vx_stark_config {
    int32_t rate_bits;       /* 1: blowup 2, constraint degree <= 3 */
    int32_t cap_height;      /* 4 */
    int32_t num_queries;     /* 84 */
    int32_t pow_bits;        /* 16 */
    int32_t arity_bits;      /* ConstantArityBits(4, 5) */
    int32_t final_poly_bits;
} vx_stark_config;
/* The AIRs of the statements are COMPILED INTO the library: 1 Fibonacci and 2 "cubic mixer" pin the
 * generic prover, 5 is the smallest AIR with an auxiliary (lookup / logUp) commitment round.  A host can add its own at run
 * time as a constraint PROGRAM (vx_air_register below; no auxiliary round).  The tables of the
 * header_range / rotate statements, each declared with its trace generator below: 6 Blake2b header chain
 * (VX_AIR_BLAKE_CHAIN), 4 SHA-256 authority-set commitment (VX_AIR_SHA_CHAIN), 7 / 8 / 9 SHA-256 Merkle trees of 256 /
 * 512 / 16 leaves, 10 / 12 Ed25519 (2^17 / 2^16 rows), 11 / 14 / 13 SHA-512 (2^16 / 2^15 / 2^10 rows), 15 epoch-end log. */
enum { VX_AIR_FIBONACCI = 1, VX_AIR_MIX = 2, VX_AIR_LOOKUP = 5 };
int32_t vx_stark_default_config(vx_stark_config* cfg);
/* Run-time AIR descriptor (SURVEY 8b `vx_air_desc`): the constraint system of a starky-style AIR as a straight-line program over a
 * register file, the form a Rust host would lower `Stark::eval_packed_generic` / `eval_ext_circuit` to (starky v0.2.0 stark.rs; the
 * reference reaches it through curta's AirParser).  The same program is run by the GPU quotient kernel (one LDE point per lane,
 * registers in LDS), by the host verifier (at zeta, in the quadratic extension) and, restated, by oracle/air_program.py.
 * One instruction = one uint64:  bits 0..7 opcode, 8..15 destination register d, 16..31 operand a, 32..47 operand b, 48..63 zero.
 *   LOC d, a      r[d] = local row, column a            NXT d, a     r[d] = next row, column a
 *   PER d, a      r[d] = periodic column a              PUB d, a     r[d] = public input a
 *   CONST d, a    r[d] = consts[a]                      ADD / SUB / MUL d, a, b   r[d] = r[a] op r[b]
 *   CHAL d, a     r[d] = lookup challenge a          APUB d, a    r[d] = word a of the values published with the auxiliary cap
 *   ASSERT a            r[a] = 0 on EVERY row (the wrap-around pair included)
 *   ASSERT_TRANSITION a r[a] (x - w^-1): every row pair but the wrap-around      ASSERT_FIRST a / ASSERT_LAST a: on that row only
 * Constraints are consumed in program order (the order is protocol, as for the compiled AIRs).  Registration checks every operand,
 * that no register is read before it is written, that constants and periodic values are canonical, and the DEGREE of every
 * asserted expression (columns and periodic columns count 1; ASSERT <= 3, the three others <= 2: rate_bits 1 carries degree 3).
 * Periodic column q has 2^periodic_log[q] values (<= 2^16), given back to back.
 * AUXILIARY ROUND (lookup / logUp arguments, starky's `lookups()`): a program may declare aux_cols further columns, n_challenges
 * base-field challenges (<= 8; extension elements are pairs) and n_aux_public published extension values (<= 4).  After the trace
 * cap the prover draws the challenges and calls gen_aux -- the HOST's witness generator for this round: trace = the main columns
 * [cols][2^log_n] and aux_out = [aux_cols][2^log_n], both resident in HBM as buffers of `ctx` (use the library's primitives or
 * the host's own kernels on vx_buf_devptr, on the ctx stream); aux_public_out receives 2 * n_aux_public words.  The rows the
 * program sees hold the main columns followed by the auxiliary ones (LOC / NXT a >= cols).  Extension-field arithmetic
 * (F[X]/(X^2 - 7)) is written out in base-field instructions by whoever writes the program (air_program.py does).  A program with
 * aux_cols > 0 and gen_aux NULL can be verified, not proven.
 * A program is immutable once registered and its id
 * (>= VX_AIR_USER_BASE, never reused) works wherever an AIR id does: vx_quotient_eval, vx_stark_aux_trace, vx_stark_proof_bound,
 * vx_stark_prove, vx_stark_verify.  Registration is process-wide, thread-safe, needs no GPU; a verifier must register the SAME
 * program (the proof does not carry it).  vx_air_unregister retires the id. */
enum { VX_AIRP_LOC = 1, VX_AIRP_NXT = 2, VX_AIRP_PER = 3, VX_AIRP_PUB = 4, VX_AIRP_CONST = 5, VX_AIRP_ADD = 6, VX_AIRP_SUB = 7, VX_AIRP_MUL = 8,
       VX_AIRP_ASSERT = 9, VX_AIRP_ASSERT_TRANSITION = 10, VX_AIRP_ASSERT_FIRST = 11, VX_AIRP_ASSERT_LAST = 12, VX_AIRP_CHAL = 13, VX_AIRP_APUB = 14 };
enum { VX_AIR_USER_BASE = 4096, VX_AIRP_MAX_REGS = 32, VX_AIRP_MAX_COLS = 65535, VX_AIRP_MAX_CODE = 1 << 20, VX_AIRP_MAX_PERIOD_LOG = 16 };
typedef int32_t (*vx_air_gen_aux_fn)(void* user, vx_ctx* ctx, const vx_buf* trace, int log_n, const uint64_t* challenges,
                                     const uint64_t* public_inputs, vx_buf* aux_out, uint64_t* aux_public_out);
typedef struct vx_air_program {
    uint32_t cols, n_public, n_periodic, n_regs;
    const uint8_t* periodic_log;      /* [n_periodic] */
    const uint64_t* periodic_values;  /* sum of 2^periodic_log[q] values */
    const uint64_t* consts;
    uint32_t n_consts;
    const uint64_t* code;
    uint32_t n_code;
    uint32_t aux_cols, n_challenges, n_aux_public; /* auxiliary round: all zero = none */
    vx_air_gen_aux_fn gen_aux;
    void* gen_aux_user;
} vx_air_program;
int32_t vx_air_register(const vx_air_program* program, int* air_id, char* err, size_t errlen);
int32_t vx_air_unregister(int air_id);
/* The witness of PoseidonAir -- the Poseidon permutation as a STARK table (0-kno-vectorx_amd/air_library.py poseidon_builder: 48
 * columns, one round per row, 32 rows per permutation; the constraint program ships with the library and is registered by the
 * host): states = n_perm x 12 input words, trace_out = [48][32 * n_perm] column-major with the state, x^2, x^4 and x^7 of every
 * round row; rows 30 / 31 of a block hold the permutation's output.  n_perm a power of two.  What a recursive verifier's hashing
 * table is filled with (plonky2 v0.2.0 hash/poseidon.rs; reached from Circuit::prove's recursion, circuits/header_range.rs:167). */
int32_t vx_poseidon_air_trace(vx_ctx* ctx, const vx_buf* states, size_t n_perm, vx_buf* trace_out);
/* K5: batched constraint / quotient-polynomial evaluation (starky prover.rs compute_quotient_polys) for an AIR compiled
 * into the library or registered as a program.  trace_lde: column-major [cols][N], N = 2^(log_n + rate_bits), natural order, values on the coset
 * 7 * <w_N>.  out[k*N + i] = (sum_j alpha_k^(K-1-j) c_j(x_i)) / Z_H(x_i) for the two challenges k = 0, 1. */
int32_t vx_quotient_eval(vx_ctx* ctx, int air_id, int rate_bits, const vx_buf* trace_lde, int log_n, const uint64_t alphas[2],
                         const uint64_t* public_inputs, size_t n_public, vx_buf* out);
/* K9: the partial products of Plonk's permutation argument, as plonky2 computes them (plonky2 v0.2.0 plonk/prover.rs
 * wires_permutation_partial_products_and_zs, util/partial_products.rs; reached from Circuit::prove, circuits/header_range.rs:167).
 * Rows x_i = g^i of the subgroup of order 2^log_n; wires / sigmas: [n_routed][2^log_n] column-major (sigmas = the values of the
 * sigma polynomials, coset shifts included); k_is: the n_routed coset shifts (host); one challenge pair (beta, gamma).
 *   q_j(i) = (w_j(i) + beta k_j x_i + gamma) / (w_j(i) + beta s_j(i) + gamma),  chunk_c = product of `chunk` consecutive q_j,
 *   out column 0 = Z (Z(x_0) = 1, Z(x_(i+1)) = Z(x_i) * all chunks of row i), column t + 1 = Z(x_i) chunk_0(i) .. chunk_t(i)
 *   for t < m - 1, m = ceil(n_routed / chunk) <= 32 columns in all.  This library's own provers are STARK-only and do not
 * call it (no Plonk layer exists here): it is the primitive a patched plonky2 would forward to. */
int32_t vx_partial_products(vx_ctx* ctx, const vx_buf* wires, const vx_buf* sigmas, int log_n, size_t n_routed, const uint64_t* k_is,
                            uint64_t beta, uint64_t gamma, size_t chunk, vx_buf* out);
/* The auxiliary (lookup / logUp) columns an AIR derives from its trace once the lookup challenges are known -- the step
 * vx_stark_prove runs between the trace cap and the constraint challenges, exposed on its own as a test surface.
 * trace: the AIR's main columns [cols][2^log_n]; challenges: the AIR's CHAL base-field elements (canonical);
 * aux_out: [aux cols][2^log_n]; aux_public_out (may be NULL): the values published with the auxiliary cap. */
int32_t vx_stark_aux_trace(vx_ctx* ctx, int air_id, const vx_buf* trace, int log_n, const uint64_t* public_inputs, size_t n_public,
                           const uint64_t* challenges, size_t n_challenges, vx_buf* aux_out, uint64_t* aux_public_out);
/* upper bound on the proof length (uint64 words) for buffer sizing */
int32_t vx_stark_proof_bound(int air_id, const vx_stark_config* cfg, int log_n, size_t* n_words);
int32_t vx_stark_prove(vx_ctx* ctx, int air_id, const vx_stark_config* cfg, const vx_buf* trace, int log_n,
                       const uint64_t* public_inputs, size_t n_public, uint64_t* proof_out, size_t proof_cap,
                       size_t* proof_len);
/* `circuit.verify` (circuits/header_range.rs:170): host-side verification of a proof produced by
 * vx_stark_prove.  expect_air 0 = any; expect_public may be NULL.  VX_OK, or VX_ERR_STATEMENT with
 * a reason written to err (optional buffer).  Needs no GPU and no ctx. */
int32_t vx_stark_verify(const vx_stark_config* cfg, const uint64_t* proof, size_t proof_len, int expect_air,
                        const uint64_t* expect_public, size_t n_expect_public, char* err, size_t errlen);

/* ---- K8: witness hashing for the header chain
 * (plonky2x curta_blake2b_variable via circuits/builder/header.rs:14-19; SimpleMerkleTree /
 *  sha256 via subchain_verification.rs:213-220, 268-274) */
/* headers: device bytes, n messages at `stride` bytes each (uint8 view of a vx_buf);
 * sizes: n uint32 on the HOST; digests_out: 32*n bytes on the HOST */
int32_t vx_blake2b_256_batch(vx_ctx* ctx, const vx_buf* msgs, size_t stride, const uint32_t* sizes, size_t n,
                             uint8_t* digests_out);
/* 64-byte -> 32-byte SHA-256 of n pairs (host in/out; device compute) */
int32_t vx_sha256_pairs(vx_ctx* ctx, const uint8_t* pairs64, size_t n, uint8_t* out32);

/* ---- K8: BlakeChainAir trace generation (the Blake2b witness behind hash_encoded_header,
 * circuits/builder/header.rs:14-19, and the parent-hash links of
 * circuits/builder/subchain_verification.rs:163-177).  headers as for vx_verify_subchain (stride a
 * multiple of 128).  Writes the column-major MAIN trace (745 columns x 2^log_n rows, 16 rows per
 * compression, padded with inactive blocks; byte cells + the multiplicities of the two 2^16-row XOR
 * lookup tables, so log_n >= 16) into trace_out, the 20 public inputs (trusted hash, target hash as
 * 32-bit little-endian limbs, first and last block number, tree_size, bus flag) and optionally the digests
 * (host).  tree_size = 16 / 256 / 512: the state roots (decoder.rs:121-128) and data roots (:132-149) of the
 * headers go onto the logUp bus towards a ShaTreeAir of that many leaves (vx_header_range_prove proves both
 * tables under shared challenges); tree_size = 0: a stand-alone hash-chain proof.  window_length > 0 (one header, tree_size 0;
 * rotate): bytes [window_offset, window_offset + window_length) of the header go onto the bus instead, towards the
 * epoch-end table (vx_epoch_end_trace); public input 19 is the bus mode (0 / 1 / 2), 18 the tree size or the offset.  The 276
 * auxiliary (logUp) columns are derived inside vx_stark_prove once the lookup challenges exist.
 * The AIR proves that header i carries block number first_block_number + i as a SCALE compact int in
 * whichever of the four modes that number takes (decoder.rs:39-92) and reads the state root right behind it.
 * KNOWN DEVIATION: a header shorter than 104 bytes (whose state root and data root would share trace rows) is
 * refused with VX_ERR_ARG; parent hash + number + three 32-byte roots already exceed that in every Avail header.
 * Prove it with vx_stark_prove(ctx, VX_AIR_BLAKE_CHAIN, ...). */
enum { VX_AIR_BLAKE_CHAIN = 6, VX_BLAKE_AIR_COLS = 745, VX_BLAKE_AIR_AUX_COLS = 276 };
int32_t vx_blake_chain_trace(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes, size_t n_headers,
                             const uint8_t trusted_hash[32], uint32_t first_block_number, uint32_t tree_size, uint32_t window_offset, uint32_t window_length,
                             int log_n, vx_buf* trace_out, uint64_t public_inputs_out[20], uint8_t* digests_out);

/* ---- K8: ShaChainAir trace generation (compute_authority_set_commitment, justification.rs:127-162):
 * the chained SHA-256 commitment h_0 = SHA256(pk_0), h_i = SHA256(h_{i-1} || pk_i) over n_keys 32-byte
 * keys (host buffer).  Writes the 414-column trace (64 rows per compression, 2*n_keys - 1
 * compressions, padded with idle blocks), the 10 public inputs (the commitment as big-endian words, the
 * number of keys, bus_on) and optionally the 32 commitment bytes.  signed_flags (may be NULL = none) marks
 * the keys whose signatures the EdDSA table verifies; with bus_on they are sent to it over the lookup bus
 * (bus_on = 0: a stand-alone proof).  Prove with vx_stark_prove(ctx, VX_AIR_SHA_CHAIN, ...). */
enum { VX_AIR_SHA_CHAIN = 4, VX_SHA_AIR_COLS = 414, VX_SHA_AIR_AUX_COLS = 4, VX_SHA_TREE_AIR_COLS = 412 /* the Merkle AIRs 7 / 8 / 9 */ };
int32_t vx_sha_chain_trace(vx_ctx* ctx, const uint8_t* pubkeys, size_t n_keys, const uint8_t* signed_flags, uint32_t bus_on, int log_n, vx_buf* trace_out,
                           uint64_t public_inputs_out[10], uint8_t commitment_out[32]);

/* ---- K8: EdAir trace generation (the curve half of the 300 conditional EdDSA verifications, justification.rs:229-243):
 * n_signatures authorities (pubkey 32 B, signature R || S 64 B, flag), all over the same message (the 53-byte precommit);
 * the flagged ones take the slots of the table in order (slot s = the s-th flagged authority; its index rides along for
 * the bus), 256 rows per slot, at least one slot stays idle: 2^16 rows serve up to 255 signatures -- 2/3 of 300.  Writes
 * the 839-column trace and the 2 public inputs (number of signatures, bus_on); VX_ERR_STATEMENT when one does not verify.  bus_on = 0 makes a stand-alone table (nothing sent to the SHA-512 / authority-set tables).
 * Prove with vx_stark_prove(ctx, VX_AIR_ED25519 (2^17 rows) or VX_AIR_ED25519_16 (2^16 rows), ...). */
enum { VX_AIR_ED25519 = 10, VX_AIR_ED25519_16 = 12, VX_ED_AIR_COLS = 839, VX_ED_AIR_AUX_COLS = 688 };
int32_t vx_ed_trace(vx_ctx* ctx, const uint8_t* pubkeys, const uint8_t* signatures, const uint8_t* message, uint32_t message_len, const uint8_t* signed_flags,
                    size_t n_signatures, int log_n, uint32_t bus_on, vx_buf* trace_out, uint64_t public_inputs_out[2]);

/* ---- K8: Sha512Air trace generation (the hash half of the same verifications): H_i = SHA-512(R_i || A_i || message) for
 * every flagged authority, in compact slots of 160 rows in EdAir's order (2^15 rows hold 204 slots).  Writes the
 * 801-column trace and the 15 public inputs (message words 8..14 of the first block as (lo, hi) halves, bus_on).
 * message_len must be 53 (the precommit).  Prove with vx_stark_prove(ctx, VX_AIR_SHA512 (2^16 rows), VX_AIR_SHA512_15
 * (2^15) or VX_AIR_SHA512_10 (2^10), ...). */
enum { VX_AIR_SHA512 = 11, VX_AIR_SHA512_10 = 13, VX_AIR_SHA512_15 = 14, VX_SHA512_AIR_COLS = 801, VX_SHA512_AIR_AUX_COLS = 4 };
int32_t vx_sha512_trace(vx_ctx* ctx, const uint8_t* pubkeys, const uint8_t* signatures, const uint8_t* message, uint32_t message_len, const uint8_t* signed_flags,
                        size_t n_signatures, int log_n, uint32_t bus_on, vx_buf* trace_out, uint64_t public_inputs_out[15]);

/* ---- K8: EpochEndAir trace generation (verify_epoch_end_header in-proof, circuits/builder/rotate.rs:74-276): the
 * ScheduledChange log of the epoch-end header -- consensus flag 4, "FRNK", a SCALE compact length, flag 1, the compact
 * number of new authorities, num_authorities records (32-byte key, weight 1 as u64 LE), a zero u32 delay -- starting at byte
 * start_position + 1 of `header` (resident in HBM), one row per record in a 512-row table.  The table RECEIVES those bytes
 * over the logUp bus from the Blake2b table that hashes the header (vx_blake_chain_trace with window_offset =
 * start_position + 1, window_length = *window_length_out) and SENDS the keys to the new set's ShaChainAir (vx_sha_chain_trace
 * with bus_on = 2), so the new authority-set commitment is the commitment of exactly those header bytes.  Writes the 52-column
 * trace and the 10 public inputs (num_authorities, bus_on, one-hot byte lengths 1/2/4/5 of the two compact ints).
 * VX_ERR_STATEMENT when the prefix is not the log's; the records themselves are checked by the proof (and named by
 * vx_verify_epoch_end_header).  Prove with vx_stark_prove(ctx, VX_AIR_EPOCH_END, ..., log_n = 9). */
enum { VX_AIR_EPOCH_END = 15, VX_EPOCH_END_AIR_COLS = 52, VX_EPOCH_END_AIR_AUX_COLS = 46, VX_EPOCH_END_LOG_ROWS = 9 };
int32_t vx_epoch_end_trace(vx_ctx* ctx, const vx_buf* header, uint32_t start_position, uint32_t num_authorities, uint32_t bus_on, vx_buf* trace_out,
                           uint64_t public_inputs_out[10], uint32_t* window_length_out);

/* ---- statement level: verify_subchain (circuits/builder/subchain_verification.rs:56-303)
 * headers: n_fetched encoded headers (blocks trusted+1 .. target) resident in HBM at `stride`
 * bytes each (zero padded), sizes on the host.  max_headers = 256 / 512
 * (bin/header_range_{256,512}.rs:15).  Runs the map stage (Blake2b header hashes, header
 * decoding, link checks, 8-leaf SHA-256 roots) and the reduce tree on the GPU and returns the
 * 96-byte public output target_header_hash || state_root_merkle_root || data_root_merkle_root
 * (circuits/header_range.rs:56-58).  VX_ERR_STATEMENT if a chain rule is violated. */
int32_t vx_verify_subchain(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes,
                           size_t n_fetched, uint32_t max_headers, uint32_t trusted_block,
                           const uint8_t trusted_hash[32], uint32_t target_block, uint8_t out96[96]);

/* ---- decoders on their own (test surface for the reference's literal vectors, decoder.rs:238-249 and :388-395).
 * vx_decode_header_batch = decode_header (circuits/builder/decoder.rs:104-157) over n encoded headers in HBM (layout as
 * vx_verify_subchain): block number and mode of the SCALE compact int at byte 32 (decode_compact_int, :39-92; all four
 * modes), ok = 0 where the mode-3 upper-bits assertion (:83-89) fails, parent hash = bytes 0..32, state root at offset
 * 33 / 34 / 36 / 37 by mode (:121-128), data root = the 32 bytes at size - 32, or at 0 for a size-0 padding header
 * (:132-149).  Outputs are host arrays of n (x 32) entries.
 * vx_decode_precommit_batch = decode_precommit (:159-200) over n 53-byte messages (host in / host out, device compute):
 * ok = (byte 0 == 1), block hash [1..33), LE u32 block number, LE u64 round, LE u64 authority set id. */
int32_t vx_decode_header_batch(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes, size_t n, uint32_t* numbers_out,
                               uint8_t* modes_out, uint8_t* ok_out, uint8_t* parent_out, uint8_t* state_root_out, uint8_t* data_root_out);
int32_t vx_decode_precommit_batch(vx_ctx* ctx, const uint8_t* precommits, size_t n, uint8_t* ok_out, uint8_t* hash_out, uint32_t* block_number_out,
                                  uint64_t* round_out, uint64_t* set_id_out);

/* ---- justification, statement level (circuits/builder/justification.rs:195-257 and the hint's native
 * checks :29-83 -> circuits/input/mod.rs:241-260).  All byte arrays are host buffers.
 * vx_ed25519_verify_batch: ok_out[i] = 1 valid, 0 invalid, 2 skipped (enabled[i] == 0); semantics of
 * ed25519-dalek `verify` (canonical s, compress([s]B - [k]A) == R), the reference's verify_signature.
 * vx_verify_simple_justification: authority-set commitment == authority_set_hash, precommit matches
 * (block number, set id, block hash), every validator marked signed has a valid signature over the
 * precommit, signed * 3 > num_authorities * 2.  pubkeys / signatures / validator_signed have
 * max_authorities entries (32 / 64 / 1 bytes each).  VX_ERR_STATEMENT when any rule fails. */
int32_t vx_ed25519_verify_batch(vx_ctx* ctx, const uint8_t* pubkeys, const uint8_t* sigs, const uint8_t* msg, uint32_t msg_len,
                                const uint8_t* enabled, size_t n, uint8_t* ok_out);
int32_t vx_verify_simple_justification(vx_ctx* ctx, uint32_t block_number, const uint8_t block_hash[32],
                                       uint64_t authority_set_id, const uint8_t authority_set_hash[32],
                                       const uint8_t precommit[53], const uint8_t* pubkeys, const uint8_t* signatures,
                                       const uint8_t* validator_signed, uint32_t num_authorities, uint32_t max_authorities);

/* ---- top level: HeaderRangeCircuit::prove (circuits/header_range.rs:26-59 via Circuit::prove, :167).
 * Inputs as vx_verify_subchain.  Output blob (uint64 words): the magic "HRRANGE6" (VX_HR_BLOB_MAGIC), max_headers,
 * trusted_block, target_block, the 96 public output bytes (12 words), word 16 = S, the number of MAP SEGMENTS of the hash-chain
 * table (1 = one table over the whole range), words 17..20 = the lengths of the authority-set commitment, Merkle, Ed25519 and
 * SHA-512 proofs, word 21 = the round of the precommit, words 22 .. 22 + S = the lengths of the S hash-chain proofs: a header
 * of VX_HR_BLOB_FIXED_WORDS + S words; then the proofs -- the S segments in range order (BlakeChainAir), commitment
 * (ShaChainAir), SHA-256 Merkle trees (ShaTreeAir), Ed25519 (EdAir), SHA-512 (Sha512Air); the commitment / Ed25519 / SHA-512
 * proofs are empty when no justification was given.  EVERY statement of the circuit is inside these STARKs:
 * the tables share one logUp bus under challenges drawn after all their trace caps; the Merkle table's public inputs are the
 * two roots and the number of headers, which forces it to take every header's roots from the bus.
 * MAP SEGMENTS (the reference's MapReduce jobs, circuits/builder/subchain_verification.rs:72-79, 81-232): segment s proves the
 * headers of a contiguous part of the range -- it starts from the hash segment s - 1 ends with, numbers its blocks on and counts
 * its Merkle leaves from the first block of the range; the verifier checks those links between the segments' public inputs
 * (the reference's reduce step, :233-289, without recursion).  Segments are independent tables: vx_header_range_prove_ex can
 * prove a subset of the tables on this GPU (n_shards > 1, one process per GPU) -- the shards of one proof exchange their trace
 * caps once (vx_hr_exchange) and vx_header_range_merge puts their partial blobs together.  Not done: the proofs of a request
 * are not aggregated into one succinct proof (S segments open S times the columns: a segmented blob is larger). */
#define VX_HR_BLOB_MAGIC 0x3645474e41525248ULL /* "HRRANGE6" little-endian */
enum { VX_HR_BLOB_FIXED_WORDS = 22, VX_HR_MAX_SEGMENTS = 64 };
/* The justification witness of circuits/vars.rs:40-46 (host buffers; what HintSimpleJustification
 * returns, justification.rs:69-82) plus the two EVM inputs it is checked against. */
typedef struct vx_justification {
    uint64_t authority_set_id;
    const uint8_t* authority_set_hash; /* 32 bytes */
    const uint8_t* precommit;          /* 53 bytes */
    const uint8_t* pubkeys;            /* max_authorities x 32 */
    const uint8_t* signatures;         /* max_authorities x 64 */
    const uint8_t* validator_signed;   /* max_authorities x 1 */
    uint32_t num_authorities, max_authorities;
} vx_justification;
int32_t vx_header_range_proof_bound(const vx_stark_config* cfg, size_t n_chunks, size_t n_authorities, size_t* n_words);
/* just may be NULL (subchain only); otherwise verify_simple_justification for (target_block, target
 * header hash) is checked on the GPU before proving, as HeaderRangeCircuit::define does (:49-54). */
int32_t vx_header_range_prove(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes, size_t n_fetched,
                              uint32_t max_headers, uint32_t trusted_block, const uint8_t trusted_hash[32],
                              uint32_t target_block, const vx_justification* just, const vx_stark_config* cfg,
                              uint8_t out96[96], uint64_t* proof_out, size_t proof_cap, size_t* proof_len);
/* The same proof with the hash-chain table split into n_segments map segments (1 .. VX_HR_MAX_SEGMENTS, at most one per
 * header; a segment is never smaller than the 2^16-row lookup tables, so more segments than compressions / 4096 only add
 * padding), and optionally only the tables of shard `shard` of `n_shards`: table t in bus order (segments, Merkle, commitment,
 * Ed25519, SHA-512) belongs to shard t mod n_shards.  With n_shards > 1 every shard calls this function on its own GPU with the
 * SAME inputs; `exchange` is called once per proof on each shard, from a prover thread, and must return the element-wise
 * (wrapping) SUM over the shards of the word array it is given (an all-reduce: every shard fills only the slots of its own
 * tables -- public inputs and trace cap -- so the sum is the union).  The blob then holds the local proofs only (the other
 * lengths are 0); vx_header_range_merge makes the request's blob from the shards' blobs.  n_shards = 1: exchange is not used. */
typedef struct vx_hr_exchange {
    int32_t (*fn)(void* user, uint64_t* words, size_t n_words); /* 0 = ok */
    void* user;
} vx_hr_exchange;
int32_t vx_header_range_proof_bound_ex(const vx_stark_config* cfg, size_t n_chunks, size_t n_authorities, uint32_t n_segments, size_t* n_words);
int32_t vx_header_range_prove_ex(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes, size_t n_fetched,
                                 uint32_t max_headers, uint32_t trusted_block, const uint8_t trusted_hash[32],
                                 uint32_t target_block, const vx_justification* just, const vx_stark_config* cfg,
                                 uint32_t n_segments, uint32_t shard, uint32_t n_shards, const vx_hr_exchange* exchange,
                                 uint8_t out96[96], uint64_t* proof_out, size_t proof_cap, size_t* proof_len);
/* blobs of the n_shards shards of ONE proof (same request, same segment count) -> the request's blob; host only */
int32_t vx_header_range_merge(const uint64_t** blobs, const size_t* blob_lens, size_t n_blobs, uint64_t* out, size_t out_cap, size_t* out_len,
                              char* err, size_t errlen);
/* HeaderRangeCircuit verify: checks the blob of vx_header_range_prove against the request (the five fields of the
 * circuit's 80-byte EVM input, header_range.rs:30-37) and the claimed 96 output bytes: every table's STARK under the
 * shared lookup challenges, the bus balance, and -- when the blob carries a justification -- that the committed
 * authority set is authority_set_hash, that more than 2/3 of it signed, and that what it signed is the precommit for
 * (target header hash, target_block, authority_set_id).  authority_set_hash = NULL for a blob proven without one. */
int32_t vx_header_range_verify(const vx_stark_config* cfg, const uint64_t* blob, size_t blob_len, uint32_t max_headers,
                               uint32_t trusted_block, const uint8_t trusted_hash[32], uint64_t authority_set_id,
                               const uint8_t* authority_set_hash /* 32 B or NULL */, uint32_t target_block, const uint8_t out96[96], char* err,
                               size_t errlen);

/* ---- RotateCircuit: circuits/rotate.rs:80-109, circuits/builder/rotate.rs:74-323 ----
 * EVM input = u64 authority_set_id || bytes32 authority_set_hash (40 B, dummy_rotate.rs:11-14);
 * output = bytes32 new authority set hash.  The RotateHint witness (rotate.rs:29-63, vars.rs RotateStruct)
 * is passed flat: the epoch-end header (device buffer of >= MAX_HEADER_SIZE bytes, zero padded past
 * header_size, as get_header_rotate builds it, input/mod.rs:848-858), its size, the epoch-end block number,
 * target_header_num_authorities, next_authority_set_start_position and the new pubkeys (host, n x 32). */
/* verify_epoch_end_header (builder/rotate.rs:176-276) on the GPU, one lane per validator slot. */
int32_t vx_verify_epoch_end_header(vx_ctx* ctx, const vx_buf* header, uint32_t num_authorities, uint32_t start_position,
                                   const uint8_t* new_pubkeys, uint32_t max_authorities);
int32_t vx_rotate_proof_bound(const vx_stark_config* cfg, size_t n_chunks, size_t n_cur_authorities, size_t n_new_authorities,
                              size_t* n_words);
/* RotateMethods::rotate (builder/rotate.rs:278-323) + Circuit::prove.  Six STARKs on two logUp buses: (A) the justification
 * by the current set (`just`, required): its commitment, the Ed25519 and the SHA-512 tables; (B) the Blake2b header hash, which
 * sends the bytes behind start_position to the epoch-end table (verify_epoch_end_header in-proof), which sends the keys it
 * reads there to the new set's commitment -- out32.  Every rule is also checked natively first and named in the error.
 * The log must lie inside the first header_size bytes (the hashed ones) and behind byte 72.
 * VX_ERR_STATEMENT where the reference circuit's assertions (or the hint's panics) would fail. */
int32_t vx_rotate_prove(vx_ctx* ctx, const vx_buf* header, uint32_t header_size, uint32_t epoch_end_block_number,
                        uint32_t num_authorities, uint32_t start_position, const uint8_t* new_pubkeys,
                        const vx_justification* just, const vx_stark_config* cfg, uint8_t out32[32], uint64_t* proof_out,
                        size_t proof_cap, size_t* proof_len);
int32_t vx_rotate_verify(const vx_stark_config* cfg, const uint64_t* blob, size_t blob_len, uint64_t authority_set_id,
                         const uint8_t authority_set_hash[32], const uint8_t out32[32], char* err, size_t errlen);

/* ---- multi-GPU: proofs are independent, one per rank; the only exchange is this gather ----
 * All-gathers `n_words` u64 words per rank over RCCL on the ctx stream: out (host, world * n_words words, rank
 * order) is filled on EVERY rank; blobs must have the same length on all ranks (pad to the proof bound).
 * `nccl_comm` is an ncclComm_t the caller created (ncclCommInitRank, one rank per GPU); the RCCL symbols are
 * looked up in the process at call time, so the library has no link-time dependency on RCCL and uses the same
 * copy the caller's communicator belongs to.  VX_ERR_DEVICE if RCCL is not loaded or the collective fails. */
int32_t vx_gather_proofs(vx_ctx* ctx, void* nccl_comm, int world, const uint64_t* mine, size_t n_words, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* VX_H */
