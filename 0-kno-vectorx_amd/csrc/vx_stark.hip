// Generic STARK prover on the GPU primitives (K5 quotient evaluation, K7 transcript, K6 FRI
// commit + query phases, proof serialisation).
//
// Protocol: starky v0.2.0 `prove_with_commitment` over plonky2 v0.2.0's PolynomialBatch /
// FRI (crates pinned at /root/reference Cargo.lock:4848-4905; the reference reaches them
// through every `circuit.prove`, circuits/header_range.rs:167, and through curta's STARKs,
// circuits/builder/header.rs:18).  Same transcript order, same FRI (ConstantArityBits(4,5),
// cap height 4, 84 queries, 16 PoW bits at rate_bits 1), but organised for the GPU:
//   - nothing is transposed: LDEs stay column-major, leaves are rows read in place;
//   - openings at zeta come from barycentric dot products over the trace values, not from
//     coefficient Horner loops;
//   - the FRI batch polynomial is assembled and folded in EVALUATION space from the committed
//     LDE values (the CPU prover works on coefficients and re-runs an FFT per layer).
// The oracle (oracle/stark_ref.py) restates the CPU prover's coefficient-space algorithm; proofs
// must match byte for byte (tests/test_gpu_stark.py).
#include <string.h>

#include <map>
#include <mutex>

#include "air.cuh"
#include "air_program.h"
#include "air_blake.cuh"
#include "air_sha.cuh"
#include "air_ed.cuh"
#include "air_epoch.cuh"
#include "air_sha512.cuh"
#include "air_sha_tree.cuh"
#include "glh_poseidon.h"
#include "vx_internal.h"

// ------------------------------------------------------------------ challenger over the host Poseidon (glh_poseidon.h)
namespace {
inline void h_poseidon(uint64_t* s) { glh::poseidon(s); }
struct Ext {
    uint64_t a, b;
};
inline Ext e_add(Ext x, Ext y) { return {glh::add(x.a, y.a), glh::add(x.b, y.b)}; }
inline Ext e_sub(Ext x, Ext y) { return {glh::sub(x.a, y.a), glh::sub(x.b, y.b)}; }
inline Ext e_mul(Ext x, Ext y) {
    return {glh::add(glh::mul(x.a, y.a), glh::mul(7, glh::mul(x.b, y.b))), glh::add(glh::mul(x.a, y.b), glh::mul(x.b, y.a))};
}
inline Ext e_scale(Ext x, uint64_t s) { return {glh::mul(x.a, s), glh::mul(x.b, s)}; }
inline Ext e_inv(Ext x) {
    uint64_t n = glh::sub(glh::mul(x.a, x.a), glh::mul(7, glh::mul(x.b, x.b)));
    uint64_t ni = glh::inv(n);
    return {glh::mul(x.a, ni), glh::mul(glh::sub(0, x.b), ni)};
}
inline Ext e_pow(Ext x, uint64_t e) {
    Ext r{1, 0};
    while (e) {
        if (e & 1) r = e_mul(r, x);
        x = e_mul(x, x);
        e >>= 1;
    }
    return r;
}
struct Challenger : glh::Challenger {  // plonky2 iop/challenger.rs (glh_poseidon.h)
    Ext ext_challenge() {
        uint64_t a = challenge(), b = challenge();
        return {a, b};
    }
};
}  // namespace

// ------------------------------------------------------------------ kernels
__device__ __forceinline__ uint64_t root_pow_f(const uint64_t* tw, uint64_t e, int log_s) {
    uint64_t E = (e << (32 - log_s)) & 0xFFFFFFFFULL;
    uint64_t r = tw[4096 + (E >> 22)];
    if (log_s > 10) r = gl_mul(r, tw[2048 + ((E >> 11) & 2047)]);
    if (log_s > 21) r = gl_mul(r, tw[E & 2047]);
    return r;
}

struct QuotArgs {
    const uint64_t* lde;   // [COLS][N]
    uint64_t* q_out;       // [2][N]
    int log_N, rate_bits;
    uint64_t shift, last, n_inv;  // coset shift g, w_n^-1, 1/n
    uint64_t alpha[2];
    uint64_t zh_inv[8];    // 1 / (g^n * w_{2^r}^k - 1), k < 2^r
    uint64_t zh[8];        // g^n * w_{2^r}^k - 1
    const uint64_t* periodic;  // periodic column q on the LDE coset: (1 << (plog(q) + rate_bits)) values, columns back to back
    const uint64_t* pub;       // [PUB] (device)
    const uint64_t* tw;        // forward w_{2^32} power table
    const uint64_t* apow;      // [2K]: apow[2k + j] = alpha_j^(K-1-k), K = the AIR's constraint count (device)
    uint64_t chal[8];          // auxiliary-round challenges (beta, gamma, ...)
    uint64_t apub[8];          // values published with the auxiliary cap
};

#ifndef VX_Q_WAVES
#define VX_Q_WAVES 4
#endif
#ifndef VX_Q_WAVES_MAX
#define VX_Q_WAVES_MAX 8
#endif
#ifndef VX_Q_BLOCK
#define VX_Q_BLOCK 256
#endif
constexpr int QB = VX_Q_BLOCK;
// one block = a tile of QB * R consecutive LDE points, lane t holding points t, t + QB, ... of the tile
template <class Air, int R>
__global__ __launch_bounds__(QB) __attribute__((amdgpu_waves_per_eu(VX_Q_WAVES, VX_Q_WAVES_MAX))) void k_quotient(QuotArgs a) {
    using F = FpN<R>;
    const size_t N = (size_t)1 << a.log_N;
    const size_t i0 = blockIdx.x * (size_t)(QB * R) + threadIdx.x;
    const size_t step = (size_t)1 << a.rate_bits;
    Consumer<F> c;
    c.init(a.apow);
    RowViewN<R> loc{a.lde, N, {}}, nxt{a.lde, N, {}};
    F per[Air::PERIODIC > 0 ? Air::PERIODIC : 1], pub[Air::PUB > 0 ? Air::PUB : 1];
    F chal[Air::CHAL > 0 ? Air::CHAL : 1], apub[Air::AUXPUB > 0 ? 2 * Air::AUXPUB : 1];
    uint64_t zinv[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const size_t i = (i0 + QB * (size_t)j) & (N - 1);  // N >= QB * R is checked by the launcher
        const uint64_t x = gl_mul(a.shift, root_pow_f(a.tw, i, a.log_N));
        const int k = (int)(i & (step - 1));
        c.z_last.v[j] = gl_sub(x, a.last);
        // L_first = Z_H(x) / (n (x - 1)),  L_last = last * Z_H(x) / (n (x - last))
        const uint64_t zh_n = gl_mul(a.zh[k], a.n_inv);
        c.l_first.v[j] = gl_mul(zh_n, gl_inv(gl_sub(x, 1)));
        c.l_last.v[j] = gl_mul(gl_mul(zh_n, a.last), gl_inv(gl_sub(x, a.last)));
        zinv[j] = a.zh_inv[k];
        loc.i[j] = i;
        nxt.i[j] = (i + step) & (N - 1);
        size_t poff = 0;
#pragma unroll
        for (int q = 0; q < Air::PERIODIC; ++q) {
            const size_t plen = (size_t)1 << (Air::plog(q) + a.rate_bits);
            per[q].v[j] = a.periodic[poff + (i & (plen - 1))];
            poff += plen;
        }
    }
#pragma unroll
    for (int q = 0; q < Air::PUB; ++q) pub[q] = F::from(a.pub[q]);
#pragma unroll
    for (int q = 0; q < Air::CHAL; ++q) chal[q] = F::from(a.chal[q]);
#pragma unroll
    for (int q = 0; q < 2 * Air::AUXPUB; ++q) apub[q] = F::from(a.apub[q]);
    Air::template eval<F>(loc, nxt, per, pub, chal, apub, c);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        a.q_out[loc.i[j]] = gl_mul(c.result(0, j), zinv[j]);
        a.q_out[N + loc.i[j]] = gl_mul(c.result(1, j), zinv[j]);
    }
}

// The same evaluation for a registered constraint PROGRAM (include/vx.h vx_air_program; csrc/air_program.h): one LDE point per
// lane, the register file in LDS (register r of lane t at regs[r * QB + t]: conflict-free), instructions fetched with scalar
// loads (the program counter is wave-uniform).  Constraints go through the same consumer as the compiled AIRs, so a program
// that restates a compiled AIR yields the same quotient values.
struct ProgArgs {
    const uint64_t* code;    // device: n_code instruction words, then the constants
    const uint64_t* consts;
    const uint64_t* per_tab; // device: per periodic column (offset into QuotArgs::periodic, length - 1)
    int n_code;
};
__global__ __launch_bounds__(QB) void k_quotient_prog(QuotArgs a, ProgArgs p) {
    extern __shared__ uint64_t prog_regs[];
    using F = Fp;
    const size_t N = (size_t)1 << a.log_N;
    const size_t i = (blockIdx.x * (size_t)QB + threadIdx.x) & (N - 1);
    const size_t step = (size_t)1 << a.rate_bits, inext = (i + step) & (N - 1);
    Consumer<F> c;
    c.init(a.apow);
    const uint64_t x = gl_mul(a.shift, root_pow_f(a.tw, i, a.log_N));
    const int kz = (int)(i & (step - 1));
    c.z_last.v[0] = gl_sub(x, a.last);
    const uint64_t zh_n = gl_mul(a.zh[kz], a.n_inv);
    c.l_first.v[0] = gl_mul(zh_n, gl_inv(gl_sub(x, 1)));
    c.l_last.v[0] = gl_mul(gl_mul(zh_n, a.last), gl_inv(gl_sub(x, a.last)));
    uint64_t* R = prog_regs + threadIdx.x;
#pragma unroll 1
    for (int pc = 0; pc < p.n_code; ++pc) {
        const uint64_t w = p.code[pc];
        const int op = (int)(w & 0xFF), d = (int)((w >> 8) & 0xFF) * QB, ra = (int)((w >> 16) & 0xFFFF), rb = (int)((w >> 32) & 0xFFFF);
        switch (op) {
            case VX_AIRP_LOC: R[d] = a.lde[(size_t)ra * N + i]; break;
            case VX_AIRP_NXT: R[d] = a.lde[(size_t)ra * N + inext]; break;
            case VX_AIRP_PER: R[d] = a.periodic[p.per_tab[2 * ra] + (i & p.per_tab[2 * ra + 1])]; break;
            case VX_AIRP_PUB: R[d] = a.pub[ra]; break;
            case VX_AIRP_CONST: R[d] = p.consts[ra]; break;
            case VX_AIRP_CHAL: R[d] = a.chal[ra & 7]; break;
            case VX_AIRP_APUB: R[d] = a.apub[ra & 7]; break;
            case VX_AIRP_ADD: R[d] = gl_add(R[ra * QB], R[rb * QB]); break;
            case VX_AIRP_SUB: R[d] = gl_sub(R[ra * QB], R[rb * QB]); break;
            case VX_AIRP_MUL: R[d] = gl_mul(R[ra * QB], R[rb * QB]); break;
            case VX_AIRP_ASSERT: c.constraint(F{{R[ra * QB]}}); break;
            case VX_AIRP_ASSERT_TRANSITION: c.transition(F{{R[ra * QB]}}); break;
            case VX_AIRP_ASSERT_FIRST: c.first_row(F{{R[ra * QB]}}); break;
            default: c.last_row(F{{R[ra * QB]}}); break;  // VX_AIRP_ASSERT_LAST: registration admits nothing else
        }
    }
    a.q_out[i] = gl_mul(c.result(0, 0), a.zh_inv[kz]);
    a.q_out[N + i] = gl_mul(c.result(1, 0), a.zh_inv[kz]);
}

// Openings from the committed LDE: the n points x_i = g * w_n^i (LDE indices i << r) determine a polynomial
// of degree < n,  P(z) = (z^n - g^n) / (n g^n) * sum_i v_i * x_i / (z - x_i).
// w0[i] = x_i / (zeta - x_i), w1[i] = x_i / (w*zeta - x_i)
__global__ __launch_bounds__(256) void k_bary_weights(int log_n, uint64_t shift, gl2 zeta, gl2 zeta_next, const uint64_t* tw, uint64_t* w0,
                                                      uint64_t* w1) {
    size_t n = (size_t)1 << log_n;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t xi = gl_mul(shift, root_pow_f(tw, i, log_n));
    gl2 d0 = gl2_inv({gl_sub(zeta.a, xi), zeta.b});
    gl2 d1 = gl2_inv({gl_sub(zeta_next.a, xi), zeta_next.b});
    d0 = gl2_scale(d0, xi);
    d1 = gl2_scale(d1, xi);
    w0[2 * i] = d0.a;
    w0[2 * i + 1] = d0.b;
    w1[2 * i] = d1.a;
    w1[2 * i + 1] = d1.b;
}
// Consume mode: after the in-place inverse NTT the trace buffer holds coefficient k at position brev(k), so
// P(zeta) = sum_j buf[j] * zeta^brev(j): w0[j] = zeta^brev(j), w1[j] = (w zeta)^brev(j) from the tables of
// squarings zp[b] = z^(2^b) -- the same dot-product kernel then reads the n coefficients (half the bytes of
// every second LDE point).
struct PowBrevArgs {
    gl2 z0[32], z1[32];  // zeta^(2^b), (w zeta)^(2^b)
    int log_n;
};
__global__ __launch_bounds__(256) void k_pow_brev_weights(PowBrevArgs a, uint64_t* w0, uint64_t* w1) {
    const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= ((size_t)1 << a.log_n)) return;
    const uint32_t e = brev32((uint32_t)j, a.log_n);
    gl2 p0{1, 0}, p1{1, 0};
    for (int b = 0; b < a.log_n; ++b)
        if ((e >> b) & 1) {
            p0 = gl2_mul(p0, a.z0[b]);
            p1 = gl2_mul(p1, a.z1[b]);
        }
    w0[2 * j] = p0.a;
    w0[2 * j + 1] = p0.b;
    w1[2 * j] = p1.a;
    w1[2 * j + 1] = p1.b;
}
// out[col] = (sum_i V[col][i << log_step] * w0[i], sum_i V[col][i << log_step] * w1[i]).  One block per BARY_CPB
// columns: the two ext weights of an index (32 bytes) are loaded once and used for BARY_CPB values (8 bytes each) --
// with one column per block the weight stream, served by L2 / Infinity Cache, was 4x the data stream.
constexpr int BARY_CPB = 2;  // measured per proof: 1 -> 9.3 ms, 4 -> 12.7 ms
// gridDim.y = BARY_SPLIT blocks share a column pair (interleaved index ranges; partial sums to out + blockIdx.y * split_stride, added
// on the host with the openings it downloads anyway): 1021 columns / 2 gave 511 blocks = two waves per SIMD, a latency-bound loop.
#ifndef VX_BARY_SPLIT
#define VX_BARY_SPLIT 4
#endif
constexpr int BARY_SPLIT = VX_BARY_SPLIT;
__global__ __launch_bounds__(256) void k_bary_dot(const uint64_t* vals, size_t col_stride, int log_step, size_t n, size_t n_cols,
                                                  const uint64_t* w0, const uint64_t* w1, uint64_t* out, size_t split_stride) {
    __shared__ uint64_t red[256 * 4];
    const uint64_t* col[BARY_CPB];
    gl_acc acc[BARY_CPB][4];  // lazy sums (160-bit integers), reduced once per lane
#pragma unroll
    for (int q = 0; q < BARY_CPB; ++q) {
        const size_t cidx = blockIdx.x * (size_t)BARY_CPB + q;
        col[q] = vals + (cidx < n_cols ? cidx : n_cols - 1) * col_stride;  // tail block: recomputed, not written
#pragma unroll
        for (int e = 0; e < 4; ++e) gl_acc_zero(acc[q][e]);
    }
    out += blockIdx.y * split_stride;
    for (size_t i = blockIdx.y * (size_t)256 + threadIdx.x; i < n; i += (size_t)256 * gridDim.y) {
        const uint64_t a0 = w0[2 * i], b0 = w0[2 * i + 1], a1 = w1[2 * i], b1 = w1[2 * i + 1];
#pragma unroll
        for (int q = 0; q < BARY_CPB; ++q) {
            const uint64_t v = col[q][i << log_step];
            gl_mac(acc[q][0], v, a0);
            gl_mac(acc[q][1], v, b0);
            gl_mac(acc[q][2], v, a1);
            gl_mac(acc[q][3], v, b1);
        }
    }
#pragma unroll
    for (int q = 0; q < BARY_CPB; ++q) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) red[4 * threadIdx.x + e] = gl_acc_reduce(acc[q][e]);
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if ((int)threadIdx.x < h)
                for (int e = 0; e < 4; ++e) red[4 * threadIdx.x + e] = gl_add(red[4 * threadIdx.x + e], red[4 * (threadIdx.x + h) + e]);
            __syncthreads();
        }
        const size_t cidx = blockIdx.x * (size_t)BARY_CPB + q;
        if (threadIdx.x < 4 && cidx < n_cols) out[4 * cidx + threadIdx.x] = red[threadIdx.x];
    }
}

struct CombineArgs {
    const uint64_t* trace_lde;  // [c][N]
    const uint64_t* quot_lde;   // [nq][N]
    int n_cols, n_q, log_N;
    const uint64_t* alpha_pow;  // (c + nq) ext powers of the FRI alpha (device)
    gl2 alpha_c;                // alpha^c
    gl2 y0, y1;                 // reduced openings at zeta / w*zeta
    gl2 zeta, zeta_next;
    uint64_t shift;
    const uint64_t* tw;
    uint64_t* out;  // [N] ext
};
// final_poly(x) = alpha^c * (S0(x) - y0) / (x - zeta) + (S1(x) - y1) / (x - w zeta)
// (PolynomialBatch::prove_openings: batch 0 = trace ++ quotient at zeta, batch 1 = trace at w*zeta)
// One block = 256 * R consecutive LDE points, lane t holding points t, t + 256, ...: R independent loads per column
// and R x 2 KB of each column per visit (the walk over thousands of columns is bound by load latency / address
// translation otherwise).  The alpha-power sums are 160-bit integer multiply-accumulates reduced once (gl_mac).
template <int R>
__global__ __launch_bounds__(256) void k_fri_combine(CombineArgs a) {
    const size_t N = (size_t)1 << a.log_N;
    const size_t i0 = blockIdx.x * (size_t)(256 * R) + threadIdx.x;
    size_t idx[R];
    gl_acc sa[R], sb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        idx[r] = (i0 + 256 * (size_t)r) & (N - 1);  // R > 1 only when N >= 256 R (launcher); duplicates write equal values
        gl_acc_zero(sa[r]);
        gl_acc_zero(sb[r]);
    }
#pragma unroll 2
    for (int j = 0; j < a.n_cols; ++j) {
        const uint64_t pa = a.alpha_pow[2 * j], pb = a.alpha_pow[2 * j + 1];
        const uint64_t* col = a.trace_lde + (size_t)j * N;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t v = col[idx[r]];
            gl_mac(sa[r], v, pa);
            gl_mac(sb[r], v, pb);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const size_t i = idx[r];
        const gl2 s1{gl_acc_reduce(sa[r]), gl_acc_reduce(sb[r])};
        gl2 s0 = s1;
        for (int j = 0; j < a.n_q; ++j) {
            const int k = a.n_cols + j;
            s0 = gl2_add(s0, gl2_scale({a.alpha_pow[2 * k], a.alpha_pow[2 * k + 1]}, a.quot_lde[(size_t)j * N + i]));
        }
        const uint64_t x = gl_mul(a.shift, root_pow_f(a.tw, i, a.log_N));
        gl2 t0 = gl2_mul(gl2_sub(s0, a.y0), gl2_inv({gl_sub(x, a.zeta.a), gl_neg(a.zeta.b)}));
        gl2 t1 = gl2_mul(gl2_sub(s1, a.y1), gl2_inv({gl_sub(x, a.zeta_next.a), gl_neg(a.zeta_next.b)}));
        gl2 f = gl2_add(gl2_mul(a.alpha_c, t0), t1);
        a.out[2 * i] = f.a;
        a.out[2 * i + 1] = f.b;
    }
}

// ------------------------------------------------------------------ AIR registry
typedef int32_t (*gen_aux_fn)(vx_ctx*, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub);
struct AirDesc {
    int id, cols, pub, periodic, period_log, exact_log;
    void (*periodic_values)(std::vector<uint64_t>&);  // one period of every periodic column, back to back
    void (*launch)(QuotArgs&, hipStream_t);
    int (*count)();  // number of constraints eval pushes
    int aux, chal, auxpub;  // auxiliary round: columns, base-field challenges, published extension values
    int (*plog)(int);       // period (log2) of periodic column q
    gen_aux_fn gen_aux;     // trace + challenges -> auxiliary columns [aux][n] (+ 2*auxpub published words, host)
    const AirProgram* prog = nullptr;  // a registered constraint program (include/vx.h vx_air_register): launch / count / plog come from it
};
template <class Air>
static int count_q() {
    static const int k = [] {
        Consumer<CountF> c;
        CountRow r;
        CountF per[Air::PERIODIC > 0 ? Air::PERIODIC : 1], pub[Air::PUB > 0 ? Air::PUB : 1], chal[Air::CHAL > 0 ? Air::CHAL : 1], apub[Air::AUXPUB > 0 ? 2 * Air::AUXPUB : 1];
        Air::template eval<CountF>(r, r, per, pub, chal, apub, c);
        return c.k;
    }();
    return k;
}
template <class Air>
static void launch_q(QuotArgs& a, hipStream_t s) {
    constexpr int R = Air::QUOT_ROWS_PER_LANE;
    const size_t N = (size_t)1 << a.log_N;
    if (R > 1 && N >= QB * (size_t)R) hipLaunchKernelGGL((k_quotient<Air, R>), dim3((unsigned)(N / (QB * R))), dim3(QB), 0, s, a);
    else hipLaunchKernelGGL((k_quotient<Air, 1>), dim3((unsigned)((N + QB - 1) / QB)), dim3(QB), 0, s, a);
}
static void no_periodic(std::vector<uint64_t>& v) { v.clear(); }
static void mix_periodic(std::vector<uint64_t>& v) { v = {0, 0, 0, 1, 3, 5, 7, 11}; }
static void blake_periodic(std::vector<uint64_t>& v) {
    v.assign(16 * 16 + 4 * 65536, 0);
    for (int k = 0; k < 16; ++k) v[k * 16 + k] = 1;  // sel_k: one-hot on row k of every 16-row block
    for (uint64_t i = 0; i < 65536; ++i) {           // the XOR tables: row i = (a = i & 255, b = i >> 8)
        const uint64_t a = i & 255, b = i >> 8;
        v[256 + i] = a, v[256 + 65536 + i] = b, v[256 + 2 * 65536 + i] = (a ^ b) & 127, v[256 + 3 * 65536 + i] = (a ^ b) >> 7;
    }
}
static void lookup_periodic(std::vector<uint64_t>& v) {
    v.resize(3 * 256);
    for (int i = 0; i < 256; ++i) v[i] = i & 15, v[256 + i] = i >> 4, v[512 + i] = (i & 15) ^ (i >> 4);
}
template <class Air>
static AirDesc desc(void (*pv)(std::vector<uint64_t>&), gen_aux_fn ga = nullptr) {
    return {Air::ID, Air::COLS, Air::PUB, Air::PERIODIC, Air::PERIOD_LOG, Air::EXACT_LOG, pv, launch_q<Air>, count_q<Air>, Air::AUX, Air::CHAL, Air::AUXPUB, Air::plog, ga};
}
static const AirDesc AIRS[] = {
    desc<ShaAir>(ShaAir::periodic_values, vx_sha_chain_gen_aux), desc<BlakeAir>(blake_periodic, vx_blake_air_gen_aux), desc<FibAir>(no_periodic), desc<MixAir>(mix_periodic),
    desc<LookupAir>(lookup_periodic, vx_lookup_air_gen_aux),
    desc<ShaTreeAir256>(ShaTreeAir256::periodic_values, vx_sha_tree_gen_aux_256), desc<ShaTreeAir512>(ShaTreeAir512::periodic_values, vx_sha_tree_gen_aux_512),
    desc<ShaTreeAir16>(ShaTreeAir16::periodic_values, vx_sha_tree_gen_aux_16),
    desc<EdAir17>(EdAir17::periodic_values, vx_ed_air_gen_aux), desc<EdAir16>(EdAir16::periodic_values, vx_ed_air_gen_aux),
    desc<Sha512Air16>(Sha512Air16::periodic_values, vx_sha512_air_gen_aux), desc<Sha512Air10>(Sha512Air10::periodic_values, vx_sha512_air_gen_aux), desc<Sha512Air15>(Sha512Air15::periodic_values, vx_sha512_air_gen_aux),
    desc<EpochEndAir>(EpochEndAir::periodic_values, vx_epoch_air_gen_aux),
};
static std::mutex g_prog_desc_mu;
static std::map<int, AirDesc> g_prog_desc;  // descriptors of registered programs (node addresses are stable)
static const AirDesc* find_air(int id) {
    for (const AirDesc& d : AIRS)
        if (d.id == id) return &d;
    if (id >= VX_AIR_USER_BASE) {
        const AirProgram* pg = vx_air_program_find(id);
        if (!pg) return nullptr;
        std::lock_guard<std::mutex> lk(g_prog_desc_mu);
        auto it = g_prog_desc.find(id);
        if (it == g_prog_desc.end()) {
            AirDesc d{};
            d.id = id, d.cols = (int)pg->cols, d.pub = (int)pg->pub, d.periodic = (int)pg->plog.size(), d.period_log = pg->period_log, d.prog = pg;
            d.aux = (int)pg->aux, d.chal = (int)pg->chal, d.auxpub = (int)pg->auxpub;
            it = g_prog_desc.emplace(id, d).first;
        }
        return &it->second;
    }
    return nullptr;
}
static int air_plog(const AirDesc* air, int q) { return air->prog ? air->prog->plog[q] : air->plog(q); }
// the auxiliary columns of an AIR: the compiled generator, or the host's callback of a registered program (which sees the
// device arrays as buffers of the context)
static bool air_has_gen_aux(const AirDesc* air) { return air->prog ? air->prog->gen_aux != nullptr : air->gen_aux != nullptr; }
static int32_t air_gen_aux(vx_ctx* ctx, const AirDesc* air, uint64_t* trace_d, int L, const uint64_t* chal, const uint64_t* pub, uint64_t* aux_d, uint64_t* apub) {
    if (!air->prog) return air->gen_aux(ctx, trace_d, L, chal, pub, aux_d, apub);
    const size_t n = (size_t)1 << L;
    vx_buf tb{trace_d, n * (size_t)air->cols}, ab{aux_d, n * (size_t)air->aux};
    const int32_t rc = air->prog->gen_aux(air->prog->gen_aux_user, ctx, &tb, L, chal, pub, &ab, apub);
    if (rc != VX_OK) return vx_fail(ctx, rc, "air program %d: the auxiliary-round generator returned %d", air->id, rc);
    return VX_OK;
}

// values v[0..p) of a periodic column on the rows -> its values on the LDE coset:
// P(Y) with P(w_p^k) = v[k]; coset point Y_i = shift^(n/p) * w_{p 2^r}^i, i < p 2^r.
static void periodic_on_coset(const uint64_t* v, int period_log, int rate_bits, uint64_t shift_pow, uint64_t* out) {
    size_t p = (size_t)1 << period_log, m = p << rate_bits;
    std::vector<uint64_t> coef(p);
    uint64_t wp_inv = glh::inv(glh::root(period_log)), pinv = glh::inv(p % glh::P);
    for (size_t k = 0; k < p; ++k) {  // inverse DFT, O(p^2)
        uint64_t acc = 0, w = glh::pow(wp_inv, k), cur = 1;
        for (size_t j = 0; j < p; ++j) {
            acc = glh::add(acc, glh::mul(v[j], cur));
            cur = glh::mul(cur, w);
        }
        coef[k] = glh::mul(acc, pinv);
    }
    uint64_t wm = glh::root(period_log + rate_bits);
    for (size_t i = 0; i < m; ++i) {
        uint64_t y = glh::mul(shift_pow, glh::pow(wm, i)), acc = 0;
        for (size_t k = p; k-- > 0;) acc = glh::add(glh::mul(acc, y), coef[k]);
        out[i] = acc;
    }
}

struct DevMem {  // RAII for the prover's temporaries (recycled through the ctx pool)
    vx_ctx* ctx;
    std::vector<void*> ptrs;
    std::vector<vx_tree*> trees;
    explicit DevMem(vx_ctx* c) : ctx(c) {}
    ~DevMem() {
        for (void* p : ptrs) vx_pool_free(ctx, p);
        for (vx_tree* t : trees) {
            vx_pool_free(ctx, t->levels);
            delete t;
        }
    }
    uint64_t* alloc(size_t n_u64) {
        void* p = vx_pool_alloc(ctx, n_u64 * 8);
        if (p) ptrs.push_back(p);
        return (uint64_t*)p;
    }
};

static const uint64_t VX_PROOF_MAGIC = 0x314b524154535856ULL;  // "VXSTARK1"

static void push_cap(vx_ctx* ctx, const vx_tree* t, std::vector<uint64_t>& out) {
    size_t cap = (size_t)4 << t->cap_height;
    size_t o = out.size();
    out.resize(o + cap);
    (void)hipMemcpyAsync(out.data() + o, t->levels + t->total - cap, cap * 8, hipMemcpyDeviceToHost, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
}

static std::vector<int> fri_arity_plan(int degree_bits, const vx_stark_config& cfg) {
    // FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits)
    std::vector<int> r;
    int d = degree_bits;
    while (d > cfg.final_poly_bits && d + cfg.rate_bits - cfg.arity_bits >= cfg.cap_height) {
        r.push_back(cfg.arity_bits);
        d -= cfg.arity_bits;
    }
    return r;
}

struct DevMem;
static int32_t quotient_eval_dev(vx_ctx* ctx, const AirDesc* air, int L, int r, const uint64_t* trace_lde, const uint64_t alphas[2],
                                 const uint64_t* public_inputs, size_t n_public, const uint64_t* chal, const uint64_t* apub, uint64_t* qv, DevMem& mem);

extern "C" {

int32_t vx_stark_default_config(vx_stark_config* cfg) {
    if (!cfg) return VX_ERR_ARG;
    // starky StarkConfig::standard_fast_config: 100-bit conjectured security
    cfg->rate_bits = 1;
    cfg->cap_height = 4;
    cfg->num_queries = 84;
    cfg->pow_bits = 16;
    cfg->arity_bits = 4;
    cfg->final_poly_bits = 5;
    return VX_OK;
}

int32_t vx_stark_proof_bound(int air_id, const vx_stark_config* cfg, int log_n, size_t* n_words) {
    const AirDesc* air = find_air(air_id);
    if (!air || !cfg || !n_words || log_n < 2 || cfg->rate_bits < 1) return VX_ERR_ARG;
    const size_t c = (size_t)air->cols + air->aux, nq = 4, LN = log_n + cfg->rate_bits, cap = (size_t)4 << cfg->cap_height;
    const std::vector<int> ar = fri_arity_plan(log_n, *cfg);
    size_t per_query = c + nq + 2 * 4 * LN, words = 16 + ar.size() + air->pub + 2 * cap + 2 * (2 * c + nq) + ar.size() * cap + 1;
    if (air->aux) per_query += 4 * LN, words += cap + 2 * (size_t)air->auxpub;  // the auxiliary tree: cap, published values, one more path
    size_t cur = LN;
    for (int a : ar) {
        per_query += 2 * (((size_t)1 << a) - 1) + 4 * cur;
        cur -= a;
    }
    words += 2 * (((size_t)1 << cur) >> cfg->rate_bits) + cfg->num_queries * per_query;
    *n_words = words;
    return VX_OK;
}

int32_t vx_quotient_eval(vx_ctx* ctx, int air_id, int rate_bits, const vx_buf* trace_lde, int log_n, const uint64_t alphas[2],
                         const uint64_t* public_inputs, size_t n_public, vx_buf* out) {
    if (!ctx || !trace_lde || !alphas || !out) return VX_ERR_ARG;
    const AirDesc* air = find_air(air_id);
    VX_CHECK(air, "quotient eval: unknown AIR id %d", air_id);
    VX_CHECK(air->aux == 0, "quotient eval: AIR %d has an auxiliary round (its challenges are not part of this entry point)", air_id);
    VX_CHECK(rate_bits >= 1 && rate_bits <= 3 && log_n >= 2 && log_n >= air->period_log && log_n + rate_bits <= 27, "quotient eval: bad log_n %d / rate_bits %d", log_n, rate_bits);
    VX_CHECK((int)n_public == air->pub && (n_public == 0 || public_inputs), "quotient eval: AIR %d takes %d public inputs", air_id, air->pub);
    const size_t N = (size_t)1 << (log_n + rate_bits);
    VX_CHECK(trace_lde->n >= N * (size_t)air->cols && out->n >= 2 * N, "quotient eval: buffers too small");
    VX_CHECK(alphas[0] < glh::P && alphas[1] < glh::P, "quotient eval: non-canonical challenge");
    DevMem mem{ctx};
    return quotient_eval_dev(ctx, air, log_n, rate_bits, trace_lde->d, alphas, public_inputs, n_public, nullptr, nullptr, out->d, mem);
}

int32_t vx_stark_aux_trace(vx_ctx* ctx, int air_id, const vx_buf* trace, int log_n, const uint64_t* public_inputs, size_t n_public,
                           const uint64_t* challenges, size_t n_challenges, vx_buf* aux_out, uint64_t* aux_public_out) {
    if (!ctx || !trace || !challenges || !aux_out) return VX_ERR_ARG;
    const AirDesc* air = find_air(air_id);
    VX_CHECK(air && air->aux > 0 && air_has_gen_aux(air), "aux trace: AIR %d has no auxiliary round (or no generator for it)", air_id);
    VX_CHECK((int)n_challenges == air->chal, "aux trace: AIR %d takes %d challenges", air_id, air->chal);
    VX_CHECK(log_n >= air->period_log && log_n <= 26, "aux trace: log_n %d out of range", log_n);
    const size_t n = (size_t)1 << log_n;
    VX_CHECK(trace->n >= n * (size_t)air->cols && aux_out->n >= n * (size_t)air->aux, "aux trace: buffers too small");
    for (size_t i = 0; i < n_challenges; ++i) VX_CHECK(challenges[i] < glh::P, "aux trace: non-canonical challenge");
    uint64_t apub[8] = {0};
    VX_CHECK((int)n_public == air->pub && (n_public == 0 || public_inputs), "aux trace: AIR %d takes %d public inputs", air_id, air->pub);
    VX_TRY(air_gen_aux(ctx, air, trace->d, log_n, challenges, public_inputs, aux_out->d, apub));
    if (aux_public_out)
        for (int q = 0; q < 2 * air->auxpub; ++q) aux_public_out[q] = apub[q];
    return VX_OK;
}

int32_t vx_stark_prove(vx_ctx* ctx, int air_id, const vx_stark_config* cfg_in, const vx_buf* trace, int log_n,
                       const uint64_t* public_inputs, size_t n_public, uint64_t* proof_out, size_t proof_cap,
                       size_t* proof_len) {
    if (!ctx || !cfg_in || !trace || !proof_len) return VX_ERR_ARG;
    return vx_stark_prove_impl(ctx, air_id, cfg_in, trace->d, trace->n, 0, log_n, public_inputs, n_public, proof_out, proof_cap, proof_len, nullptr);
}
}  // extern "C"

// compute_quotient_polys on the size-N coset g<w_N>: qv[k*N + i] = (sum_j alpha_k^(K-1-j) c_j(x_i)) / Z_H(x_i), k = 0, 1
// The periodic columns of an AIR on the LDE coset (column q: (1 << (plog(q) + r)) values, back to back).  Short periods
// are interpolated on the host; a long one (a 2^16-entry lookup table) goes through the device LDE with the coset
// shift g^(n/p).  The table depends on (AIR, L, r) only, so it is built once per context and shape.
static int32_t periodic_table_dev(vx_ctx* ctx, const AirDesc* air, int L, int r, const uint64_t** out) {
    const uint64_t key = ((uint64_t)air->id << 16) | ((uint64_t)L << 8) | (uint64_t)r;
    auto it = ctx->periodic_cache.find(key);
    if (it != ctx->periodic_cache.end()) {
        *out = it->second;
        return VX_OK;
    }
    const size_t n = (size_t)1 << L;
    std::vector<uint64_t> pv;
    if (air->prog) pv = air->prog->periodic;
    else air->periodic_values(pv);
    size_t total = 0, in_total = 0;
    for (int q = 0; q < air->periodic; ++q) total += (size_t)1 << (air_plog(air, q) + r), in_total += (size_t)1 << air_plog(air, q);
    VX_CHECK(pv.size() == in_total, "periodic table of AIR %d has %zu values, expected %zu", air->id, pv.size(), in_total);
    uint64_t* d = nullptr;
    VX_HIP(hipMalloc((void**)&d, total * 8));
    std::vector<uint64_t> tab(total);
    size_t off = 0, in_off = 0;
    for (int q = 0; q < air->periodic; ++q) {
        const int pl = air_plog(air, q);
        const size_t p = (size_t)1 << pl, m = p << r;
        const uint64_t shift_pow = glh::pow(7, n >> pl);
        if (pl <= 6) {
            periodic_on_coset(pv.data() + in_off, pl, r, shift_pow, tab.data() + off);
            VX_HIP(hipMemcpyAsync(d + off, tab.data() + off, m * 8, hipMemcpyHostToDevice, ctx->stream));
        } else {
            uint64_t* tmp = (uint64_t*)vx_pool_alloc(ctx, p * 8);
            if (!tmp) {
                (void)hipFree(d);
                return vx_fail(ctx, VX_ERR_OOM, "stark prove: out of device memory (periodic table)");
            }
            VX_HIP(hipMemcpyAsync(tmp, pv.data() + in_off, p * 8, hipMemcpyHostToDevice, ctx->stream));
            int32_t rc = vx_lde_dev(ctx, tmp, pl, 1, r, shift_pow, VX_LDE_SRC_VALUES, d + off, nullptr);
            VX_HIP(hipStreamSynchronize(ctx->stream));
            vx_pool_free(ctx, tmp);
            if (rc != VX_OK) {
                (void)hipFree(d);
                return rc;
            }
        }
        off += m, in_off += p;
    }
    VX_HIP(hipStreamSynchronize(ctx->stream));  // tab / pv (host vectors) must outlive the copies
    ctx->periodic_cache[key] = d;
    *out = d;
    return VX_OK;
}

static int32_t quotient_eval_dev(vx_ctx* ctx, const AirDesc* air, int L, int r, const uint64_t* trace_lde, const uint64_t alphas[2],
                                 const uint64_t* public_inputs, size_t n_public, const uint64_t* chal, const uint64_t* apub, uint64_t* qv, DevMem& mem) {
    const int LN = L + r;
    const size_t n = (size_t)1 << L;
    const uint64_t g = 7;
    uint64_t* d_pub = mem.alloc(n_public ? n_public : 1);
    VX_CHECK(d_pub, "stark prove: out of device memory (quotient)");
    if (n_public) VX_HIP(hipMemcpyAsync(d_pub, public_inputs, n_public * 8, hipMemcpyHostToDevice, ctx->stream));
    const uint64_t* d_per = nullptr;
    if (air->periodic) VX_TRY(periodic_table_dev(ctx, air, L, r, &d_per));
    {
        QuotArgs qa{};
        qa.lde = trace_lde;
        qa.q_out = qv;
        qa.log_N = LN;
        qa.rate_bits = r;
        qa.shift = g;
        qa.last = glh::inv(glh::root(L));
        qa.n_inv = glh::inv(n % glh::P);
        qa.alpha[0] = alphas[0];
        qa.alpha[1] = alphas[1];
        const uint64_t gn = glh::pow(g, n), wr = glh::root(r);
        for (int k = 0; k < (1 << r); ++k) {
            qa.zh[k] = glh::sub(glh::mul(gn, glh::pow(wr, k)), 1);
            qa.zh_inv[k] = glh::inv(qa.zh[k]);
        }
        qa.periodic = d_per;
        qa.pub = d_pub;
        for (int q = 0; q < air->chal && q < 8; ++q) qa.chal[q] = chal[q];
        for (int q = 0; q < 2 * air->auxpub && q < 8; ++q) qa.apub[q] = apub[q];
        qa.tw = ctx->tw_fwd.d;
        // powers of the two alphas for the K constraints (the Horner recurrence as a dot product, air.cuh)
        const int K = air->prog ? (int)air->prog->n_constraints : air->count();
        std::vector<uint64_t> apow(2 * (size_t)K);
        uint64_t pw[2] = {1, 1};
        for (int kk = K - 1; kk >= 0; --kk)
            for (int j = 0; j < 2; ++j) {
                apow[2 * (size_t)kk + j] = pw[j];
                pw[j] = glh::mul(pw[j], alphas[j]);
            }
        uint64_t* d_apow_q = mem.alloc(apow.size());
        VX_CHECK(d_apow_q, "stark prove: out of device memory (alpha powers)");
        VX_HIP(hipMemcpyAsync(d_apow_q, apow.data(), apow.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        qa.apow = d_apow_q;
        std::vector<uint64_t> per_tab;
        if (air->prog) {
            // the program on the device (cached per context: programs are immutable), and where each periodic column sits in qa.periodic
            const AirProgram& pg = *air->prog;
            const uint64_t key = ((uint64_t)air->id << 16) | 0xFFFFULL;  // (periodic tables use L << 8 | r <= 0x1B03 in the low half)
            uint64_t* d_code = nullptr;
            auto it = ctx->periodic_cache.find(key);
            if (it != ctx->periodic_cache.end()) d_code = it->second;
            else {
                std::vector<uint64_t> blob(pg.code);
                blob.insert(blob.end(), pg.consts.begin(), pg.consts.end());
                VX_HIP(hipMalloc((void**)&d_code, blob.size() * 8));
                VX_HIP(hipMemcpy(d_code, blob.data(), blob.size() * 8, hipMemcpyHostToDevice));
                ctx->periodic_cache[key] = d_code;
            }
            size_t off = 0;
            for (size_t q = 0; q < pg.plog.size(); ++q) {
                const size_t plen = (size_t)1 << (pg.plog[q] + r);
                per_tab.push_back(off), per_tab.push_back(plen - 1);
                off += plen;
            }
            uint64_t* d_per_tab = mem.alloc(per_tab.size() ? per_tab.size() : 1);
            VX_CHECK(d_per_tab, "stark prove: out of device memory (program tables)");
            if (!per_tab.empty()) VX_HIP(hipMemcpyAsync(d_per_tab, per_tab.data(), per_tab.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            ProgArgs pa{d_code, d_code + pg.code.size(), d_per_tab, (int)pg.code.size()};
            const size_t N = (size_t)1 << LN;
            hipLaunchKernelGGL(k_quotient_prog, dim3((unsigned)((N + QB - 1) / QB)), dim3(QB), pg.n_regs * QB * sizeof(uint64_t), ctx->stream, qa, pa);
        } else air->launch(qa, ctx->stream);
        VX_HIP(hipStreamSynchronize(ctx->stream));  // apow / per_tab (host vectors) must outlive the copies
        VX_HIP(hipGetLastError());
    }
    return VX_OK;
}

// Lookup challenges shared by two tables: a transcript of both tables' public inputs and trace caps.
void vx_shared_challenges_n(const uint64_t* const* pubs, const size_t* n_pubs, const uint64_t* const* caps, size_t k, size_t cap_words, uint64_t* out, size_t n_out) {
    Challenger sc;
    for (size_t t = 0; t < k; ++t) {
        sc.observe(pubs[t], n_pubs[t]);
        sc.observe(caps[t], cap_words);
    }
    for (size_t q = 0; q < n_out; ++q) out[q] = sc.challenge();
}
void vx_shared_challenges(const uint64_t* pub_a, size_t n_a, const uint64_t* cap_a, const uint64_t* pub_b, size_t n_b, const uint64_t* cap_b,
                          size_t cap_words, uint64_t* out, size_t n_out) {
    const uint64_t *pubs[2] = {pub_a, pub_b}, *caps[2] = {cap_a, cap_b};
    const size_t ns[2] = {n_a, n_b};
    vx_shared_challenges_n(pubs, ns, caps, 2, cap_words, out, n_out);
}

// consume_trace != 0: the trace buffer is overwritten (it ends up holding the bit-reversed coefficients);
// saves the n*c coefficient scratch -- the caller must own the buffer.
int32_t vx_stark_prove_impl(vx_ctx* ctx, int air_id, const vx_stark_config* cfg_in, uint64_t* trace_d, size_t trace_len, int consume_trace,
                            int log_n, const uint64_t* public_inputs, size_t n_public, uint64_t* proof_out, size_t proof_cap,
                            size_t* proof_len, const vx_chal_hook* hook) {
    if (!ctx || !cfg_in || !trace_d || !proof_len) return VX_ERR_ARG;
    const AirDesc* air = find_air(air_id);
    VX_CHECK(air, "stark prove: unknown AIR id %d", air_id);
    const vx_stark_config cfg = *cfg_in;
    VX_CHECK(cfg.rate_bits >= 1 && cfg.rate_bits <= 3, "stark prove: rate_bits %d not in [1,3]", cfg.rate_bits);
    VX_CHECK(cfg.arity_bits >= 1 && cfg.arity_bits <= 5 && cfg.final_poly_bits >= 0 && cfg.num_queries >= 1 && cfg.num_queries <= 1024 &&
                 cfg.pow_bits >= 0 && cfg.pow_bits <= 32, "stark prove: bad FRI config");
    const int L = log_n, r = cfg.rate_bits, LN = L + r;
    VX_CHECK(L >= air->period_log && L >= 2 && LN <= 27, "stark prove: log_n %d out of range", L);
    VX_CHECK(!air->exact_log || L == air->period_log, "stark prove: AIR %d has positional columns of period 2^%d, the trace must have exactly that many rows", air_id, air->period_log);
    VX_CHECK(cfg.cap_height >= 0 && cfg.cap_height <= LN, "stark prove: cap_height %d > log2(lde size) %d", cfg.cap_height, LN);
    VX_CHECK((int)n_public == air->pub && (n_public == 0 || public_inputs), "stark prove: AIR %d takes %d public inputs", air_id, air->pub);
    // c = every committed trace column (main ++ auxiliary); the first cm come from the caller, the other ca are derived
    // after the lookup challenges are known
    const size_t n = (size_t)1 << L, N = (size_t)1 << LN, cm = air->cols, ca = air->aux, c = cm + ca;
    VX_CHECK(trace_len >= n * cm, "stark prove: trace holds %zu < %zu elements", trace_len, n * cm);
    VX_CHECK(air->chal <= 8 && 2 * air->auxpub <= 8 && (ca == 0 || air_has_gen_aux(air)), "stark prove: AIR %d auxiliary round is misconfigured (a program needs its gen_aux callback to be proven)", air_id);
    for (size_t i = 0; i < n_public; ++i) VX_CHECK(public_inputs[i] < glh::P, "stark prove: public input %zu not canonical", i);
    const int Q = 2, nq = 2 * Q;  // quotient_degree_factor 2 (constraint degree 3), 2 challenges
    const uint64_t g = 7;         // F::coset_shift()
    DevMem mem(ctx);
    std::vector<uint64_t> proof;

    // ---- 1. trace commitment: PolynomialBatch::from_values
    uint64_t* trace_lde = mem.alloc(N * c);
    VX_CHECK(trace_lde, "stark prove: out of device memory (trace LDE)");
    // coef_main: the main columns' coefficients in bit-reversed positions (openings at zeta as plain dot products)
    uint64_t* coef_main = nullptr;
    if (ca) {  // the trace values are needed again for the auxiliary columns: the inverse transform writes elsewhere
        coef_main = mem.alloc(n * cm);
        VX_CHECK(coef_main, "stark prove: out of device memory (coefficients)");
        VX_TRY(vx_lde_keep_dev(ctx, trace_d, L, cm, r, g, coef_main, trace_lde));
    } else if (consume_trace) {
        VX_TRY(vx_lde_consume_dev(ctx, trace_d, L, cm, r, g, trace_lde));
        coef_main = trace_d;
    } else VX_TRY(vx_lde_dev(ctx, trace_d, L, cm, r, g, VX_LDE_SRC_VALUES, trace_lde, nullptr));
    vx_tree* t_trace = nullptr;
    VX_TRY(vx_merkle_build_dev(ctx, trace_lde, N, cm, VX_LEAVES_COLS_BITREV, cfg.cap_height, &t_trace));
    mem.trees.push_back(t_trace);

    const std::vector<int> arities = fri_arity_plan(L, cfg);
    int final_log = LN;
    for (int a : arities) final_log -= a;
    const size_t final_len = ((size_t)1 << final_log) >> r;

    proof.push_back(VX_PROOF_MAGIC);
    for (uint64_t w : {(uint64_t)air_id, (uint64_t)L, (uint64_t)cm, (uint64_t)nq, (uint64_t)r, (uint64_t)cfg.cap_height,
                       (uint64_t)cfg.num_queries, (uint64_t)cfg.pow_bits, (uint64_t)arities.size()})
        proof.push_back(w);
    for (int a : arities) proof.push_back((uint64_t)a);
    proof.push_back((uint64_t)final_len);
    proof.push_back((uint64_t)n_public);
    for (size_t i = 0; i < n_public; ++i) proof.push_back(public_inputs[i]);
    const size_t cap_words = (size_t)4 << cfg.cap_height;
    push_cap(ctx, t_trace, proof);

    Challenger ch;
    ch.observe(public_inputs, n_public);
    ch.observe(proof.data() + proof.size() - cap_words, cap_words);
    // ---- 1b. auxiliary round (lookup arguments): challenges after the trace cap, derived columns in a second tree
    uint64_t chal[8] = {0}, apub[8] = {0};
    uint64_t* aux_d = nullptr;  // [ca][n]: values, then (after the in-place inverse NTT) coefficients in bit-reversed positions
    vx_tree* t_aux = nullptr;
    if (ca) {
        if (hook) {
            // lookup challenges SHARED with other tables (a bus between AIRs): the caller derives them once every
            // table's trace cap exists; this transcript absorbs them so that everything after depends on them
            const int32_t hr = hook->fn(hook->user, public_inputs, n_public, proof.data() + proof.size() - cap_words, cap_words, chal, (size_t)air->chal);
            if (hr != VX_OK) return vx_fail(ctx, hr, "stark prove: no shared challenges (the prover of the other table on the bus gave up)");
            for (int q = 0; q < air->chal; ++q) VX_CHECK(chal[q] < glh::P, "stark prove: shared challenge %d is not canonical", q);
            ch.observe(chal, (size_t)air->chal);
        } else
            for (int q = 0; q < air->chal; ++q) chal[q] = ch.challenge();
        aux_d = mem.alloc(n * ca);
        VX_CHECK(aux_d, "stark prove: out of device memory (auxiliary trace)");
        VX_TRY(air_gen_aux(ctx, air, trace_d, L, chal, public_inputs, aux_d, apub));
        VX_TRY(vx_lde_consume_dev(ctx, aux_d, L, ca, r, g, trace_lde + N * cm));
        VX_TRY(vx_merkle_build_dev(ctx, trace_lde + N * cm, N, ca, VX_LEAVES_COLS_BITREV, cfg.cap_height, &t_aux));
        mem.trees.push_back(t_aux);
        for (int q = 0; q < 2 * air->auxpub; ++q) proof.push_back(apub[q]);
        push_cap(ctx, t_aux, proof);
        ch.observe(apub, 2 * (size_t)air->auxpub);
        ch.observe(proof.data() + proof.size() - cap_words, cap_words);
    }
    uint64_t alphas[2] = {ch.challenge(), 0};
    alphas[1] = ch.challenge();

    // ---- 2. quotient polynomials (compute_quotient_polys) on the size-N coset
    uint64_t* qv = mem.alloc(2 * N);
    VX_CHECK(qv, "stark prove: out of device memory (quotient)");
    VX_TRY(quotient_eval_dev(ctx, air, L, r, trace_lde, alphas, public_inputs, n_public, chal, apub, qv, mem));
    // values on the coset -> coefficients (coset_ifft), split into Q chunks of n, commit (from_coeffs)
    VX_TRY(vx_ntt_dev(ctx, qv, LN, 2, N, 1, g, VX_ORDER_NATURAL));
    // chunk j of challenge k = qv[k*N + j*n .. +n): already contiguous as 2*Q columns of n coefficients
    uint64_t* quot_lde = mem.alloc(N * nq);
    VX_CHECK(quot_lde, "stark prove: out of device memory (quotient LDE)");
    VX_TRY(vx_lde_dev(ctx, qv, L, nq, r, g, VX_LDE_SRC_COEFFS, quot_lde, nullptr));
    vx_tree* t_quot = nullptr;
    VX_TRY(vx_merkle_build_dev(ctx, quot_lde, N, nq, VX_LEAVES_COLS_BITREV, cfg.cap_height, &t_quot));
    mem.trees.push_back(t_quot);
    push_cap(ctx, t_quot, proof);
    ch.observe(proof.data() + proof.size() - cap_words, cap_words);
    const Ext zeta = ch.ext_challenge();
    const uint64_t wn = glh::root(L);
    const Ext zeta_next = e_scale(zeta, wn);
    {
        Ext zn = e_pow(zeta, n);
        VX_CHECK(!(zn.a == 1 && zn.b == 0), "stark prove: zeta landed in the trace subgroup");
    }

    // ---- 3. openings (StarkOpeningSet::new): barycentric dot products over the committed LDE values at the
    // coset points g * w_n^i (every 2^r-th LDE point) -- neither the trace values nor coefficients are needed
    uint64_t* w0 = mem.alloc(2 * n);
    uint64_t* w1 = mem.alloc(2 * n);
    const size_t open_words = 4 * (c + nq);
    uint64_t* d_open = mem.alloc(BARY_SPLIT * open_words);
    VX_CHECK(w0 && w1 && d_open, "stark prove: out of device memory (openings)");
    hipLaunchKernelGGL(k_bary_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, L, g, gl2{zeta.a, zeta.b},
                       gl2{zeta_next.a, zeta_next.b}, (const uint64_t*)ctx->tw_fwd.d, w0, w1);
    hipLaunchKernelGGL(k_bary_dot, dim3((unsigned)((nq + BARY_CPB - 1) / BARY_CPB), BARY_SPLIT), dim3(256), 0, ctx->stream, (const uint64_t*)quot_lde, N, r, n, (size_t)nq, (const uint64_t*)w0,
                       (const uint64_t*)w1, d_open + 4 * c, open_words);
    if (coef_main) {
        // the coefficients are at hand (bit-reversed positions): plain evaluation, no barycentric factor
        PowBrevArgs pa{};
        pa.log_n = L;
        Ext z0 = zeta, z1 = zeta_next;
        for (int b = 0; b < L; ++b) {
            pa.z0[b] = {z0.a, z0.b};
            pa.z1[b] = {z1.a, z1.b};
            z0 = e_mul(z0, z0);
            z1 = e_mul(z1, z1);
        }
        hipLaunchKernelGGL(k_pow_brev_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, pa, w0, w1);
        hipLaunchKernelGGL(k_bary_dot, dim3((unsigned)((cm + BARY_CPB - 1) / BARY_CPB), BARY_SPLIT), dim3(256), 0, ctx->stream, (const uint64_t*)coef_main, n, 0, n, cm, (const uint64_t*)w0,
                           (const uint64_t*)w1, d_open, open_words);
        if (ca)
            hipLaunchKernelGGL(k_bary_dot, dim3((unsigned)((ca + BARY_CPB - 1) / BARY_CPB), BARY_SPLIT), dim3(256), 0, ctx->stream, (const uint64_t*)aux_d, n, 0, n, ca, (const uint64_t*)w0,
                               (const uint64_t*)w1, d_open + 4 * cm, open_words);
    } else {
        hipLaunchKernelGGL(k_bary_dot, dim3((unsigned)((c + BARY_CPB - 1) / BARY_CPB), BARY_SPLIT), dim3(256), 0, ctx->stream, (const uint64_t*)trace_lde, N, r, n, c, (const uint64_t*)w0,
                           (const uint64_t*)w1, d_open, open_words);
    }
    VX_HIP(hipGetLastError());
    std::vector<uint64_t> h_open(BARY_SPLIT * open_words);
    VX_HIP(hipMemcpyAsync(h_open.data(), d_open, h_open.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    for (int sp = 1; sp < BARY_SPLIT; ++sp)  // the partial sums of the split blocks
        for (size_t j = 0; j < open_words; ++j) h_open[j] = glh::add(h_open[j], h_open[sp * open_words + j]);
    const uint64_t gn_ = glh::pow(g, n), fsc = glh::inv(glh::mul(n % glh::P, gn_));
    const Ext f0 = e_scale(e_sub(e_pow(zeta, n), Ext{gn_, 0}), fsc);       // (zeta^n - g^n) / (n g^n)
    const Ext f1 = e_scale(e_sub(e_pow(zeta_next, n), Ext{gn_, 0}), fsc);  // ((w zeta)^n - g^n) / (n g^n)
    std::vector<Ext> o_local(c), o_next(c), o_quot(nq);
    const Ext one_{1, 0};
    const Ext t0 = coef_main ? one_ : f0, t1 = coef_main ? one_ : f1;  // coefficient sums need no factor
    for (size_t j = 0; j < c; ++j) {
        o_local[j] = e_mul(t0, Ext{h_open[4 * j], h_open[4 * j + 1]});
        o_next[j] = e_mul(t1, Ext{h_open[4 * j + 2], h_open[4 * j + 3]});
    }
    for (int j = 0; j < nq; ++j) o_quot[j] = e_mul(f0, Ext{h_open[4 * (c + j)], h_open[4 * (c + j) + 1]});
    for (const Ext& e : o_local) proof.push_back(e.a), proof.push_back(e.b);
    for (const Ext& e : o_next) proof.push_back(e.a), proof.push_back(e.b);
    for (const Ext& e : o_quot) proof.push_back(e.a), proof.push_back(e.b);
    // challenger.observe_openings: batch 0 = local ++ quotient, batch 1 = next
    for (const Ext& e : o_local) ch.observe(e.a), ch.observe(e.b);
    for (const Ext& e : o_quot) ch.observe(e.a), ch.observe(e.b);
    for (const Ext& e : o_next) ch.observe(e.a), ch.observe(e.b);

    // ---- 4. FRI batch polynomial (prove_openings), in evaluation space
    const Ext alpha = ch.ext_challenge();
    std::vector<uint64_t> apow(2 * (c + nq));
    Ext cur{1, 0}, y0{0, 0}, y1{0, 0}, alpha_c{1, 0};
    for (size_t j = 0; j < c + nq; ++j) {
        apow[2 * j] = cur.a;
        apow[2 * j + 1] = cur.b;
        if (j < c) {
            y0 = e_add(y0, e_mul(cur, o_local[j]));
            y1 = e_add(y1, e_mul(cur, o_next[j]));
        } else y0 = e_add(y0, e_mul(cur, o_quot[j - c]));
        cur = e_mul(cur, alpha);
        if (j + 1 == c) alpha_c = cur;
    }
    uint64_t* d_apow = mem.alloc(apow.size());
    std::vector<uint64_t*> layers;
    layers.push_back(mem.alloc(2 * N));
    VX_CHECK(d_apow && layers[0], "stark prove: out of device memory (FRI)");
    VX_HIP(hipMemcpyAsync(d_apow, apow.data(), apow.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    {
        CombineArgs ca{};
        ca.trace_lde = trace_lde;
        ca.quot_lde = quot_lde;
        ca.n_cols = (int)c;
        ca.n_q = nq;
        ca.log_N = LN;
        ca.alpha_pow = d_apow;
        ca.alpha_c = {alpha_c.a, alpha_c.b};
        ca.y0 = {y0.a, y0.b};
        ca.y1 = {y1.a, y1.b};
        ca.zeta = {zeta.a, zeta.b};
        ca.zeta_next = {zeta_next.a, zeta_next.b};
        ca.shift = g;
        ca.tw = ctx->tw_fwd.d;
        ca.out = layers[0];
        if (N >= 1024) hipLaunchKernelGGL(k_fri_combine<4>, dim3((unsigned)(N / 1024)), dim3(256), 0, ctx->stream, ca);
        else hipLaunchKernelGGL(k_fri_combine<1>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, ca);
        VX_HIP(hipGetLastError());
    }

    // ---- 5. FRI commit phase (fri_committed_trees)
    std::vector<vx_tree*> ltrees;
    std::vector<int> llog;
    int cur_log = LN;
    uint64_t shift = g;
    for (int a : arities) {
        vx_tree* t = nullptr;
        VX_TRY(vx_fri_layer_tree_dev(ctx, layers.back(), cur_log, a, cfg.cap_height, &t));
        mem.trees.push_back(t);
        ltrees.push_back(t);
        llog.push_back(cur_log);
        push_cap(ctx, t, proof);
        ch.observe(proof.data() + proof.size() - cap_words, cap_words);
        const Ext beta = ch.ext_challenge();
        const uint64_t b2[2] = {beta.a, beta.b};
        uint64_t* nxt = mem.alloc((size_t)2 << (cur_log - a));
        VX_CHECK(nxt, "stark prove: out of device memory (FRI layer)");
        VX_TRY(vx_fri_fold_dev(ctx, layers.back(), cur_log, a, b2, shift, nxt));
        layers.push_back(nxt);
        shift = glh::pow(shift, (uint64_t)1 << a);
        cur_log -= a;
    }
    // final polynomial: coset_ifft of the last layer on the host (<= 2^(final_poly_bits + arity + r) points)
    const size_t fm = (size_t)1 << cur_log;
    std::vector<uint64_t> fv(2 * fm), fc(2 * fm);
    VX_HIP(hipMemcpyAsync(fv.data(), layers.back(), 2 * fm * 8, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    {
        const uint64_t wi = glh::inv(glh::root(cur_log)), minv = glh::inv(fm % glh::P), sinv = glh::inv(shift);
        uint64_t sk = 1;
        for (size_t k = 0; k < fm; ++k) {
            Ext acc{0, 0};
            const uint64_t wk = glh::pow(wi, k);
            uint64_t wcur = 1;
            for (size_t i = 0; i < fm; ++i) {
                acc = e_add(acc, e_scale(Ext{fv[2 * i], fv[2 * i + 1]}, wcur));
                wcur = glh::mul(wcur, wk);
            }
            acc = e_scale(acc, glh::mul(minv, sk));
            fc[2 * k] = acc.a;
            fc[2 * k + 1] = acc.b;
            sk = glh::mul(sk, sinv);
        }
    }
    for (size_t k = final_len; k < fm; ++k)
        VX_CHECK(fc[2 * k] == 0 && fc[2 * k + 1] == 0, "stark prove: final polynomial has degree >= %zu: the trace violates the AIR constraints", final_len);
    for (size_t k = 0; k < 2 * final_len; ++k) proof.push_back(fc[k]);
    ch.observe(fc.data(), 2 * final_len);

    // ---- 6. proof of work (fri_proof_of_work), smallest nonce
    {
        uint64_t st[12];
        memcpy(st, ch.st, sizeof st);
        for (int i = 0; i < ch.n_in; ++i) st[i] = ch.in[i];
        uint64_t nonce = 0;
        VX_TRY(vx_fri_pow(ctx, st, ch.n_in, cfg.pow_bits, &nonce));
        proof.push_back(nonce);
        ch.observe(nonce);
        const uint64_t resp = ch.challenge();
        VX_CHECK(cfg.pow_bits == 0 || (resp >> (64 - cfg.pow_bits)) == 0, "stark prove: PoW response check failed");
    }

    // ---- 7. query phase (fri_prover_query_rounds)
    const size_t nqr = cfg.num_queries;
    std::vector<uint64_t> qidx(nqr);
    for (size_t k = 0; k < nqr; ++k) qidx[k] = ch.challenge() % N;
    const int depth0 = LN - cfg.cap_height;
    std::vector<uint64_t> rows_t(nqr * cm), rows_a(nqr * ca), rows_q(nqr * nq), sib_t(nqr * depth0 * 4), sib_a(ca ? nqr * depth0 * 4 : 0), sib_q(nqr * depth0 * 4);
    VX_TRY(vx_gather_rows_dev(ctx, trace_lde, LN, cm, qidx.data(), nqr, rows_t.data()));
    if (ca) VX_TRY(vx_gather_rows_dev(ctx, trace_lde + N * cm, LN, ca, qidx.data(), nqr, rows_a.data()));
    VX_TRY(vx_gather_rows_dev(ctx, quot_lde, LN, nq, qidx.data(), nqr, rows_q.data()));
    if (depth0 > 0) {
        VX_TRY(vx_merkle_open(ctx, t_trace, qidx.data(), nqr, sib_t.data()));
        if (ca) VX_TRY(vx_merkle_open(ctx, t_aux, qidx.data(), nqr, sib_a.data()));
        VX_TRY(vx_merkle_open(ctx, t_quot, qidx.data(), nqr, sib_q.data()));
    }
    std::vector<std::vector<uint64_t>> l_leaves(arities.size()), l_sibs(arities.size());
    std::vector<uint64_t> lidx = qidx;
    for (size_t l = 0; l < arities.size(); ++l) {
        const int a = arities[l], depth = llog[l] - a - cfg.cap_height;
        for (size_t k = 0; k < nqr; ++k) lidx[k] >>= a;  // leaf index in this layer's tree
        l_leaves[l].resize(nqr * ((size_t)2 << a));
        VX_TRY(vx_fri_leaves_dev(ctx, layers[l], llog[l], a, lidx.data(), nqr, l_leaves[l].data()));
        l_sibs[l].resize(nqr * (depth > 0 ? depth : 0) * 4);
        if (depth > 0) VX_TRY(vx_merkle_open(ctx, ltrees[l], lidx.data(), nqr, l_sibs[l].data()));
    }
    for (size_t k = 0; k < nqr; ++k) {
        proof.insert(proof.end(), rows_t.begin() + k * cm, rows_t.begin() + (k + 1) * cm);
        proof.insert(proof.end(), sib_t.begin() + k * depth0 * 4, sib_t.begin() + (k + 1) * depth0 * 4);
        if (ca) {
            proof.insert(proof.end(), rows_a.begin() + k * ca, rows_a.begin() + (k + 1) * ca);
            proof.insert(proof.end(), sib_a.begin() + k * depth0 * 4, sib_a.begin() + (k + 1) * depth0 * 4);
        }
        proof.insert(proof.end(), rows_q.begin() + k * nq, rows_q.begin() + (k + 1) * nq);
        proof.insert(proof.end(), sib_q.begin() + k * depth0 * 4, sib_q.begin() + (k + 1) * depth0 * 4);
        uint64_t x_index = qidx[k];
        for (size_t l = 0; l < arities.size(); ++l) {
            const int a = arities[l], depth = llog[l] - a - cfg.cap_height;
            const size_t arity = (size_t)1 << a, within = x_index & (arity - 1);
            const uint64_t* leaf = l_leaves[l].data() + k * 2 * arity;
            for (size_t t = 0; t < arity; ++t)  // evals.remove(x_index & (arity - 1))
                if (t != within) proof.push_back(leaf[2 * t]), proof.push_back(leaf[2 * t + 1]);
            if (depth > 0) proof.insert(proof.end(), l_sibs[l].begin() + k * depth * 4, l_sibs[l].begin() + (k + 1) * depth * 4);
            x_index >>= a;
        }
    }
    *proof_len = proof.size();
    if (!proof_out || proof_cap < proof.size()) return vx_fail(ctx, VX_ERR_BUFSZ, "stark prove: proof needs %zu words, buffer has %zu", proof.size(), proof_cap);
    memcpy(proof_out, proof.data(), proof.size() * 8);
    return VX_OK;
}
