// AIR interface shared by the GPU quotient kernel (base field, one LDE point per lane) and
// the host verifier (extension field, at zeta).  An AIR is a struct with
//   static constexpr int COLS, PUB, PERIODIC, PERIOD_LOG;      // PERIOD_LOG = the largest period
//   static constexpr int AUX, CHAL, AUXPUB;                     // auxiliary round (0 0 0 = none), see below
//   static constexpr int EXACT_LOG;                             // 1: positional columns have the period of the trace, so log2(rows) must EQUAL PERIOD_LOG
//   static constexpr int plog(int q);                           // period (log2) of periodic column q
//   template <class F, class Row, class C> static void eval(const Row& loc, const Row& nxt, const F* per,
//                                                           const F* pub, const F* chal, const F* apub, C& c);
// `eval` pushes constraints into the consumer IN A FIXED ORDER (the order is part of the
// protocol: the oracle restates it independently in oracle/stark_ref.py).
//
// Auxiliary round (lookup arguments, logUp): after the trace cap the transcript yields CHAL base-field challenges;
// the prover derives AUX further columns from the trace and the challenges (helper sums 1/(beta + tuple), running
// sums), commits them in a second Merkle tree, and only then the constraint challenges alpha are drawn.  Rows seen
// by eval hold the COLS main columns followed by the AUX auxiliary ones; `apub` are 2*AUXPUB values the prover
// publishes with the auxiliary cap (bus totals).  Lookup arithmetic runs in the quadratic extension (X2 below).
//
// Consumer semantics follow starky v0.2.0 ConstraintConsumer (crate `starky`, same git rev as
// plonky2 in /root/reference Cargo.lock:4848): acc = acc * alpha + c for each of the
// num_challenges alphas; transition constraints are multiplied by z_last = x - w^-1,
// first/last-row constraints by the Lagrange basis polynomial of that row.  Constraints
// pushed with `constraint()` must hold on EVERY row including the wrap-around pair
// (last, first): the total degree bound is 3 (quotient_degree_factor 2, rate_bits 1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gl.cuh"

#define VX_HD __host__ __device__ __forceinline__

// Device base-field element, R LDE points per lane (rows i, i + 256, ... of the block's tile).  R > 1 gives every
// load R independent streams per lane (the kernel is bound by load latency -- each column lives in its own pages --
// not by bytes) and makes a block touch R x 2 KB of a column per visit.
template <int R>
struct FpN {
    static constexpr int LANES = R;
    uint64_t v[R];
    __device__ __forceinline__ FpN operator+(const FpN& o) const {
        FpN r;
#pragma unroll
        for (int j = 0; j < R; ++j) r.v[j] = gl_add(v[j], o.v[j]);
        return r;
    }
    __device__ __forceinline__ FpN operator-(const FpN& o) const {
        FpN r;
#pragma unroll
        for (int j = 0; j < R; ++j) r.v[j] = gl_sub(v[j], o.v[j]);
        return r;
    }
    __device__ __forceinline__ FpN operator*(const FpN& o) const {
        FpN r;
#pragma unroll
        for (int j = 0; j < R; ++j) r.v[j] = gl_mul(v[j], o.v[j]);
        return r;
    }
    __device__ __forceinline__ static FpN from(uint64_t x) {
        FpN r;
#pragma unroll
        for (int j = 0; j < R; ++j) r.v[j] = x;
        return r;
    }
};
using Fp = FpN<1>;
template <class F>
struct is_device_field : std::false_type {};
template <int R>
struct is_device_field<FpN<R>> : std::true_type {};

// host-side extension-field element (verifier: constraints evaluated at zeta)
struct Fx {
    uint64_t a, b;
    static constexpr uint64_t P_ = 0xFFFFFFFF00000001ULL;
    static inline uint64_t addm(uint64_t x, uint64_t y) {
        uint64_t s = x + y;
        return (s < x || s >= P_) ? s - P_ : s;
    }
    static inline uint64_t subm(uint64_t x, uint64_t y) { return x >= y ? x - y : x + (P_ - y); }
    static inline uint64_t mulm(uint64_t x, uint64_t y) { return (uint64_t)(((unsigned __int128)x * y) % P_); }
    Fx operator+(Fx o) const { return {addm(a, o.a), addm(b, o.b)}; }
    Fx operator-(Fx o) const { return {subm(a, o.a), subm(b, o.b)}; }
    Fx operator*(Fx o) const {
        return {addm(mulm(a, o.a), mulm(7, mulm(b, o.b))), addm(mulm(a, o.b), mulm(b, o.a))};
    }
    static Fx from(uint64_t x) { return {x % P_, 0}; }
};
struct HostRow {
    const Fx* v;
    Fx operator[](int col) const { return v[col]; }
};

// Quadratic-extension element F[X]/(X^2 - 7) over any field type of this file (device FpN, host Fx, CountF): the
// challenges beta / gamma and the helper and running-sum columns of a lookup argument are extension elements kept as
// two base columns; a constraint on X2 values is two base constraints (consumer.constraint_x2).
template <class F>
struct X2 {
    F a, b;
};
template <class F>
VX_HD F f_mul7(const F& x) {  // 7x = 8x - x: three doublings and a subtraction instead of a modular multiply
    const F x2 = x + x, x4 = x2 + x2;
    return x4 + x4 - x;
}
template <class F>
VX_HD X2<F> operator+(const X2<F>& x, const X2<F>& y) { return {x.a + y.a, x.b + y.b}; }
template <class F>
VX_HD X2<F> operator-(const X2<F>& x, const X2<F>& y) { return {x.a - y.a, x.b - y.b}; }
template <class F>
VX_HD X2<F> operator+(const X2<F>& x, const F& y) { return {x.a + y, x.b}; }
template <class F>
VX_HD X2<F> operator-(const X2<F>& x, const F& y) { return {x.a - y, x.b}; }
template <class F>
VX_HD X2<F> operator*(const X2<F>& x, const X2<F>& y) { return {x.a * y.a + f_mul7(x.b * y.b), x.a * y.b + x.b * y.a}; }
template <class F>
VX_HD X2<F> operator*(const X2<F>& x, const F& y) { return {x.a * y, x.b * y}; }

// Gates: a run of constraints that share one factor g (a row selector) may be pushed as
//   auto G = c.open(g); ... c.gated(G, e) ... ; c.close(G);
// which stands for c.constraint(g * e) at each position (several gates may be open at once, interleaved with plain
// constraints).  The generic consumer does exactly that; the device consumer factors g out of the run.
template <class F>
struct Consumer {
    struct Gate {
        F g;
    };
    F acc[2], alpha[2], z_last, l_first, l_last;
    VX_HD void constraint(F c) {
        acc[0] = acc[0] * alpha[0] + c;
        acc[1] = acc[1] * alpha[1] + c;
    }
    VX_HD void transition(F c) { constraint(c * z_last); }
    VX_HD void first_row(F c) { constraint(c * l_first); }
    VX_HD void last_row(F c) { constraint(c * l_last); }
    VX_HD void constraint_x2(const X2<F>& e) {
        constraint(e.a);
        constraint(e.b);
    }
    VX_HD Gate open(F g) { return Gate{g}; }
    VX_HD void gated(Gate& G, F e) { constraint(G.g * e); }
    VX_HD void close(Gate&) {}
};

// Device consumer: the Horner recurrence acc = acc alpha + c_k over K constraints equals sum_k c_k alpha^(K-1-k),
// so with the powers tabulated (apow[2k + j] = alpha_j^(K-1-k), uniform loads) each constraint costs two
// multiply-accumulates into 160-bit integers (gl_mac) and nothing is reduced until the end.  A gate has its own
// pair of accumulators: sum over the run of alpha^(..) g e_k = g * sum alpha^(..) e_k, one extra product per run.
// The result is the same field element the generic consumer computes -- proofs stay byte-identical.
template <int R>
struct Consumer<FpN<R>> {
    using F = FpN<R>;
    struct Gate {
        gl_acc a0[R], a1[R];
        F g;
    };
    gl_acc acc0[R], acc1[R];
    const uint64_t* apow;
    int k;
    F z_last, l_first, l_last;
    __device__ __forceinline__ void init(const uint64_t* apow_) {
        apow = apow_, k = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) gl_acc_zero(acc0[j]), gl_acc_zero(acc1[j]);
    }
    __device__ __forceinline__ void push(gl_acc* a0, gl_acc* a1, const F& c) {
        const int ku = __builtin_amdgcn_readfirstlane(k);  // the constraint index is wave-uniform by construction
        const uint64_t p0 = apow[2 * ku], p1 = apow[2 * ku + 1];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            gl_mac(a0[j], c.v[j], p0);
            gl_mac(a1[j], c.v[j], p1);
        }
        k = ku + 1;
    }
    __device__ __forceinline__ void constraint(const F& c) { push(acc0, acc1, c); }
    __device__ __forceinline__ void transition(const F& c) { constraint(c * z_last); }
    __device__ __forceinline__ void first_row(const F& c) { constraint(c * l_first); }
    __device__ __forceinline__ void last_row(const F& c) { constraint(c * l_last); }
    __device__ __forceinline__ void constraint_x2(const X2<F>& e) {
        constraint(e.a);
        constraint(e.b);
    }
    __device__ __forceinline__ Gate open(const F& g) {
        Gate G;
        G.g = g;
#pragma unroll
        for (int j = 0; j < R; ++j) gl_acc_zero(G.a0[j]), gl_acc_zero(G.a1[j]);
        return G;
    }
    __device__ __forceinline__ void gated(Gate& G, const F& e) { push(G.a0, G.a1, e); }
    __device__ __forceinline__ void close(Gate& G) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            gl_mac(acc0[j], G.g.v[j], gl_acc_reduce(G.a0[j]));
            gl_mac(acc1[j], G.g.v[j], gl_acc_reduce(G.a1[j]));
        }
    }
    __device__ __forceinline__ uint64_t result(int challenge, int j) const { return gl_acc_reduce(challenge ? acc1[j] : acc0[j]); }
};

// counts the constraints an AIR pushes (host, once per AIR): K of the power table
struct CountF {
    VX_HD CountF operator+(CountF) const { return {}; }
    VX_HD CountF operator-(CountF) const { return {}; }
    VX_HD CountF operator*(CountF) const { return {}; }
    static CountF from(uint64_t) { return {}; }
};
template <>
struct Consumer<CountF> {
    struct Gate {};
    int k = 0;
    void constraint(CountF) { ++k; }
    void transition(CountF) { ++k; }
    void first_row(CountF) { ++k; }
    void last_row(CountF) { ++k; }
    void constraint_x2(const X2<CountF>&) { k += 2; }
    Gate open(CountF) { return {}; }
    void gated(Gate&, CountF) { ++k; }
    void close(Gate&) {}
};
struct CountRow {
    CountF operator[](int) const { return {}; }
};

template <int R>
struct RowViewN {  // column-major LDE, rows i[0..R) of it
    const uint64_t* base;
    size_t stride, i[R];
    __device__ __forceinline__ FpN<R> operator[](int col) const {
        const uint64_t* c = base + (size_t)col * stride;
        FpN<R> r;
#pragma unroll
        for (int j = 0; j < R; ++j) r.v[j] = c[i[j]];
        return r;
    }
};

// ---- AIR 1: Fibonacci (the canonical starky example; used to pin the generic prover) ----
// columns (x0, x1); public inputs (x0[0], x1[0], x1[n-1]); next.x0 = x1, next.x1 = x0 + x1.
struct FibAir {
    static constexpr int ID = 1, COLS = 2, PUB = 3, PERIODIC = 0, PERIOD_LOG = 0, QUOT_ROWS_PER_LANE = 1, AUX = 0, CHAL = 0, AUXPUB = 0, EXACT_LOG = 0;
    static constexpr int plog(int) { return 0; }
    template <class F, class Row, class C>
    VX_HD static void eval(const Row& loc, const Row& nxt, const F*, const F* pub, const F*, const F*, C& c) {
        c.first_row(loc[0] - pub[0]);
        c.first_row(loc[1] - pub[1]);
        c.last_row(loc[1] - pub[2]);
        c.transition(nxt[0] - loc[1]);
        c.transition(nxt[1] - loc[0] - loc[1]);
    }
};

// ---- AIR 2: a degree-3, periodic-column test AIR ("cubic mixer") -------------------------
// 4 columns, period-4 selector s (1,0,0,0) and round constant k (periodic):
//   rows with s = 0: next.a = a*b + k   (gated: degree 3, holds cyclically, no z_last)
//   rows with s = 1: next.a = d         (re-seed; s = (0,0,0,1) so the wrap-around pair re-seeds)
//   next.b = a + b, next.c = c*c + d    (transition: degree 2 * z_last = 3), d boolean
struct MixAir {
    static constexpr int ID = 2, COLS = 4, PUB = 2, PERIODIC = 2, PERIOD_LOG = 2, QUOT_ROWS_PER_LANE = 2, AUX = 0, CHAL = 0, AUXPUB = 0, EXACT_LOG = 0;
    static constexpr int plog(int) { return 2; }
    template <class F, class Row, class C>
    VX_HD static void eval(const Row& loc, const Row& nxt, const F* per, const F* pub, const F*, const F*, C& c) {
        F a = loc[0], b = loc[1], cc = loc[2], d = loc[3];
        F s = per[0], k = per[1];
        F one = F::from(1);
        c.constraint((one - s) * (nxt[0] - a * b - k) + s * (nxt[0] - d));
        c.transition(nxt[1] - a - b);
        c.transition(nxt[2] - cc * cc - d);
        c.constraint(d * (d - one));
        c.first_row(a - pub[0]);
        c.last_row(b - pub[1]);
    }
};

// ---- AIR 5: the smallest AIR with an auxiliary round (pins the logUp machinery; oracle/stark_ref.py LookupAir) ----
// Main: two lookups per row (x0,y0,z0), (x1,y1,z1) claiming z = x ^ y on 4-bit values, and the multiplicity m of the
// table row living in this trace row.  Table (ta, tb, ta ^ tb): periodic, period 2^8.  Challenges beta, gamma (X2).
// Aux: h = 1/(beta+fp0) + 1/(beta+fp1), ht = m/(beta+fp_t), Z with Z(wx) = Z(x) + h(x) - ht(x) cyclically.
struct LookupAir {
    static constexpr int ID = 5, COLS = 7, PUB = 0, PERIODIC = 3, PERIOD_LOG = 8, QUOT_ROWS_PER_LANE = 1, AUX = 6, CHAL = 4, AUXPUB = 0, EXACT_LOG = 0;
    static constexpr int plog(int) { return 8; }
    template <class F, class Row, class C>
    VX_HD static void eval(const Row& loc, const Row& nxt, const F* per, const F*, const F* chal, const F*, C& c) {
        const X2<F> beta{chal[0], chal[1]}, gamma{chal[2], chal[3]}, g2 = gamma * gamma;
        auto fp = [&](const F& a, const F& b, const F& cc) { return beta + a + gamma * b + g2 * cc; };
        const X2<F> d0 = fp(loc[0], loc[1], loc[2]), d1 = fp(loc[3], loc[4], loc[5]), dt = fp(per[0], per[1], per[2]);
        const X2<F> h{loc[7], loc[8]}, ht{loc[9], loc[10]}, z{loc[11], loc[12]}, zn{nxt[11], nxt[12]};
        c.constraint_x2(h * d0 * d1 - d0 - d1);
        c.constraint_x2(ht * dt - loc[6]);
        c.constraint_x2(zn - z - h + ht);
    }
};
