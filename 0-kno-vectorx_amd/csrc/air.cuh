// AIR interface shared by the GPU quotient kernel (base field, one LDE point per lane) and
// the host verifier (extension field, at zeta).  An AIR is a struct with
//   static constexpr int COLS, PUB, PERIODIC, PERIOD_LOG;
//   static void periodic_values(std::vector<uint64_t>& out);   // [PERIODIC][1 << PERIOD_LOG], host
//   template <class F, class Row, class C> static void eval(const Row& loc, const Row& nxt,
//                                                           const F* per, const F* pub, C& c);
// `eval` pushes constraints into the consumer IN A FIXED ORDER (the order is part of the
// protocol: the oracle restates it independently in oracle/stark_ref.py).
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

#include "gl.cuh"

#define VX_HD __host__ __device__ __forceinline__

struct Fp {  // device base-field element
    uint64_t v;
    __device__ __forceinline__ Fp operator+(Fp o) const { return {gl_add(v, o.v)}; }
    __device__ __forceinline__ Fp operator-(Fp o) const { return {gl_sub(v, o.v)}; }
    __device__ __forceinline__ Fp operator*(Fp o) const { return {gl_mul(v, o.v)}; }
    __device__ __forceinline__ static Fp from(uint64_t x) { return {x}; }
};

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

template <class F>
struct Consumer {
    F acc[2], alpha[2], z_last, l_first, l_last;
    VX_HD void constraint(F c) {
        acc[0] = acc[0] * alpha[0] + c;
        acc[1] = acc[1] * alpha[1] + c;
    }
    VX_HD void transition(F c) { constraint(c * z_last); }
    VX_HD void first_row(F c) { constraint(c * l_first); }
    VX_HD void last_row(F c) { constraint(c * l_last); }
};

struct RowView {  // column-major LDE, one row
    const uint64_t* base;
    size_t stride, i;
    __device__ __forceinline__ Fp operator[](int col) const { return {base[(size_t)col * stride + i]}; }
};

// ---- AIR 1: Fibonacci (the canonical starky example; used to pin the generic prover) ----
// columns (x0, x1); public inputs (x0[0], x1[0], x1[n-1]); next.x0 = x1, next.x1 = x0 + x1.
struct FibAir {
    static constexpr int ID = 1, COLS = 2, PUB = 3, PERIODIC = 0, PERIOD_LOG = 0;
    template <class F, class Row, class C>
    VX_HD static void eval(const Row& loc, const Row& nxt, const F*, const F* pub, C& c) {
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
    static constexpr int ID = 2, COLS = 4, PUB = 2, PERIODIC = 2, PERIOD_LOG = 2;
    template <class F, class Row, class C>
    VX_HD static void eval(const Row& loc, const Row& nxt, const F* per, const F* pub, C& c) {
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
