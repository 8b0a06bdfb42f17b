/*
 * oracle/goldilocks.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Goldilocks field F_p, p = 2^64 - 2^32 + 1, and its quadratic extension
 * F_p[X]/(X^2 - 7), restated from the published definitions used by
 * plonky2 v0.2.0 (git 0xPolygonZero/plonky2 #7445ec91, pinned by the
 * reference at Cargo.lock:4848-4873; crate `plonky2_field`:
 * goldilocks_field.rs, extension/quadratic.rs).  The crate is NOT vendored in
 * /root/reference; the reference's call sites that fix the field are
 * circuits/builder/header.rs:47 (`type F = GoldilocksField`) and
 * circuits/builder/subchain_verification.rs:448 (`const D: usize = 2`).
 *
 * All values at API edges are canonical (< p) little-endian uint64_t.
 */
#ifndef VXO_GOLDILOCKS_H
#define VXO_GOLDILOCKS_H
#include <stdint.h>

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL /* 2^64 mod p = 2^32 - 1 */
/* multiplicative generator g = 7; omega_{2^32} = 7^((p-1)/2^32) */
#define GL_GEN 7ULL
#define GL_ROOT_2_32 1753635133440165772ULL
#define GL_EXT_W 7ULL /* X^2 = 7 */

static inline uint64_t gl_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a || s >= GL_P) s -= GL_P;
    return s;
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b) {
    return a >= b ? a - b : a + (GL_P - b);
}
static inline uint64_t gl_neg(uint64_t a) { return a ? GL_P - a : 0; }
static inline uint64_t gl_mul(uint64_t a, uint64_t b) {
    return (uint64_t)(((unsigned __int128)a * b) % GL_P);
}
static inline uint64_t gl_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, a);
        a = gl_mul(a, a);
        e >>= 1;
    }
    return r;
}
static inline uint64_t gl_inv(uint64_t a) { return gl_pow(a, GL_P - 2); }
/* primitive 2^k-th root of unity, plonky2 Field::primitive_root_of_unity */
static inline uint64_t gl_root(int log_n) {
    uint64_t r = GL_ROOT_2_32;
    for (int i = 32; i > log_n; --i) r = gl_mul(r, r);
    return r;
}

typedef struct { uint64_t c[2]; } gl2_t;
static inline gl2_t gl2_add(gl2_t a, gl2_t b) {
    gl2_t r = {{gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1])}};
    return r;
}
static inline gl2_t gl2_sub(gl2_t a, gl2_t b) {
    gl2_t r = {{gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1])}};
    return r;
}
static inline gl2_t gl2_mul(gl2_t a, gl2_t b) {
    gl2_t r;
    r.c[0] = gl_add(gl_mul(a.c[0], b.c[0]), gl_mul(GL_EXT_W, gl_mul(a.c[1], b.c[1])));
    r.c[1] = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    return r;
}
static inline gl2_t gl2_scale(gl2_t a, uint64_t s) {
    gl2_t r = {{gl_mul(a.c[0], s), gl_mul(a.c[1], s)}};
    return r;
}
static inline gl2_t gl2_from(uint64_t a) {
    gl2_t r = {{a, 0}};
    return r;
}
static inline gl2_t gl2_inv(gl2_t a) {
    /* (a0 + a1 X)^-1 = (a0 - a1 X) / (a0^2 - 7 a1^2) */
    uint64_t n = gl_sub(gl_mul(a.c[0], a.c[0]), gl_mul(GL_EXT_W, gl_mul(a.c[1], a.c[1])));
    uint64_t ni = gl_inv(n);
    gl2_t r = {{gl_mul(a.c[0], ni), gl_mul(gl_neg(a.c[1]), ni)}};
    return r;
}
static inline gl2_t gl2_pow(gl2_t a, uint64_t e) {
    gl2_t r = gl2_from(1);
    while (e) {
        if (e & 1) r = gl2_mul(r, a);
        a = gl2_mul(a, a);
        e >>= 1;
    }
    return r;
}
static inline int gl2_eq(gl2_t a, gl2_t b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1]; }
#endif
