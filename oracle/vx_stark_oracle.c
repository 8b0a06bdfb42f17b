/*
 * oracle/vx_stark_oracle.c -- TEST INFRASTRUCTURE: coefficient-space helpers for the
 * reference STARK prover restated in oracle/stark_ref.py (plonky2 v0.2.0
 * fri/oracle.rs PolynomialBatch::prove_openings, util/reducing.rs ReducingFactor,
 * field/polynomial: divide_by_linear, eval).  Never linked into the product.
 */
#include "goldilocks.h"
#include <stdlib.h>
#include <string.h>
#define EXPORT __attribute__((visibility("default")))

/* PolynomialCoeffs::eval at an extension point: Horner from the top coefficient. */
EXPORT void vxo_poly_eval_ext(const uint64_t* coeffs, size_t n, const uint64_t* z, uint64_t* out) {
    gl2_t zz = {{z[0], z[1]}}, acc = gl2_from(0);
    for (size_t k = n; k-- > 0;) acc = gl2_add(gl2_mul(acc, zz), gl2_from(coeffs[k]));
    out[0] = acc.c[0];
    out[1] = acc.c[1];
}
EXPORT void vxo_ext_poly_eval_ext(const uint64_t* coeffs_ext, size_t n, const uint64_t* z, uint64_t* out) {
    gl2_t zz = {{z[0], z[1]}}, acc = gl2_from(0);
    for (size_t k = n; k-- > 0;) {
        gl2_t c = {{coeffs_ext[2 * k], coeffs_ext[2 * k + 1]}};
        acc = gl2_add(gl2_mul(acc, zz), c);
    }
    out[0] = acc.c[0];
    out[1] = acc.c[1];
}
/* ReducingFactor::reduce_polys_base: sum_j alpha^j * p_j  (base-field polys, extension alpha). */
EXPORT void vxo_reduce_polys_base(const uint64_t* polys, size_t n_polys, size_t n, const uint64_t* alpha, uint64_t* out_ext) {
    gl2_t a = {{alpha[0], alpha[1]}};
    memset(out_ext, 0, 2 * n * sizeof(uint64_t));
    for (size_t j = n_polys; j-- > 0;) { /* fold from the last: acc = acc * alpha + p_j */
        for (size_t k = 0; k < n; ++k) {
            gl2_t acc = {{out_ext[2 * k], out_ext[2 * k + 1]}};
            acc = gl2_add(gl2_mul(acc, a), gl2_from(polys[j * n + k]));
            out_ext[2 * k] = acc.c[0];
            out_ext[2 * k + 1] = acc.c[1];
        }
    }
}
/* PolynomialCoeffs::divide_by_linear(z): (P(X) - P(z)) / (X - z), n-1 coefficients, then one zero pushed. */
EXPORT void vxo_ext_divide_by_linear(const uint64_t* p_ext, size_t n, const uint64_t* z, uint64_t* out_ext) {
    gl2_t zz = {{z[0], z[1]}}, carry = gl2_from(0);
    /* synthetic division from the top: b_{k-1} = a_k + z * b_k */
    for (size_t k = n; k-- > 1;) {
        gl2_t a = {{p_ext[2 * k], p_ext[2 * k + 1]}};
        carry = gl2_add(a, gl2_mul(zz, carry));
        out_ext[2 * (k - 1)] = carry.c[0];
        out_ext[2 * (k - 1) + 1] = carry.c[1];
    }
    out_ext[2 * (n - 1)] = 0;
    out_ext[2 * (n - 1) + 1] = 0;
}
/* out = a * s + b  (extension polys, extension scalar s) */
EXPORT void vxo_ext_poly_scale_add(const uint64_t* a, const uint64_t* s, const uint64_t* b, size_t n, uint64_t* out) {
    gl2_t ss = {{s[0], s[1]}};
    for (size_t k = 0; k < n; ++k) {
        gl2_t x = {{a[2 * k], a[2 * k + 1]}}, y = {{b[2 * k], b[2 * k + 1]}};
        gl2_t r = gl2_add(gl2_mul(x, ss), y);
        out[2 * k] = r.c[0];
        out[2 * k + 1] = r.c[1];
    }
}
