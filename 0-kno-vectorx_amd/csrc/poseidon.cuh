// Poseidon-Goldilocks permutation (width 12, x^7, 4 + 22 + 4 rounds) for gfx950.
// Replaces plonky2::hash::poseidon::Poseidon::poseidon (v0.2.0), the hash behind
// every Merkle cap and the Fiat-Shamir challenger of the reference's prover
// (AlgebraicHasher bound at circuits/header_range.rs:28-29).
//
// One permutation per lane, state in 24 VGPRs.  The MDS layer never forms
// 128-bit products: each state word is split into 32-bit halves, the circulant
// row sums of the halves stay below 2^41, and the two sums are recombined with
// 2^64 = 2^32 - 1 (mod p).
#pragma once
#include "gl.cuh"
#include "poseidon_constants.h"

static __constant__ uint64_t POSEIDON_RC[360] = VX_POSEIDON_RC_INIT;

__device__ __forceinline__ uint64_t poseidon_sbox(uint64_t x) {
    uint64_t x2 = gl_sqr(x), x3 = gl_mul(x2, x), x4 = gl_sqr(x2);
    return gl_mul(x3, x4);
}

__device__ __forceinline__ void poseidon_mds(uint64_t* s) {
    constexpr uint32_t C[12] = VX_POSEIDON_MDS_CIRC_INIT;
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
#pragma unroll
    for (int r = 0; r < 12; ++r) {
        uint64_t al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            al += (uint64_t)lo[(i + r) % 12] * C[i];
            ah += (uint64_t)hi[(i + r) % 12] * C[i];
        }
        if (r == 0) {
            al += (uint64_t)lo[0] * VX_POSEIDON_MDS_DIAG0;
            ah += (uint64_t)hi[0] * VX_POSEIDON_MDS_DIAG0;
        }
        // al + ah * 2^32, ah < 2^42:  ah*2^32 = (ah>>32)*2^64 + (ah & eps) << 32
        uint64_t t = gl_add(al, (ah & GL_EPS) << 32);
        s[r] = gl_add(t, (ah >> 32) * GL_EPS);
    }
}

__device__ __forceinline__ void poseidon_permute(uint64_t* s) {
    int rc = 0;
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox(gl_add(s[i], POSEIDON_RC[rc + i]));
        rc += 12;
        poseidon_mds(s);
    }
#pragma unroll 1
    for (int r = 0; r < 22; ++r) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = gl_add(s[i], POSEIDON_RC[rc + i]);
        rc += 12;
        s[0] = poseidon_sbox(s[0]);
        poseidon_mds(s);
    }
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 12; ++i) s[i] = poseidon_sbox(gl_add(s[i], POSEIDON_RC[rc + i]));
        rc += 12;
        poseidon_mds(s);
    }
}
