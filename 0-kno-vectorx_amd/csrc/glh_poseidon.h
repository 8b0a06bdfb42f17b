// The ONE host-side Poseidon-Goldilocks permutation and duplex challenger (plonky2 v0.2.0 hash/poseidon.rs,
// iop/challenger.rs: width 12, rate 8, overwrite-mode duplexing) -- used by the prover's transcript (vx_stark.hip) and by
// the host verifiers (vx_verify.hip: transcript, Merkle paths, leaf hashing).  Host code only; the device permutation
// is poseidon.cuh.  Round constants and the circulant MDS row come from the generated poseidon_constants.h.
#pragma once
#include <stdint.h>
#include <string.h>

#include "poseidon_constants.h"
#include "vx_internal.h"

namespace glh {
inline void poseidon(uint64_t* s) {
    static const uint64_t RC[360] = VX_POSEIDON_RC_INIT;
    static const uint64_t MDS[12] = VX_POSEIDON_MDS_CIRC_INIT;
    for (int r = 0; r < 30; ++r) {
        for (int i = 0; i < 12; ++i) s[i] = add(s[i], RC[12 * r + i]);
        const int nsb = (r < 4 || r >= 26) ? 12 : 1;  // 4 + 4 full rounds around 22 partial ones
        for (int i = 0; i < nsb; ++i) {
            const uint64_t x = s[i], x2 = mul(x, x), x3 = mul(x2, x), x4 = mul(x2, x2);
            s[i] = mul(x3, x4);
        }
        uint64_t o[12];
        for (int row = 0; row < 12; ++row) {
            unsigned __int128 acc = 0;
            for (int i = 0; i < 12; ++i) acc += (unsigned __int128)s[(i + row) % 12] * MDS[i];
            if (row == 0) acc += (unsigned __int128)s[0] * VX_POSEIDON_MDS_DIAG0;
            o[row] = reduce128(acc);
        }
        memcpy(s, o, sizeof o);
    }
}
struct Challenger {
    uint64_t st[12] = {0}, in[8], out[8];
    int n_in = 0, n_out = 0;
    void duplex() {
        for (int i = 0; i < n_in; ++i) st[i] = in[i];
        n_in = 0;
        poseidon(st);
        memcpy(out, st, sizeof out);
        n_out = 8;
    }
    void observe(uint64_t x) {
        n_out = 0;
        in[n_in++] = x;
        if (n_in == 8) duplex();
    }
    void observe(const uint64_t* x, size_t n) {
        for (size_t i = 0; i < n; ++i) observe(x[i]);
    }
    uint64_t challenge() {
        if (n_in > 0 || n_out == 0) duplex();
        return out[--n_out];
    }
};
}  // namespace glh
