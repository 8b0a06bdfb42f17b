// Run-time AIR descriptors (include/vx.h vx_air_program): the registered form of a constraint program and the host
// interpreter the verifier evaluates it with.  The GPU interpreter is k_quotient_prog in vx_stark.hip; the registry and the
// checks of vx_air_register live in vx_verify.hip (host-only code: the verifier must work without a GPU).  The instruction
// set stands in for starky's `Stark::eval_packed_generic` / `eval_ext_circuit` pair (starky v0.2.0 stark.rs), which the
// reference reaches through curta's AirParser -- one definition, two evaluators.
#pragma once
#include <stdint.h>

#include <vector>

#include "vx_internal.h"  // include/vx.h with default visibility

struct AirProgram {
    int id = 0;
    uint32_t cols = 0, pub = 0, n_regs = 0, n_constraints = 0;
    int period_log = 0;             // the largest period
    std::vector<uint8_t> plog;      // per periodic column
    std::vector<uint64_t> periodic; // one period of every periodic column, back to back
    std::vector<uint64_t> consts, code;
    uint32_t aux = 0, chal = 0, auxpub = 0;  // auxiliary round: columns, base-field challenges, published extension values
    vx_air_gen_aux_fn gen_aux = nullptr;     // the host's witness generator of that round (prover only)
    void* gen_aux_user = nullptr;
};
// nullptr when the id is unknown or retired; a registered program is never freed
const AirProgram* vx_air_program_find(int id);

struct AirpInsn {
    int op, d, a, b;
};
static inline AirpInsn airp_decode(uint64_t w) { return {(int)(w & 0xFF), (int)((w >> 8) & 0xFF), (int)((w >> 16) & 0xFFFF), (int)((w >> 32) & 0xFFFF)}; }

// host evaluation over any field type with + - * (the verifier's Fx); C = the constraint consumer of air.cuh
template <class F, class Row, class C>
static void air_program_eval(const AirProgram& p, const Row& loc, const Row& nxt, const F* per, const F* pub, const F* chal, const F* apub, C& c) {
    std::vector<F> r(p.n_regs ? p.n_regs : 1);
    for (uint64_t w : p.code) {
        const AirpInsn i = airp_decode(w);
        switch (i.op) {
            case VX_AIRP_LOC: r[i.d] = loc[i.a]; break;
            case VX_AIRP_NXT: r[i.d] = nxt[i.a]; break;
            case VX_AIRP_PER: r[i.d] = per[i.a]; break;
            case VX_AIRP_PUB: r[i.d] = pub[i.a]; break;
            case VX_AIRP_CONST: r[i.d] = F::from(p.consts[i.a]); break;
            case VX_AIRP_CHAL: r[i.d] = chal[i.a]; break;
            case VX_AIRP_APUB: r[i.d] = apub[i.a]; break;
            case VX_AIRP_ADD: r[i.d] = r[i.a] + r[i.b]; break;
            case VX_AIRP_SUB: r[i.d] = r[i.a] - r[i.b]; break;
            case VX_AIRP_MUL: r[i.d] = r[i.a] * r[i.b]; break;
            case VX_AIRP_ASSERT: c.constraint(r[i.a]); break;
            case VX_AIRP_ASSERT_TRANSITION: c.transition(r[i.a]); break;
            case VX_AIRP_ASSERT_FIRST: c.first_row(r[i.a]); break;
            case VX_AIRP_ASSERT_LAST: c.last_row(r[i.a]); break;
            default: break;  // unreachable: vx_air_register admits no other opcode
        }
    }
}
