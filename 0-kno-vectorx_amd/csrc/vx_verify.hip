// Host-side verifier for the STARK proofs vx_stark_prove emits: the `circuit.verify` half of
// the reference's library seam (/root/reference circuits/header_range.rs:170).  Mirrors starky
// v0.2.0 verify_stark_proof_with_challenges + plonky2 v0.2.0 verify_fri_proof
// (fri_combine_initial, compute_evaluation, verify_merkle_proof_to_cap); crates pinned at
// Cargo.lock:4848-4905, not vendored.  Verification is cheap scalar work (a few thousand
// Poseidon permutations) and stays on the host, as in the reference.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <memory>
#include <mutex>

#include "air.cuh"
#include "air_blake.cuh"
#include "air_sha.cuh"
#include "air_ed.cuh"
#include "air_epoch.cuh"
#include "air_sha512.cuh"
#include "air_program.h"
#include "air_sha_tree.cuh"
#include "glh_poseidon.h"
#include "vx_bus.h"
#include "vx_internal.h"

namespace {
inline void v_poseidon(uint64_t* s) { glh::poseidon(s); }
void v_hash_or_noop(const uint64_t* in, size_t len, uint64_t* out4) {
    uint64_t s[12] = {0};
    if (len <= 4) {
        memcpy(s, in, len * 8);
    } else {
        for (size_t off = 0; off < len; off += 8) {
            const size_t k = len - off < 8 ? len - off : 8;
            memcpy(s, in + off, k * 8);
            v_poseidon(s);
        }
    }
    memcpy(out4, s, 32);
}
bool v_merkle(const uint64_t* leaf, size_t leaf_len, size_t idx, const uint64_t* sib, size_t n_sib, const uint64_t* cap) {
    uint64_t cur[4];
    v_hash_or_noop(leaf, leaf_len, cur);
    for (size_t k = 0; k < n_sib; ++k) {
        uint64_t s[12] = {0};
        if (idx & 1) {
            memcpy(s, sib + 4 * k, 32);
            memcpy(s + 4, cur, 32);
        } else {
            memcpy(s, cur, 32);
            memcpy(s + 4, sib + 4 * k, 32);
        }
        v_poseidon(s);
        memcpy(cur, s, 32);
        idx >>= 1;
    }
    return memcmp(cur, cap + 4 * idx, 32) == 0;
}
struct VChallenger : glh::Challenger {
    Fx ext() {
        const uint64_t a = challenge(), b = challenge();
        return {a, b};
    }
};
Fx fx_inv(Fx x) {
    const uint64_t n = glh::sub(glh::mul(x.a, x.a), glh::mul(7, glh::mul(x.b, x.b)));
    const uint64_t ni = glh::inv(n);
    return {glh::mul(x.a, ni), glh::mul(glh::sub(0, x.b), ni)};
}
Fx fx_pow(Fx x, uint64_t e) {
    Fx r{1, 0};
    while (e) {
        if (e & 1) r = r * x;
        x = x * x;
        e >>= 1;
    }
    return r;
}
bool fx_eq(Fx x, Fx y) { return x.a == y.a && x.b == y.b; }

struct AirV {
    int id, cols, pub, periodic, period_log, exact_log;
    void (*periodic_values)(std::vector<uint64_t>&);
    void (*eval)(const HostRow&, const HostRow&, const Fx*, const Fx*, const Fx*, const Fx*, Consumer<Fx>&);
    int aux, chal, auxpub;
    int (*plog)(int);
    const AirProgram* prog = nullptr;  // a registered constraint program instead of a compiled AIR
};
template <class Air>
void eval_host(const HostRow& l, const HostRow& n, const Fx* per, const Fx* pub, const Fx* chal, const Fx* apub, Consumer<Fx>& c) {
    Air::template eval<Fx>(l, n, per, pub, chal, apub, c);
}
template <class Air>
AirV vdesc(void (*pv)(std::vector<uint64_t>&)) {
    return {Air::ID, Air::COLS, Air::PUB, Air::PERIODIC, Air::PERIOD_LOG, Air::EXACT_LOG, pv, eval_host<Air>, Air::AUX, Air::CHAL, Air::AUXPUB, Air::plog};
}
// coefficients of P(Y), deg < p, with P(w_p^k) = v[k]: in-place radix-2 inverse NTT on the host (a 2^16-entry lookup
// table is far too long for the O(p^2) sum the short selectors get away with)
void host_intt(std::vector<uint64_t>& a) {
    const size_t p = a.size();
    int lg = 0;
    while (((size_t)1 << lg) < p) ++lg;
    for (size_t i = 0; i < p; ++i) {
        size_t j = 0;
        for (int b = 0; b < lg; ++b) j |= ((i >> b) & 1) << (lg - 1 - b);
        if (j > i) std::swap(a[i], a[j]);
    }
    for (int st = 1; st <= lg; ++st) {
        const size_t half = (size_t)1 << (st - 1);
        const uint64_t wl = glh::inv(glh::root(st));
        for (size_t blk = 0; blk < p; blk += 2 * half) {
            uint64_t w = 1;
            for (size_t k = 0; k < half; ++k) {
                const uint64_t u = a[blk + k], t = glh::mul(a[blk + k + half], w);
                a[blk + k] = glh::add(u, t);
                a[blk + k + half] = glh::sub(u, t);
                w = glh::mul(w, wl);
            }
        }
    }
    const uint64_t pinv = glh::inv(p % glh::P);
    for (uint64_t& x : a) x = glh::mul(x, pinv);
}
void v_no_periodic(std::vector<uint64_t>& v) { v.clear(); }
void v_mix_periodic(std::vector<uint64_t>& v) { v = {0, 0, 0, 1, 3, 5, 7, 11}; }
void v_blake_periodic(std::vector<uint64_t>& v) {
    v.assign(16 * 16 + 4 * 65536, 0);
    for (int k = 0; k < 16; ++k) v[k * 16 + k] = 1;
    for (uint64_t i = 0; i < 65536; ++i) {
        const uint64_t a = i & 255, b = i >> 8;
        v[256 + i] = a, v[256 + 65536 + i] = b, v[256 + 2 * 65536 + i] = (a ^ b) & 127, v[256 + 3 * 65536 + i] = (a ^ b) >> 7;
    }
}
void v_lookup_periodic(std::vector<uint64_t>& v) {
    v.resize(3 * 256);
    for (int i = 0; i < 256; ++i) v[i] = i & 15, v[256 + i] = i >> 4, v[512 + i] = (i & 15) ^ (i >> 4);
}
const AirV V_AIRS[] = {
    vdesc<ShaAir>(ShaAir::periodic_values), vdesc<FibAir>(v_no_periodic), vdesc<MixAir>(v_mix_periodic), vdesc<BlakeAir>(v_blake_periodic),
    vdesc<LookupAir>(v_lookup_periodic), vdesc<ShaTreeAir256>(ShaTreeAir256::periodic_values), vdesc<ShaTreeAir512>(ShaTreeAir512::periodic_values),
    vdesc<ShaTreeAir16>(ShaTreeAir16::periodic_values), vdesc<EdAir17>(EdAir17::periodic_values), vdesc<EdAir16>(EdAir16::periodic_values), vdesc<Sha512Air16>(Sha512Air16::periodic_values), vdesc<Sha512Air10>(Sha512Air10::periodic_values), vdesc<Sha512Air15>(Sha512Air15::periodic_values), vdesc<EpochEndAir>(EpochEndAir::periodic_values),
};
size_t brev(size_t x, int bits) {
    size_t r = 0;
    for (int i = 0; i < bits; ++i) r = (r << 1) | ((x >> i) & 1);
    return r;
}
int32_t v_fail(char* err, size_t errlen, const char* fmt, ...) {
    if (err && errlen) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, errlen, fmt, ap);
        va_end(ap);
    }
    return VX_ERR_STATEMENT;
}
#define NEED(cond, ...) \
    do {                \
        if (!(cond)) return v_fail(err, errlen, __VA_ARGS__); \
    } while (0)
}  // namespace


// ---- run-time AIR descriptors: the registry (process-wide; programs are immutable and never freed, ids never reused)
namespace {
std::mutex g_airp_mu;
std::map<int, std::shared_ptr<const AirProgram>> g_airp;       // live ids
std::vector<std::shared_ptr<const AirProgram>> g_airp_retired;  // kept alive: a prover may still hold the pointer
int g_airp_next = VX_AIR_USER_BASE;
int32_t airp_fail(char* err, size_t errlen, const char* fmt, ...) {
    if (err && errlen) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, errlen, fmt, ap);
        va_end(ap);
    }
    return VX_ERR_ARG;
}
#define AIRP_NEED(cond, ...) \
    do {                     \
        if (!(cond)) return airp_fail(err, errlen, __VA_ARGS__); \
    } while (0)
}  // namespace
const AirProgram* vx_air_program_find(int id) {
    std::lock_guard<std::mutex> lk(g_airp_mu);
    auto it = g_airp.find(id);
    return it == g_airp.end() ? nullptr : it->second.get();
}
extern "C" {
int32_t vx_air_register(const vx_air_program* in, int* air_id, char* err, size_t errlen) {
    if (!in || !air_id) return VX_ERR_ARG;
    AIRP_NEED(in->cols >= 1 && in->cols <= VX_AIRP_MAX_COLS, "air program: %u columns (1..%d)", in->cols, VX_AIRP_MAX_COLS);
    AIRP_NEED(in->n_public <= 64, "air program: %u public inputs (at most 64)", in->n_public);
    AIRP_NEED(in->n_periodic <= 256 && (in->n_periodic == 0 || (in->periodic_log && in->periodic_values)), "air program: bad periodic columns");
    AIRP_NEED(in->n_regs >= 1 && in->n_regs <= VX_AIRP_MAX_REGS, "air program: %u registers (1..%d)", in->n_regs, VX_AIRP_MAX_REGS);
    AIRP_NEED(in->n_consts <= 65536 && (in->n_consts == 0 || in->consts), "air program: bad constant table");
    AIRP_NEED(in->code && in->n_code >= 1 && in->n_code <= VX_AIRP_MAX_CODE, "air program: %u instructions (1..%d)", in->n_code, VX_AIRP_MAX_CODE);
    AIRP_NEED(in->aux_cols <= VX_AIRP_MAX_COLS && in->cols + in->aux_cols <= VX_AIRP_MAX_COLS && in->n_challenges <= 8 && in->n_aux_public <= 4,
              "air program: auxiliary round out of range (%u columns, %u challenges <= 8, %u published values <= 4)", in->aux_cols, in->n_challenges, in->n_aux_public);
    AIRP_NEED(in->aux_cols > 0 || (in->n_challenges == 0 && in->n_aux_public == 0 && !in->gen_aux), "air program: challenges / published values / a generator without auxiliary columns");
    AIRP_NEED(in->aux_cols == 0 || in->n_challenges > 0, "air program: an auxiliary round without a challenge is a second trace commitment, not a lookup round");
    auto pg = std::make_shared<AirProgram>();
    pg->cols = in->cols, pg->pub = in->n_public, pg->n_regs = in->n_regs;
    pg->aux = in->aux_cols, pg->chal = in->n_challenges, pg->auxpub = in->n_aux_public, pg->gen_aux = in->gen_aux, pg->gen_aux_user = in->gen_aux_user;
    size_t n_per_values = 0;
    for (uint32_t q = 0; q < in->n_periodic; ++q) {
        AIRP_NEED(in->periodic_log[q] <= VX_AIRP_MAX_PERIOD_LOG, "air program: periodic column %u has period 2^%u (at most 2^%d)", q, in->periodic_log[q], VX_AIRP_MAX_PERIOD_LOG);
        pg->plog.push_back(in->periodic_log[q]);
        if (in->periodic_log[q] > pg->period_log) pg->period_log = in->periodic_log[q];
        n_per_values += (size_t)1 << in->periodic_log[q];
    }
    pg->periodic.assign(in->periodic_values, in->periodic_values + n_per_values);
    for (uint64_t v : pg->periodic) AIRP_NEED(v < glh::P, "air program: non-canonical periodic value");
    pg->consts.assign(in->consts, in->consts + in->n_consts);
    for (uint64_t v : pg->consts) AIRP_NEED(v < glh::P, "air program: non-canonical constant");
    pg->code.assign(in->code, in->code + in->n_code);
    // operands, def-before-use and degrees by abstract interpretation (degree of a register: columns and periodic columns 1)
    std::vector<int> deg(in->n_regs, -1);
    for (uint32_t pc = 0; pc < in->n_code; ++pc) {
        const uint64_t w = pg->code[pc];
        const AirpInsn i = airp_decode(w);
        AIRP_NEED((w >> 48) == 0, "air program: instruction %u has reserved bits set", pc);
        auto reg_ok = [&](int r) { return r >= 0 && r < (int)in->n_regs; };
        auto src = [&](int r) { return reg_ok(r) && deg[r] >= 0; };
        switch (i.op) {
            case VX_AIRP_LOC:
            case VX_AIRP_NXT:
                AIRP_NEED(reg_ok(i.d) && i.a < (int)(in->cols + in->aux_cols) && i.b == 0, "air program: instruction %u: bad column load", pc);
                deg[i.d] = 1;
                break;
            case VX_AIRP_PER:
                AIRP_NEED(reg_ok(i.d) && i.a < (int)in->n_periodic && i.b == 0, "air program: instruction %u: bad periodic load", pc);
                deg[i.d] = 1;
                break;
            case VX_AIRP_PUB:
                AIRP_NEED(reg_ok(i.d) && i.a < (int)in->n_public && i.b == 0, "air program: instruction %u: bad public-input load", pc);
                deg[i.d] = 0;
                break;
            case VX_AIRP_CONST:
                AIRP_NEED(reg_ok(i.d) && i.a < (int)in->n_consts && i.b == 0, "air program: instruction %u: bad constant load", pc);
                deg[i.d] = 0;
                break;
            case VX_AIRP_CHAL:
                AIRP_NEED(reg_ok(i.d) && i.a < (int)in->n_challenges && i.b == 0, "air program: instruction %u: bad challenge load", pc);
                deg[i.d] = 0;
                break;
            case VX_AIRP_APUB:
                AIRP_NEED(reg_ok(i.d) && i.a < (int)(2 * in->n_aux_public) && i.b == 0, "air program: instruction %u: bad published-value load", pc);
                deg[i.d] = 0;
                break;
            case VX_AIRP_ADD:
            case VX_AIRP_SUB:
            case VX_AIRP_MUL: {
                AIRP_NEED(reg_ok(i.d) && src(i.a) && src(i.b), "air program: instruction %u reads a register that was never written (or out of range)", pc);
                const int dg = i.op == VX_AIRP_MUL ? deg[i.a] + deg[i.b] : (deg[i.a] > deg[i.b] ? deg[i.a] : deg[i.b]);
                deg[i.d] = dg > 1000 ? 1000 : dg;
                break;
            }
            case VX_AIRP_ASSERT:
            case VX_AIRP_ASSERT_TRANSITION:
            case VX_AIRP_ASSERT_FIRST:
            case VX_AIRP_ASSERT_LAST: {
                AIRP_NEED(i.d == 0 && i.b == 0 && src(i.a), "air program: instruction %u asserts a register that was never written (or out of range)", pc);
                const int lim = i.op == VX_AIRP_ASSERT ? 3 : 2;
                AIRP_NEED(deg[i.a] <= lim, "air program: instruction %u asserts an expression of degree %d (limit %d at rate_bits 1)", pc, deg[i.a], lim);
                ++pg->n_constraints;
                break;
            }
            default: AIRP_NEED(false, "air program: instruction %u has unknown opcode %d", pc, i.op);
        }
    }
    AIRP_NEED(pg->n_constraints >= 1, "air program: no constraint");
    std::lock_guard<std::mutex> lk(g_airp_mu);
    AIRP_NEED(g_airp_next < 0x7FFFFFF0, "air program: id space exhausted");
    pg->id = g_airp_next++;
    *air_id = pg->id;
    g_airp[pg->id] = pg;
    return VX_OK;
}
int32_t vx_air_unregister(int air_id) {
    std::lock_guard<std::mutex> lk(g_airp_mu);
    auto it = g_airp.find(air_id);
    if (it == g_airp.end()) return VX_ERR_ARG;
    g_airp_retired.push_back(it->second);
    g_airp.erase(it);
    return VX_OK;
}
}  // extern "C"

extern "C" {

int32_t vx_stark_verify(const vx_stark_config* cfg, const uint64_t* pr, size_t len, int expect_air,
                        const uint64_t* expect_public, size_t n_expect_public, char* err, size_t errlen) {
    return vx_stark_verify_ext(cfg, pr, len, expect_air, expect_public, n_expect_public, nullptr, nullptr, nullptr, err, errlen);
}
}  // extern "C"

bool vx_stark_proof_peek(const uint64_t* pr, size_t len, int cap_height, const uint64_t** pub, size_t* n_pub, const uint64_t** cap) {
    if (len < 12 || pr[9] > 16) return false;
    const size_t pos = 10 + pr[9];
    if (pos + 2 > len) return false;
    const size_t np = pr[pos + 1];
    if (np > 64 || cap_height < 0 || cap_height > 16 || pos + 2 + np + ((size_t)4 << cap_height) > len) return false;
    *pub = pr + pos + 2, *n_pub = np, *cap = pr + pos + 2 + np;
    return true;
}
void v_shared_challenges_n(const uint64_t* const* pubs, const size_t* n_pubs, const uint64_t* const* caps, size_t k, size_t cap_words, uint64_t* out, size_t n_out) {
    VChallenger sc;
    for (size_t t = 0; t < k; ++t) {
        sc.observe(pubs[t], n_pubs[t]);
        sc.observe(caps[t], cap_words);
    }
    for (size_t q = 0; q < n_out; ++q) out[q] = sc.challenge();
}

int32_t vx_stark_verify_ext(const vx_stark_config* cfg, const uint64_t* pr, size_t len, int expect_air, const uint64_t* expect_public,
                            size_t n_expect_public, const uint64_t* ext_chal, const uint64_t** apub_out, int* log_n_out, char* err, size_t errlen) {
    if (!cfg || !pr) return VX_ERR_ARG;
    // the configuration steers loops below (a circuit.json can carry it): the same ranges the prover accepts (vx_stark.hip)
    if (cfg->rate_bits < 1 || cfg->rate_bits > 3 || cfg->arity_bits < 1 || cfg->arity_bits > 5 || cfg->final_poly_bits < 0 || cfg->final_poly_bits > 27 ||
        cfg->num_queries < 1 || cfg->num_queries > 1024 || cfg->pow_bits < 0 || cfg->pow_bits > 32 || cfg->cap_height < 0 || cfg->cap_height > 27) {
        if (err && errlen) snprintf(err, errlen, "stark verify: configuration out of range");
        return VX_ERR_ARG;
    }
    size_t pos = 0;
    auto have = [&](size_t k) { return pos + k <= len; };
    NEED(have(10), "proof truncated (header)");
    NEED(pr[0] == 0x314b524154535856ULL, "bad magic");
    // (every narrow header field is compared as the 64-bit word it is: a proof has ONE encoding)
    NEED(((pr[1] | pr[2] | pr[5] | pr[6] | pr[7] | pr[8]) >> 31) == 0, "header word out of range");
    const int air_id = (int)pr[1], L = (int)pr[2];
    const size_t nq = pr[4];
    const int r = (int)pr[5], cap_h = (int)pr[6];
    const size_t n_queries = pr[7];
    const int pow_bits = (int)pr[8];
    const size_t n_layers = pr[9];
    pos = 10;
    NEED(r == cfg->rate_bits && cap_h == cfg->cap_height && (int)n_queries == cfg->num_queries && pow_bits == cfg->pow_bits, "config mismatch");
    NEED(L >= 2 && L <= 26 && n_layers <= 16, "bad shape");
    const AirV* air = nullptr;
    for (const AirV& a : V_AIRS)
        if (a.id == air_id) air = &a;
    AirV prog_air{};
    if (!air && air_id >= VX_AIR_USER_BASE) {
        if (const AirProgram* pg = vx_air_program_find(air_id)) {
            prog_air = {air_id, (int)pg->cols, (int)pg->pub, (int)pg->plog.size(), pg->period_log, 0, nullptr, nullptr, (int)pg->aux, (int)pg->chal, (int)pg->auxpub, nullptr, pg};
            air = &prog_air;
        }
    }
    NEED(air && (expect_air == 0 || expect_air == air_id), "unexpected AIR %d", air_id);
    const size_t cm = air->cols, ca = air->aux, c = cm + ca;  // main ++ auxiliary columns
    // FRI reduction plan (ConstantArityBits)
    std::vector<int> arities;
    {
        int d = L;
        while (d > cfg->final_poly_bits && d + r - cfg->arity_bits >= cap_h) {
            arities.push_back(cfg->arity_bits);
            d -= cfg->arity_bits;
        }
    }
    NEED(have(n_layers + 2) && n_layers == arities.size(), "FRI plan mismatch");
    for (size_t i = 0; i < n_layers; ++i) NEED(pr[pos + i] == (uint64_t)arities[i], "FRI plan mismatch");
    pos += n_layers;
    const size_t final_len = pr[pos], n_pub = pr[pos + 1];
    pos += 2;
    const int LN = L + r;
    const size_t n = (size_t)1 << L, N = (size_t)1 << LN;
    int final_log = LN;
    for (int a : arities) final_log -= a;
    NEED(pr[3] == (uint64_t)air->cols && n_pub == (size_t)air->pub && nq == 4 && final_len == (((size_t)1 << final_log) >> r), "shape mismatch");
    // the proof is untrusted input: the shapes the prover refuses (vx_stark_prove_impl) are refused here too, so no
    // Merkle depth below can go negative (a crafted L = 2 proof used to reach v_merkle with n_sib = SIZE_MAX)
    NEED(r >= 1 && r <= 3 && cap_h >= 0 && LN >= cap_h && LN <= 27 && L >= air->period_log, "degree bits %d out of range for this AIR / cap height", L);
    NEED(!air->exact_log || L == air->period_log, "this AIR has positional columns of period 2^%d: a trace of 2^%d rows is not acceptable", air->period_log, L);
    {
        int cur = LN;
        for (int a : arities) {
            NEED(cur - a - cap_h >= 0, "FRI layer below the cap height");
            cur -= a;
        }
    }
    for (size_t i = 0; i < len; ++i) NEED(pr[i] < glh::P || i < pos, "non-canonical element at word %zu", i);
    NEED(have(n_pub), "proof truncated (public inputs)");
    const uint64_t* pub = pr + pos;
    pos += n_pub;
    if (expect_public) {
        NEED(n_expect_public == n_pub, "public input count differs");
        for (size_t i = 0; i < n_pub; ++i) NEED(pub[i] == expect_public[i], "public input %zu differs", i);
    }
    const size_t cap_words = (size_t)4 << cap_h;
    NEED(have((ca ? 3 : 2) * cap_words + 2 * (size_t)air->auxpub + 2 * (2 * c + nq)), "proof truncated (caps/openings)");
    const uint64_t* cap_t = pr + pos;
    pos += cap_words;
    const uint64_t *apub = nullptr, *cap_a = nullptr;
    if (ca) {
        apub = pr + pos;
        cap_a = pr + pos + 2 * (size_t)air->auxpub;
        pos += 2 * (size_t)air->auxpub + cap_words;
    }
    const uint64_t* cap_q = pr + pos;
    pos += cap_words;
    std::vector<Fx> o_local(c), o_next(c), o_quot(nq);
    for (size_t j = 0; j < c; ++j) o_local[j] = {pr[pos + 2 * j], pr[pos + 2 * j + 1]};
    pos += 2 * c;
    for (size_t j = 0; j < c; ++j) o_next[j] = {pr[pos + 2 * j], pr[pos + 2 * j + 1]};
    pos += 2 * c;
    for (size_t j = 0; j < nq; ++j) o_quot[j] = {pr[pos + 2 * j], pr[pos + 2 * j + 1]};
    pos += 2 * nq;

    VChallenger ch;
    ch.observe(pub, n_pub);
    ch.observe(cap_t, cap_words);
    uint64_t chal[8] = {0};
    if (ca) {  // auxiliary round: lookup challenges after the trace cap, then the published values and the second cap
        NEED(air->chal <= 8 && 2 * air->auxpub <= 8, "AIR %d auxiliary round is misconfigured", air_id);
        if (ext_chal) {
            for (int q = 0; q < air->chal; ++q) chal[q] = ext_chal[q];
            ch.observe(chal, (size_t)air->chal);
        } else
            for (int q = 0; q < air->chal; ++q) chal[q] = ch.challenge();
        ch.observe(apub, 2 * (size_t)air->auxpub);
        ch.observe(cap_a, cap_words);
    }
    const uint64_t alphas[2] = {ch.challenge(), ch.challenge()};
    ch.observe(cap_q, cap_words);
    const Fx zeta = ch.ext();
    const uint64_t wn = glh::root(L), last = glh::inv(wn), ninv = glh::inv(n % glh::P);
    const Fx zeta_next = zeta * Fx{wn, 0};
    // ---- constraint identity at zeta
    {
        const Fx zn = fx_pow(zeta, n), one{1, 0};
        const Fx zh = zn - one;
        Consumer<Fx> cons;
        cons.acc[0] = cons.acc[1] = {0, 0};
        cons.alpha[0] = {alphas[0], 0};
        cons.alpha[1] = {alphas[1], 0};
        cons.z_last = zeta - Fx{last, 0};
        cons.l_first = zh * Fx{ninv, 0} * fx_inv(zeta - one);
        cons.l_last = zh * Fx{glh::mul(ninv, last), 0} * fx_inv(zeta - Fx{last, 0});
        std::vector<Fx> per(air->periodic ? air->periodic : 1), pubx(n_pub ? n_pub : 1), chalx(8), apubx(8);
        if (air->periodic) {
            std::vector<uint64_t> pv;
            if (air->prog) pv = air->prog->periodic;
            else air->periodic_values(pv);
            size_t in_off = 0;
            for (int j = 0; j < air->periodic; ++j) {
                const int pl = air->prog ? air->prog->plog[j] : air->plog(j);
                const size_t p = (size_t)1 << pl;
                NEED(in_off + p <= pv.size(), "periodic table of AIR %d is short", air_id);
                std::vector<uint64_t> coef(pv.begin() + in_off, pv.begin() + in_off + p);
                host_intt(coef);
                const Fx y = fx_pow(zeta, n >> pl);
                Fx a{0, 0};
                for (size_t k = p; k-- > 0;) a = a * y + Fx{coef[k], 0};
                per[j] = a;
                in_off += p;
            }
        }
        for (size_t i = 0; i < n_pub; ++i) pubx[i] = {pub[i], 0};
        for (int q = 0; q < air->chal; ++q) chalx[q] = {chal[q], 0};
        for (int q = 0; q < 2 * air->auxpub; ++q) apubx[q] = {apub[q], 0};
        HostRow loc{o_local.data()}, nxt{o_next.data()};
        if (air->prog) air_program_eval<Fx>(*air->prog, loc, nxt, per.data(), pubx.data(), chalx.data(), apubx.data(), cons);
        else air->eval(loc, nxt, per.data(), pubx.data(), chalx.data(), apubx.data(), cons);
        for (int k = 0; k < 2; ++k) {
            const Fx q = o_quot[2 * k] + o_quot[2 * k + 1] * zn;
            NEED(fx_eq(cons.acc[k], zh * q), "constraint identity fails at zeta (challenge %d)", k);
        }
    }
    for (size_t j = 0; j < c; ++j) ch.observe(o_local[j].a), ch.observe(o_local[j].b);
    for (size_t j = 0; j < nq; ++j) ch.observe(o_quot[j].a), ch.observe(o_quot[j].b);
    for (size_t j = 0; j < c; ++j) ch.observe(o_next[j].a), ch.observe(o_next[j].b);
    const Fx alpha = ch.ext();
    std::vector<const uint64_t*> layer_caps;
    std::vector<Fx> betas;
    for (size_t l = 0; l < n_layers; ++l) {
        NEED(have(cap_words), "proof truncated (FRI caps)");
        layer_caps.push_back(pr + pos);
        ch.observe(pr + pos, cap_words);
        pos += cap_words;
        betas.push_back(ch.ext());
    }
    NEED(have(2 * final_len + 1), "proof truncated (final poly)");
    std::vector<Fx> fpoly(final_len);
    for (size_t k = 0; k < final_len; ++k) {
        fpoly[k] = {pr[pos + 2 * k], pr[pos + 2 * k + 1]};
        ch.observe(fpoly[k].a), ch.observe(fpoly[k].b);
    }
    pos += 2 * final_len;
    const uint64_t nonce = pr[pos++];
    ch.observe(nonce);
    const uint64_t resp = ch.challenge();
    NEED(pow_bits == 0 || (resp >> (64 - pow_bits)) == 0, "proof of work invalid");
    // reduced openings
    Fx apow{1, 0}, y0{0, 0}, y1{0, 0};
    for (size_t j = 0; j < c + nq; ++j) {
        if (j < c) {
            y0 = y0 + apow * o_local[j];
            y1 = y1 + apow * o_next[j];
        } else y0 = y0 + apow * o_quot[j - c];
        apow = apow * alpha;
    }
    const Fx alpha_c = fx_pow(alpha, c);
    const int depth0 = LN - cap_h;
    const uint64_t wN = glh::root(LN);
    for (size_t qi = 0; qi < n_queries; ++qi) {
        size_t x_index = ch.challenge() % N;
        NEED(have(c + nq + (ca ? 12 : 8) * (size_t)depth0), "proof truncated (query %zu)", qi);
        const uint64_t* row_t = pr + pos;
        const uint64_t* sib_t = row_t + cm;
        const uint64_t* row_a = sib_t + 4 * depth0;
        const uint64_t* sib_a = row_a + ca;
        const uint64_t* row_q = ca ? sib_a + 4 * depth0 : row_a;
        const uint64_t* sib_q = row_q + nq;
        pos += c + nq + (ca ? 12 : 8) * (size_t)depth0;
        NEED(v_merkle(row_t, cm, x_index, sib_t, depth0, cap_t), "trace Merkle proof invalid (query %zu)", qi);
        if (ca) NEED(v_merkle(row_a, ca, x_index, sib_a, depth0, cap_a), "auxiliary Merkle proof invalid (query %zu)", qi);
        NEED(v_merkle(row_q, nq, x_index, sib_q, depth0, cap_q), "quotient Merkle proof invalid (query %zu)", qi);
        uint64_t x = glh::mul(7, glh::pow(wN, brev(x_index, LN)));
        Fx s1{0, 0}, ap{1, 0};
        for (size_t j = 0; j < c; ++j) {
            s1 = s1 + ap * Fx{j < cm ? row_t[j] : row_a[j - cm], 0};
            ap = ap * alpha;
        }
        Fx s0 = s1;
        for (size_t j = 0; j < nq; ++j) {
            s0 = s0 + ap * Fx{row_q[j], 0};
            ap = ap * alpha;
        }
        Fx ev = alpha_c * (s0 - y0) * fx_inv(Fx{x, 0} - zeta) + (s1 - y1) * fx_inv(Fx{x, 0} - zeta_next);
        int cur_log = LN;
        for (size_t l = 0; l < n_layers; ++l) {
            const int a = arities[l];
            const size_t arity = (size_t)1 << a, within = x_index & (arity - 1);
            const int depth = cur_log - a - cap_h;
            NEED(have(2 * (arity - 1) + 4 * (size_t)depth), "proof truncated (query %zu layer %zu)", qi, l);
            std::vector<uint64_t> leaf(2 * arity);
            for (size_t t = 0, src = 0; t < arity; ++t) {
                if (t == within) {
                    leaf[2 * t] = ev.a, leaf[2 * t + 1] = ev.b;
                } else {
                    leaf[2 * t] = pr[pos + 2 * src], leaf[2 * t + 1] = pr[pos + 2 * src + 1];
                    ++src;
                }
            }
            pos += 2 * (arity - 1);
            NEED(v_merkle(leaf.data(), 2 * arity, x_index >> a, pr + pos, depth, layer_caps[l]), "FRI layer %zu Merkle proof invalid (query %zu)", l, qi);
            pos += 4 * (size_t)depth;
            // compute_evaluation: interpolate the coset {x g^i} and evaluate at beta
            const uint64_t g = glh::root(a);
            std::vector<Fx> evn(arity);
            for (size_t t = 0; t < arity; ++t) evn[brev(t, a)] = {leaf[2 * t], leaf[2 * t + 1]};
            const uint64_t start = glh::mul(x, glh::pow(g, arity - brev(within, a)));
            std::vector<uint64_t> pts(arity);
            uint64_t gp = 1;
            for (size_t t = 0; t < arity; ++t) {
                pts[t] = glh::mul(start, gp);
                gp = glh::mul(gp, g);
            }
            Fx acc{0, 0};
            for (size_t i = 0; i < arity; ++i) {
                Fx num{1, 0};
                uint64_t den = 1;
                for (size_t j = 0; j < arity; ++j) {
                    if (j == i) continue;
                    num = num * (betas[l] - Fx{pts[j], 0});
                    den = glh::mul(den, glh::sub(pts[i], pts[j]));
                }
                acc = acc + evn[i] * num * Fx{glh::inv(den), 0};
            }
            ev = acc;
            x = glh::pow(x, arity);
            x_index >>= a;
            cur_log -= a;
        }
        Fx fp{0, 0};
        for (size_t k = final_len; k-- > 0;) fp = fp * Fx{x, 0} + fpoly[k];
        NEED(fx_eq(fp, ev), "final polynomial evaluation mismatch (query %zu)", qi);
    }
    NEED(pos == len, "trailing data in proof (%zu of %zu words used)", pos, len);
    if (ca && !ext_chal)  // a stand-alone proof has nobody to cancel a bus total against
        for (int q = 0; q < 2 * air->auxpub; ++q) NEED(apub[q] == 0, "stand-alone proof publishes a non-zero bus total");
    if (apub_out) *apub_out = apub;
    if (log_n_out) *log_n_out = L;
    return VX_OK;
}

// host-side shared challenges for verifiers outside this file
void vx_shared_challenges_host(const uint64_t* const* pubs, const size_t* n_pubs, const uint64_t* const* caps, size_t k, size_t cap_words, uint64_t* out, size_t n_out) {
    v_shared_challenges_n(pubs, n_pubs, caps, k, cap_words, out, n_out);
}

// Expected public inputs and AIR ids of the three justification tables (vx_bus.h).
int32_t vx_justification_expect(const uint64_t* ppub_chain, size_t n_chain, const uint64_t* ppub_ed, size_t n_ed, size_t n_s512, const uint8_t authority_set_hash[32],
                                uint64_t authority_set_id, const uint8_t block_hash[32], uint32_t block_number, uint64_t round, uint64_t spub[10], uint64_t epub[2],
                                uint64_t hpub[15], int air[3], char* err, size_t errlen) {
    NEED(n_chain == 10 && n_ed == 2 && n_s512 == 15, "justification proofs have the wrong number of public inputs");
    // the request's authority_set_hash must be the proven commitment; the number of authorities it binds and the number of verified
    // signatures are read from the proofs: signed * 3 > authorities * 2 (justification.rs:164-186)
    for (int j = 0; j < 8; ++j)
        spub[j] = ((uint64_t)authority_set_hash[4 * j] << 24) | ((uint64_t)authority_set_hash[4 * j + 1] << 16) | ((uint64_t)authority_set_hash[4 * j + 2] << 8) |
                  authority_set_hash[4 * j + 3];
    const uint64_t n_auth = ppub_chain[8], n_signed = ppub_ed[0];
    NEED(n_auth >= 1 && n_auth <= 512 && n_signed <= n_auth, "implausible authority counts");
    NEED(n_signed * 3 > n_auth * 2, "fewer than 2/3 of the authority set signed (%llu of %llu)", (unsigned long long)n_signed, (unsigned long long)n_auth);
    spub[8] = n_auth, spub[9] = 1;
    epub[0] = n_signed, epub[1] = 1;
    air[0] = VX_AIR_SHA_CHAIN;
    air[1] = n_signed <= 255 ? VX_AIR_ED25519_16 : VX_AIR_ED25519;  // the tables are sized by the number of signatures they verify
    air[2] = n_signed <= 6 ? VX_AIR_SHA512_10 : n_signed <= 204 ? VX_AIR_SHA512_15 : VX_AIR_SHA512;
    // the signed message: the precommit for (block hash, block number, round, set id) -- decoder.rs:159-200
    uint8_t msg[64];
    memset(msg, 0, sizeof msg);
    msg[0] = 1;
    memcpy(msg + 1, block_hash, 32);
    for (int b = 0; b < 4; ++b) msg[33 + b] = (uint8_t)(block_number >> (8 * b));
    for (int b = 0; b < 8; ++b) msg[37 + b] = (uint8_t)(round >> (8 * b)), msg[45 + b] = (uint8_t)(authority_set_id >> (8 * b));
    msg[53] = 0x80;
    for (int j = 0; j < 7; ++j) {
        uint64_t v = 0;
        for (int b = 0; b < 8; ++b) v = (v << 8) | msg[8 * j + b];
        hpub[2 * j] = v & 0xFFFFFFFFULL, hpub[2 * j + 1] = v >> 32;
    }
    hpub[14] = 1;
    return VX_OK;
}

extern "C" {

int32_t vx_header_range_verify(const vx_stark_config* cfg, const uint64_t* blob, size_t len, uint32_t max_headers,
                               uint32_t trusted_block, const uint8_t trusted_hash[32], uint64_t authority_set_id, const uint8_t* authority_set_hash,
                               uint32_t target_block, const uint8_t out96[96], char* err, size_t errlen) {
    if (!cfg || !blob || !trusted_hash || !out96) return VX_ERR_ARG;
    NEED(len > VX_HR_BLOB_FIXED_WORDS + 1 && blob[0] == VX_HR_BLOB_MAGIC, "bad header_range blob");
    NEED(blob[1] == max_headers && blob[2] == trusted_block && blob[3] == target_block, "blob is for a different request");
    NEED(memcmp(blob + 4, out96, 96) == 0, "public outputs differ from the blob");
    NEED(target_block > trusted_block, "empty block range");
    const uint64_t S = blob[16];
    NEED(S >= 1 && S <= VX_HR_MAX_SEGMENTS && S <= (uint64_t)target_block - trusted_block, "bad number of map segments");
    const size_t HDR = VX_HR_BLOB_FIXED_WORDS + (size_t)S;
    NEED(len > HDR, "bad header_range blob");
    // proofs in blob order: the S hash-chain segments, authority-set commitment, Merkle, Ed25519, SHA-512
    const size_t NB = (size_t)S + 4;
    std::vector<size_t> plen(NB), off(NB);
    size_t tot = HDR;
    for (size_t t = 0; t < NB; ++t) {
        plen[t] = t < S ? blob[VX_HR_BLOB_FIXED_WORDS + t] : blob[17 + (t - S)];
        NEED(plen[t] <= len, "blob lengths are inconsistent");
        off[t] = tot, tot += plen[t];
    }
    NEED(tot == len, "blob lengths are inconsistent");
    for (size_t s = 0; s < S; ++s) NEED(plen[s] > 0, "a map segment of the hash-chain table is missing");
    NEED(plen[S + 1] > 0, "the Merkle table is missing");
    const bool justified = plen[S] > 0;
    NEED(justified ? (plen[S + 2] > 0 && plen[S + 3] > 0) : (plen[S + 2] == 0 && plen[S + 3] == 0), "blob carries part of a justification");
    NEED(!authority_set_hash || justified, "blob carries no authority-set commitment proof");
    NEED(!justified || authority_set_hash, "blob carries a justification: the request's authority_set_hash is needed to check it");
    const int tree_id = max_headers == 256 ? 7 : max_headers == 512 ? 8 : max_headers == 16 ? 9 : 0;
    NEED(tree_id, "max_headers %u has no Merkle AIR", max_headers);
    // bus order (the order of the shared-challenge transcript): segments, Merkle, commitment, Ed25519, SHA-512
    const size_t n_tab = (size_t)S + (justified ? 4 : 1);
    std::vector<const uint64_t*> proof(n_tab), ppub(n_tab), pcap(n_tab);
    std::vector<size_t> pl(n_tab), npub(n_tab);
    for (size_t t = 0; t < n_tab; ++t) {
        const size_t b = t < S ? t : (t == S ? S + 1 : (t == S + 1 ? S : t));  // bus index -> blob index
        proof[t] = blob + off[b], pl[t] = plen[b];
        NEED(vx_stark_proof_peek(proof[t], pl[t], cfg->cap_height, &ppub[t], &npub[t], &pcap[t]), "proofs are too short to hold a trace cap");
    }
    // public inputs of every table, rebuilt from the request and the claimed outputs.  The hash a segment ends with is read from
    // its own proof and must be the hash the next segment starts from (the reference's reduce step, subchain_verification.rs:
    // 247-257): the first starts at trusted_header_hash / trusted_block + 1, the last ends at target_header_hash / target_block,
    // block numbers run on from segment to segment, and every segment counts its Merkle leaves from trusted_block + 1
    std::vector<uint64_t> spubs(20 * (size_t)S);
    uint64_t tpub[17], cpub[10], epub[2], hpub[15];
    uint64_t next_first = (uint64_t)trusted_block + 1;
    for (size_t s = 0; s < S; ++s) {
        uint64_t* pub = spubs.data() + 20 * s;
        NEED(npub[s] == 20, "a hash-chain segment has %zu public inputs", npub[s]);
        for (int j = 0; j < 8; ++j) {
            uint32_t a;
            memcpy(&a, trusted_hash + 4 * j, 4);
            pub[j] = s == 0 ? a : ppub[s - 1][8 + j];                        // starts where the previous segment ended
            uint32_t b;
            memcpy(&b, out96 + 4 * j, 4);                                     // target_header_hash = first 32 output bytes
            pub[8 + j] = s + 1 == S ? b : ppub[s][8 + j];                     // an inner boundary: the segment's own claim, bound by the next one
            NEED(pub[8 + j] >> 32 == 0, "a segment hash limb is out of range");
        }
        const uint64_t last = s + 1 == S ? target_block : ppub[s][17];
        NEED(last >= next_first && last <= target_block, "the block numbers of the map segments do not run on");
        pub[16] = next_first, pub[17] = last;
        pub[18] = (uint64_t)trusted_block + 1;  // the block of Merkle leaf 0
        pub[19] = 1;                            // bus on
        next_first = last + 1;
    }
    NEED(next_first == (uint64_t)target_block + 1, "the map segments do not cover the block range");
    for (int j = 0; j < 16; ++j)  // state_root_merkle_root || data_root_merkle_root as big-endian words
        tpub[j] = ((uint64_t)out96[32 + 4 * j] << 24) | ((uint64_t)out96[33 + 4 * j] << 16) | ((uint64_t)out96[34 + 4 * j] << 8) | out96[35 + 4 * j];
    tpub[16] = (uint64_t)target_block - trusted_block;  // the number of headers = of enabled leaves: the Merkle table MUST take every header's roots from the bus
    NEED(tpub[16] <= max_headers, "the block range exceeds max_headers");
    std::vector<int> air(n_tab, VX_AIR_BLAKE_CHAIN);
    std::vector<const uint64_t*> want(n_tab);
    std::vector<size_t> n_want(n_tab, 20);
    for (size_t s = 0; s < S; ++s) want[s] = spubs.data() + 20 * s;
    air[S] = tree_id, want[S] = tpub, n_want[S] = 17;
    if (justified) {
        int jair[3];
        const int32_t rc = vx_justification_expect(ppub[S + 1], npub[S + 1], ppub[S + 2], npub[S + 2], npub[S + 3], authority_set_hash, authority_set_id, out96, target_block, blob[21], cpub, epub,
                                                   hpub, jair, err, errlen);
        if (rc != VX_OK) return rc;
        air[S + 1] = jair[0], air[S + 2] = jair[1], air[S + 3] = jair[2];
        want[S + 1] = cpub, want[S + 2] = epub, want[S + 3] = hpub;
        n_want[S + 1] = 10, n_want[S + 2] = 2, n_want[S + 3] = 15;
    }
    // the lookup challenges every proof must have used: a transcript of all public inputs and trace caps
    uint64_t chal[4];
    v_shared_challenges_n(ppub.data(), npub.data(), pcap.data(), n_tab, (size_t)4 << cfg->cap_height, chal, 4);
    uint64_t bus[2] = {0, 0};
    for (size_t t = 0; t < n_tab; ++t) {
        const uint64_t* apub = nullptr;
        int L = 0;
        const int32_t rc = vx_stark_verify_ext(cfg, proof[t], pl[t], air[t], want[t], n_want[t], chal, &apub, &L, err, errlen);
        if (rc != VX_OK) return rc;
        for (int q = 0; q < 2; ++q) bus[q] = glh::add(bus[q], glh::mul(apub[q], ((uint64_t)1 << L) % glh::P));  // a table publishes its total / rows
    }
    // the bus closes: state roots and data roots of the hashed headers = the leaves of the Merkle trees; the keys of the signed
    // authorities = the keys the signatures verify under; R || A and H between the curve table and the SHA-512 table
    NEED(bus[0] == 0 && bus[1] == 0, "the lookup bus between the tables does not balance");
    return VX_OK;
}

// RotateCircuit verify (the prover is vx_rotate.hip; every host verifier lives in this file, which holds no GPU code): the blob must
// be for this (authority_set_id, authority_set_hash) request and claim out32; then its six STARKs are verified in their two
// shared-challenge groups against the public inputs those values imply, and both buses must balance.
int32_t vx_rotate_verify(const vx_stark_config* cfg, const uint64_t* blob, size_t len, uint64_t authority_set_id,
                         const uint8_t authority_set_hash[32], const uint8_t out32[32], char* err, size_t errlen) {
    if (!cfg || !blob || !authority_set_hash || !out32) return VX_ERR_ARG;
    auto bad = [&](const char* why) {
        if (err && errlen) snprintf(err, errlen, "%s", why);
        return (int32_t)VX_ERR_STATEMENT;
    };
    if (len <= VX_ROT_HDR || blob[0] != VX_ROT_MAGIC) return bad("bad rotate blob");
    if (blob[1] != authority_set_id || memcmp(blob + 8, authority_set_hash, 32) != 0) return bad("blob is for a different request");
    if (memcmp(blob + 12, out32, 32) != 0) return bad("public output differs from the blob");
    const size_t l0 = blob[16], l1 = blob[17], l2 = blob[18], l3 = blob[19], l4 = blob[24], l5 = blob[27];
    if (l0 > len || l1 > len || l2 > len || l3 > len || l4 > len || l5 > len || VX_ROT_HDR + l0 + l1 + l2 + l3 + l4 + l5 != len) return bad("blob lengths are inconsistent");
    if (blob[2] >> 32 || blob[26] >= VX_MAX_HEADER_SIZE || blob[3] == 0 || blob[3] > 510) return bad("block number, start position or authority count out of range");
    const uint64_t* p0 = blob + VX_ROT_HDR;
    int32_t rc;
    // Bus B: the Blake2b table (a chain of exactly one header, numbered epoch_end_block, hashing to the blob's header hash; the anchor
    // is the parent hash the header itself carries -- free in this statement) sends the bytes from start_position + 1 on; the
    // epoch-end table reads the ScheduledChange log of blob[3] authorities there and sends its keys; the new set's commitment table
    // receives every key and hashes to out32.
    {
        const uint64_t* proof[3] = {p0, p0 + l0 + l1 + l2 + l3 + l4, p0 + l0 + l1};
        const size_t pl[3] = {l0, l5, l2};
        const int air[3] = {VX_AIR_BLAKE_CHAIN, VX_AIR_EPOCH_END, VX_AIR_SHA_CHAIN};
        const uint64_t *ppub[3], *pcap[3];
        size_t npub[3];
        for (int t = 0; t < 3; ++t)
            if (!vx_stark_proof_peek(proof[t], pl[t], cfg->cap_height, &ppub[t], &npub[t], &pcap[t])) return bad("epoch-end proofs are too short to hold a trace cap");
        uint64_t bpub[20], epub[10], spub[10];
        for (int j = 0; j < 8; ++j) {
            uint32_t a, b;
            memcpy(&a, (const uint8_t*)(blob + 20) + 4 * j, 4);
            memcpy(&b, (const uint8_t*)(blob + 4) + 4 * j, 4);
            bpub[j] = a;
            bpub[8 + j] = b;
        }
        bpub[16] = bpub[17] = blob[2];
        bpub[18] = blob[26] + 1, bpub[19] = 2;  // window mode: the bytes behind start_position
        // the byte lengths of the log's two compact ints are the prover's to state (one-hot); the table's constraints tie them to the bytes
        if (npub[1] != 10) return bad("epoch-end proof is malformed");
        epub[0] = blob[3], epub[1] = 1;
        for (int g = 0; g < 2; ++g) {
            uint64_t sum = 0;
            for (int a = 0; a < 4; ++a) {
                const uint64_t f = ppub[1][2 + 4 * g + a];
                if (f > 1) return bad("epoch-end proof: length flags are not one-hot");
                epub[2 + 4 * g + a] = f, sum += f;
            }
            if (sum != 1) return bad("epoch-end proof: length flags are not one-hot");
        }
        be_limbs(out32, spub);
        spub[8] = blob[3], spub[9] = 2;  // receives every key
        uint64_t chal[4];
        vx_shared_challenges_host(ppub, npub, pcap, 3, (size_t)4 << cfg->cap_height, chal, 4);
        const uint64_t* want[3] = {bpub, epub, spub};
        const size_t n_want[3] = {20, 10, 10};
        uint64_t bus[2] = {0, 0};
        for (int t = 0; t < 3; ++t) {
            const uint64_t* apub = nullptr;
            int L = 0;
            rc = vx_stark_verify_ext(cfg, proof[t], pl[t], air[t], want[t], n_want[t], chal, &apub, &L, err, errlen);
            if (rc != VX_OK) return rc;
            for (int q = 0; q < 2; ++q) bus[q] = glh::add(bus[q], glh::mul(apub[q], ((uint64_t)1 << L) % glh::P));
        }
        if (bus[0] || bus[1]) return bad("the lookup bus between the header hash, the epoch-end table and the new set does not balance");
    }
    // the justification by the current set: commitment, Ed25519 and SHA-512 tables under shared lookup challenges; the signed
    // message is the precommit for (the proven header hash, the block number, the round, the request's set id)
    {
        const uint64_t* proof[3] = {p0 + l0, p0 + l0 + l1 + l2, p0 + l0 + l1 + l2 + l3};
        const size_t pl[3] = {l1, l3, l4};
        const uint64_t *ppub[3], *pcap[3];
        size_t npub[3];
        for (int t = 0; t < 3; ++t)
            if (!vx_stark_proof_peek(proof[t], pl[t], cfg->cap_height, &ppub[t], &npub[t], &pcap[t])) return bad("justification proofs are too short to hold a trace cap");
        uint64_t spub[10], epub[2], hpub[15];
        int air[3];
        rc = vx_justification_expect(ppub[0], npub[0], ppub[1], npub[1], npub[2], authority_set_hash, authority_set_id, (const uint8_t*)(blob + 4), (uint32_t)blob[2], blob[25],
                                     spub, epub, hpub, air, err, errlen);
        if (rc != VX_OK) return rc;
        uint64_t chal[4];
        vx_shared_challenges_host(ppub, npub, pcap, 3, (size_t)4 << cfg->cap_height, chal, 4);
        const uint64_t* want[3] = {spub, epub, hpub};
        const size_t n_want[3] = {10, 2, 15};
        uint64_t bus[2] = {0, 0};
        for (int t = 0; t < 3; ++t) {
            const uint64_t* apub = nullptr;
            int L = 0;
            rc = vx_stark_verify_ext(cfg, proof[t], pl[t], air[t], want[t], n_want[t], chal, &apub, &L, err, errlen);
            if (rc != VX_OK) return rc;
            for (int q = 0; q < 2; ++q) bus[q] = glh::add(bus[q], glh::mul(apub[q], ((uint64_t)1 << L) % glh::P));
        }
        if (bus[0] || bus[1]) return bad("the lookup bus between the justification tables does not balance");
    }
    return VX_OK;
}

}
