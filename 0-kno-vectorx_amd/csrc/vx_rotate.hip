// RotateCircuit (circuits/rotate.rs:80-109; builder/rotate.rs:74-323): the epoch-end header is hashed
// (Blake2b STARK over its compressions), justified by > 2/3 of the CURRENT authority set (Ed25519 batch on
// the GPU + authority-set commitment STARK), checked to carry the ScheduledChange log that encodes the NEW
// authority set (k_epoch_end_check, one lane per validator, names the failing rule; EpochEndAir proves it), and the
// new set's commitment is proved (second SHA-256 chain STARK) and returned as the 32 output bytes.  Two logUp buses:
//   A  current-set commitment -> Ed25519 <-> SHA-512                 (the justification, vx_bus.h)
//   B  Blake2b header hash -> EpochEndAir -> new-set commitment     (the header bytes of the log ARE the committed keys)
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "vx_bus.h"
#include "vx_internal.h"

namespace {

constexpr uint32_t MAX_HEADER_SIZE = 35840;  // consts.rs:16
constexpr uint32_t VALIDATOR_LENGTH = 40, PUBKEY_LENGTH = 32, DELAY_LENGTH = 4, MAX_PREFIX_LENGTH = 17;

// failure codes written by k_epoch_end_check (smallest wins); low 16 bits carry the validator index
enum : uint32_t { EE_OK = 0xffffffffu, EE_RANGE = 1, EE_FLAG, EE_ENGINE, EE_COMPACT, EE_SCHED, EE_COUNT, EE_PUBKEY, EE_WEIGHT, EE_DELAY };

__device__ bool ee_compact(const uint8_t* b, uint32_t* val, uint32_t* len) {  // decoder.rs:39-103
    const uint32_t m = b[0] & 3;
    bool ok = true;
    switch (m) {
        case 0: *val = b[0] >> 2; *len = 1; break;
        case 1: *val = ((uint32_t)b[0] | ((uint32_t)b[1] << 8)) >> 2; *len = 2; break;
        case 2: *val = ((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24)) >> 2; *len = 4; break;
        default:
            *val = (uint32_t)b[1] | ((uint32_t)b[2] << 8) | ((uint32_t)b[3] << 16) | ((uint32_t)b[4] << 24);
            *len = 5;
            ok = (b[0] >> 2) == 0;
    }
    return ok;
}

// lane t < max_authorities: validator t (pubkey, weight; delay when t is the last one).  Every lane parses
// the 17-byte prefix itself (rotate.rs:74-174) -- cheaper than a second launch.
__global__ __launch_bounds__(64) void k_epoch_end_check(const uint8_t* __restrict__ header, uint32_t num_authorities, uint32_t start_position,
                                                        const uint8_t* __restrict__ new_pubkeys, uint32_t max_authorities,
                                                        uint32_t* __restrict__ result) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= max_authorities) return;
    auto fail = [&](uint32_t code, uint32_t idx) { atomicMin(result, (code << 16) | (idx & 0xffff)); };
    if ((uint64_t)start_position + MAX_PREFIX_LENGTH > MAX_HEADER_SIZE) return fail(EE_RANGE, 0);
    const uint8_t* p = header + start_position;
    if (p[1] != 4) return fail(EE_FLAG, 0);                                                // :84-86
    if (p[2] != 70 || p[3] != 82 || p[4] != 78 || p[5] != 75) return fail(EE_ENGINE, 0);  // :90-95 "FRNK"
    uint32_t val, len;
    if (!ee_compact(p + 6, &val, &len)) return fail(EE_COMPACT, 0);                        // :113-121, value unused
    uint32_t cursor = 6 + len;
    if (p[cursor] != 1) return fail(EE_SCHED, 0);                                          // :133-137
    ++cursor;                                                                              // <= 12, so cursor + 5 <= 17 always
    if (!ee_compact(p + cursor, &val, &len)) return fail(EE_COMPACT, 1);
    if (val != num_authorities) return fail(EE_COUNT, 0);                                  // :161-165
    const uint64_t base = (uint64_t)start_position + cursor + len;                         // :224
    if (base + (uint64_t)max_authorities * VALIDATOR_LENGTH + DELAY_LENGTH > MAX_HEADER_SIZE) return fail(EE_RANGE, 1);  // :236-240
    if (t >= num_authorities) return;  // validator_disabled
    const uint8_t* v = header + base + (uint64_t)t * VALIDATOR_LENGTH;
    const uint8_t* pk = new_pubkeys + (size_t)t * PUBKEY_LENGTH;
    bool same = true;
    for (int j = 0; j < 32; ++j) same &= v[j] == pk[j];
    if (!same) return fail(EE_PUBKEY, t);                                                  // :251-255
    bool w = v[32] == 1;
    for (int j = 33; j < 40; ++j) w &= v[j] == 0;
    if (!w) return fail(EE_WEIGHT, t);                                                     // :259-265
    if (t + 1 == num_authorities && (v[40] | v[41] | v[42] | v[43]) != 0) return fail(EE_DELAY, t);  // :267-274
}

int sha_rows_log(size_t n_keys) {
    int log_n = 6;
    while (((size_t)1 << log_n) < 64 * (2 * n_keys - 1)) ++log_n;
    return log_n;
}
int blake_rows_log(size_t chunks) {
    int log_n = 16;  // one copy of the 2^16-row XOR tables (BlakeChainAir)
    while (((size_t)1 << log_n) < 16 * chunks) ++log_n;
    return log_n;
}


}  // namespace

extern "C" {

int32_t vx_verify_epoch_end_header(vx_ctx* ctx, const vx_buf* header, uint32_t num_authorities, uint32_t start_position,
                                   const uint8_t* new_pubkeys, uint32_t max_authorities) {
    if (!ctx || !header || !new_pubkeys) return VX_ERR_ARG;
    VX_CHECK(header->n * 8 >= MAX_HEADER_SIZE, "rotate: header buffer holds %zu bytes, need MAX_HEADER_SIZE = %u", header->n * 8, MAX_HEADER_SIZE);
    VX_CHECK(max_authorities >= 1 && max_authorities <= 4096, "rotate: max_authorities %u out of range", max_authorities);
    if (num_authorities == 0) return vx_fail(ctx, VX_ERR_STATEMENT, "rotate: no authorities");                       // rotate.rs:190-192
    if (num_authorities > max_authorities)                                                                            // circuits/rotate.rs:43-45
        return vx_fail(ctx, VX_ERR_STATEMENT, "rotate: %u authorities exceed the maximum authority set size %u", num_authorities, max_authorities);
    const size_t key_words = ((size_t)num_authorities * 32 + 7) / 8;
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, key_words + 1, &sc));
    uint32_t init = EE_OK;
    VX_HIP(hipMemcpyAsync(sc, new_pubkeys, (size_t)num_authorities * 32, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemcpyAsync(sc + key_words, &init, 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_epoch_end_check, dim3((max_authorities + 63) / 64), dim3(64), 0, ctx->stream, (const uint8_t*)header->d,
                       num_authorities, start_position, (const uint8_t*)sc, max_authorities, (uint32_t*)(sc + key_words));
    VX_HIP(hipGetLastError());
    uint32_t res = 0;
    VX_HIP(hipMemcpyAsync(&res, sc + key_words, 4, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    if (res == EE_OK) return VX_OK;
    static const char* WHY[] = {"", "subarray out of the header buffer", "consensus flag", "engine id", "compact int", "scheduled change flag",
                                "authority count", "pubkey", "weight", "delay"};
    const uint32_t code = res >> 16, idx = res & 0xffff;
    return vx_fail(ctx, VX_ERR_STATEMENT, "rotate: epoch end header rejected: %s (%u)", code < 10 ? WHY[code] : "?", idx);
}

int32_t vx_rotate_proof_bound(const vx_stark_config* cfg, size_t n_chunks, size_t n_cur_authorities, size_t n_new_authorities, size_t* n_words) {
    if (!cfg || !n_words || n_chunks == 0 || n_cur_authorities == 0 || n_new_authorities == 0) return VX_ERR_ARG;
    size_t w1 = 0, w2 = 0, w3 = 0, w4 = 0;
    int32_t rc = vx_stark_proof_bound(VX_AIR_BLAKE_CHAIN, cfg, blake_rows_log(n_chunks), &w1);
    if (rc == VX_OK) rc = vx_stark_proof_bound(VX_AIR_EPOCH_END, cfg, VX_EPOCH_END_LOG_ROWS, &w4);
    if (rc == VX_OK) w2 = vx_justification_proof_bound(cfg, n_cur_authorities, &rc);  // commitment of the current set + Ed25519 + SHA-512
    if (rc == VX_OK) rc = vx_stark_proof_bound(VX_AIR_SHA_CHAIN, cfg, sha_rows_log(n_new_authorities), &w3);
    *n_words = VX_ROT_HDR + w1 + w2 + w3 + w4;
    return rc;
}

int32_t vx_rotate_prove(vx_ctx* ctx, const vx_buf* header, uint32_t header_size, uint32_t epoch_end_block_number, uint32_t num_authorities,
                        uint32_t start_position, const uint8_t* new_pubkeys, const vx_justification* just, const vx_stark_config* cfg,
                        uint8_t out32[32], uint64_t* proof_out, size_t proof_cap, size_t* proof_len) {
    if (!ctx || !header || !new_pubkeys || !just || !cfg || !out32 || !proof_len) return VX_ERR_ARG;
    VX_CHECK(header->n * 8 >= MAX_HEADER_SIZE, "rotate: header buffer holds %zu bytes, need MAX_HEADER_SIZE = %u", header->n * 8, MAX_HEADER_SIZE);
    if (header_size > MAX_HEADER_SIZE)  // input/mod.rs:851-856
        return vx_fail(ctx, VX_ERR_STATEMENT, "rotate: header size %u is greater than MAX_HEADER_SIZE %u", header_size, MAX_HEADER_SIZE);
    VX_CHECK(header_size >= 36, "rotate: header of %u bytes cannot hold a parent hash and a block number", header_size);
    VX_CHECK(just->num_authorities >= 1 && just->num_authorities <= 512, "rotate: %u authorities (the EdDSA table holds 512)", just->num_authorities);
    // 0. the header's own rules, natively first (they name the error, and nothing is proven for a header that breaks them)
    uint8_t head[40], header_hash[32], new_commit[32];
    const uint8_t* parent = head;
    VX_HIP(hipMemcpyAsync(head, header->d, 40, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    {
        // The hash AIR pins the header's own SCALE number (4-byte compact at bytes 32..36) to its public input.  The
        // reference binds number and hash only through the signed precommit; for any header GRANDPA finalised the
        // two coincide, so a header numbered differently from epoch_end_block_number is refused here.
        const uint32_t enc = (epoch_end_block_number << 2) | 2;
        uint32_t got;
        memcpy(&got, head + 32, 4);
        if (got != enc)
            return vx_fail(ctx, VX_ERR_STATEMENT, "rotate: header does not carry block number %u as a 4-byte SCALE compact", epoch_end_block_number);
    }
    //    the header encodes the new authority set (rotate.rs:306-312)
    VX_TRY(vx_verify_epoch_end_header(ctx, header, num_authorities, start_position, new_pubkeys, just->max_authorities));
    // 1. the tables.  Bus A: the justification by the CURRENT set (rotate.rs:297-302) -- its commitment (binds the EVM input hash;
    //    sends the chosen signers' keys), the Ed25519 table and the SHA-512 table (vx_bus.h).  Bus B: the Blake2b table of the header
    //    hash (this context) sends the bytes of the ScheduledChange log to the epoch-end table, which sends the keys it reads there to
    //    the NEW set's commitment table.  Every table but the first is proven on a side context from a host thread.
    BusMeet rv, rvb;
    rv.n_parties = rvb.n_parties = 3;
    BusParty pb[3] = {{&rvb, 0}, {&rvb, 1}, {&rvb, 2}};
    const vx_chal_hook hb[3] = {{vx_bus_hook, &pb[0]}, {vx_bus_hook, &pb[1]}, {vx_bus_hook, &pb[2]}};
    JustificationTables jt;
    TableJob epoch, newset;
    vx_ctx* side[5];
    {
        vx_ctx* c = ctx;
        for (int t = 0; t < 5; ++t) side[t] = c = c ? vx_side_ctx(c) : nullptr;
        VX_CHECK(c, "rotate: no side context for every table");
    }
    epoch.c = side[3], newset.c = side[4];
    // the epoch-end trace (512 rows) is written from this thread: its prefix gives the byte window the Blake2b table sends
    vx_buf* et = nullptr;
    uint64_t epub[10];
    uint32_t wlen = 0;
    VX_TRY(vx_alloc(epoch.c, (size_t)VX_EPOCH_END_AIR_COLS << VX_EPOCH_END_LOG_ROWS, &et));
    struct FreeEt {
        vx_ctx* c;
        vx_buf*& b;
        ~FreeEt() {
            if (b) (void)vx_free(c, b);
        }
    } free_et{epoch.c, et};
    {
        const int32_t r = vx_epoch_end_trace_dev(epoch.c, (const uint8_t*)header->d, header->n * 8, start_position, num_authorities, 1, et->d, epub, &wlen);
        if (r != VX_OK) return vx_fail(ctx, r, "rotate: %s", vx_last_error(epoch.c));
    }
    if (start_position + 1 < 72 || (uint64_t)start_position + 1 + wlen > header_size)  // (the hash covers header_size bytes; 72 = parent hash + number + state root)
        return vx_fail(ctx, VX_ERR_STATEMENT, "rotate: the log at %u..%llu lies outside the hashed digest bytes of the %u-byte header", start_position + 1,
                       (unsigned long long)start_position + 1 + wlen, header_size);
    struct Joiner {  // every exit path waits for the threads
        JustificationTables& j;
        TableJob &a, &b;
        ~Joiner() {
            for (int t = 0; t < 3; ++t)
                if (j.job[t].th.joinable()) j.job[t].th.join();
            if (a.th.joinable()) a.th.join();
            if (b.th.joinable()) b.th.join();
        }
    } joiner{jt, epoch, newset};
    {
        const int32_t rs = vx_justification_tables_start(side, just, cfg, &rv, 0, nullptr, nullptr, &jt);
        if (rs != VX_OK) return vx_fail(ctx, rs, "rotate: no host thread for the justification tables");
    }
    auto prove_epoch = [&](vx_ctx* c, TableJob& j) -> int32_t {
        size_t bound = 0;
        VX_TRY(vx_stark_proof_bound(VX_AIR_EPOCH_END, cfg, VX_EPOCH_END_LOG_ROWS, &bound));
        j.proof.resize(bound);
        return vx_stark_prove_impl(c, VX_AIR_EPOCH_END, cfg, et->d, et->n, /*consume_trace=*/0, VX_EPOCH_END_LOG_ROWS, epub, 10, j.proof.data(), j.proof.size(), &j.len, &hb[1]);
    };
    auto prove_newset = [&](vx_ctx* c, TableJob& j) -> int32_t {  // the output (rotate.rs:317-320): receives every key from the epoch-end table
        const int sl = sha_rows_log(num_authorities);
        size_t bound = 0;
        VX_TRY(vx_stark_proof_bound(VX_AIR_SHA_CHAIN, cfg, sl, &bound));
        j.proof.resize(bound);
        vx_buf* st = nullptr;
        VX_TRY(vx_alloc(c, ((size_t)VX_SHA_AIR_COLS) << sl, &st));
        uint64_t spub[10];
        int32_t r = vx_sha_chain_trace_dev(c, new_pubkeys, num_authorities, nullptr, 2, sl, st->d, spub, new_commit);
        if (r == VX_OK) r = vx_stark_prove_impl(c, VX_AIR_SHA_CHAIN, cfg, st->d, st->n, /*consume_trace=*/0, sl, spub, 10, j.proof.data(), j.proof.size(), &j.len, &hb[2]);
        (void)vx_free(c, st);
        return r;
    };
    int32_t rc = VX_OK;
    try {
        epoch.th = std::thread([&] {
            (void)hipSetDevice(epoch.c->device);
            epoch.rc = prove_epoch(epoch.c, epoch);
            if (epoch.rc != VX_OK) rvb.fail();  // do not leave the other provers waiting at their hooks
        });
        newset.th = std::thread([&] {
            (void)hipSetDevice(newset.c->device);
            newset.rc = prove_newset(newset.c, newset);
            if (newset.rc != VX_OK) rvb.fail();
        });
    } catch (...) {
        rvb.fail();
        rc = vx_fail(ctx, VX_ERR_DEVICE, "rotate: no host thread for the epoch-end tables");
    }
    // 2. header hash = Blake2b-256 of the first header_size bytes (rotate.rs:293); the trace of its compressions
    //    is the witness of the hash STARK (one-header chain anchored at the header's own parent hash)
    const size_t chunks = (header_size + 127) / 128;
    const int bl = blake_rows_log(chunks);
    vx_buf* trace = nullptr;
    if (rc == VX_OK) rc = vx_alloc(ctx, ((size_t)VX_BLAKE_AIR_COLS) << bl, &trace);
    uint64_t pub[20];
    if (rc == VX_OK) rc = vx_blake_chain_trace(ctx, header, MAX_HEADER_SIZE, &header_size, 1, parent, epoch_end_block_number, 0, start_position + 1, wlen, bl, trace, pub, header_hash);
    // 3. justification by the current set over (epoch_end_block_number, header hash) (rotate.rs:297-302), natively
    if (rc == VX_OK)
        rc = vx_verify_simple_justification(ctx, epoch_end_block_number, header_hash, just->authority_set_id, just->authority_set_hash,
                                            just->precommit, just->pubkeys, just->signatures, just->validator_signed, just->num_authorities,
                                            just->max_authorities);
    size_t len[3] = {0, 0, 0};
    int32_t rc_room = VX_OK;
    auto room = [&](size_t off) { return rc_room == VX_OK && proof_out && proof_cap > off; };
    if (rc == VX_OK) {
        rc = vx_stark_prove_impl(ctx, VX_AIR_BLAKE_CHAIN, cfg, trace->d, trace->n, 1, bl, pub, 20, room(VX_ROT_HDR) ? proof_out + VX_ROT_HDR : nullptr,
                                 room(VX_ROT_HDR) ? proof_cap - VX_ROT_HDR : 0, &len[0], &hb[0]);
        if (rc == VX_ERR_BUFSZ) rc_room = rc, rc = VX_OK;
    }
    if (rc != VX_OK) rvb.fail();
    if (trace) (void)vx_free(ctx, trace);
    // 4. collect the side tables
    if (epoch.th.joinable()) epoch.th.join();
    if (newset.th.joinable()) newset.th.join();
    const int32_t rc_just = vx_justification_tables_join(ctx, &jt);
    if (rc == VX_OK && rc_just != VX_OK) rc = rc_just;
    for (TableJob* j : {&epoch, &newset})
        if (rc == VX_OK && j->rc != VX_OK) rc = vx_fail(ctx, j->rc, "rotate: %s", vx_last_error(j->c)[0] ? vx_last_error(j->c) : "an epoch-end table failed");
    len[1] = jt.job[0].len, len[2] = newset.len;
    const size_t len_ed = jt.job[1].len, len_h = jt.job[2].len, len_ep = epoch.len, total = VX_ROT_HDR + len[0] + len[1] + len[2] + len_ed + len_h + len_ep;
    if (rc == VX_OK && rc_room == VX_OK && proof_out && proof_cap >= total) {
        size_t off = VX_ROT_HDR + len[0];
        memcpy(proof_out + off, jt.job[0].proof.data(), len[1] * 8), off += len[1];
        memcpy(proof_out + off, newset.proof.data(), len[2] * 8), off += len[2];
        memcpy(proof_out + off, jt.job[1].proof.data(), len_ed * 8), off += len_ed;
        memcpy(proof_out + off, jt.job[2].proof.data(), len_h * 8), off += len_h;
        memcpy(proof_out + off, epoch.proof.data(), len_ep * 8);
    }
    if (rc != VX_OK) return rc;
    *proof_len = total;
    memcpy(out32, new_commit, 32);
    if (rc_room != VX_OK || !proof_out || proof_cap < *proof_len)
        return vx_fail(ctx, VX_ERR_BUFSZ, "rotate: proof needs %zu words, buffer has %zu", *proof_len, proof_cap);
    proof_out[0] = VX_ROT_MAGIC;
    proof_out[1] = just->authority_set_id;
    proof_out[2] = epoch_end_block_number;
    proof_out[3] = num_authorities;
    memcpy(proof_out + 4, header_hash, 32);
    memcpy(proof_out + 8, just->authority_set_hash, 32);
    memcpy(proof_out + 12, new_commit, 32);
    proof_out[16] = len[0];
    proof_out[17] = len[1];
    proof_out[18] = len[2];
    proof_out[19] = len_ed;
    memcpy(proof_out + 20, parent, 32);
    proof_out[24] = len_h;
    uint64_t round = 0;
    memcpy(&round, just->precommit + 37, 8);  // 0x01 || hash 32 || block 4 || round 8 || set id 8 (decoder.rs:159-200)
    proof_out[25] = round;
    proof_out[26] = start_position;
    proof_out[27] = len_ep;
    return VX_OK;
}

}  // extern "C"
