// Host-side plumbing shared by the circuit provers (vx_header_range_prove, vx_rotate_prove): the tables of one statement
// sit on ONE logUp bus and must use the same lookup challenges, drawn after every trace is committed.  Each table is proven
// from its own host thread on its own context; the provers stop after their trace caps (vx_chal_hook) and MEET: every one
// deposits its public inputs + cap, waits for all the others and derives the challenges from the transcript of all
// (public inputs, cap) pairs in table order.
#pragma once
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "vx_internal.h"

// A statement proven by several processes (one per GPU, vx_header_range_prove_ex with n_shards > 1): `local[t]` says which
// of the n_parties tables are proven here; once every LOCAL table has arrived -- or failed: a failed table arrives as a
// failure marker, so that the other shards are not left waiting -- one thread calls `xch` with an array of n_parties slots
// (only the local ones filled) and gets the union back.
struct BusMeet {
    static constexpr int MAX = 72, MAX_PUB = 32;
    std::mutex m;
    std::condition_variable cv;
    int n_parties = 0, arrived = 0;
    bool failed = false, done = false;
    std::vector<uint64_t> pub[MAX], cap[MAX];
    bool local[MAX], deposited[MAX], local_failed[MAX];
    size_t capw = 64;  // words of a trace cap (4 << cap_height); set by the caller for a sharded proof, by the first arrival otherwise
    const vx_hr_exchange* xch = nullptr;
    BusMeet() {
        for (int t = 0; t < MAX; ++t) local[t] = true, deposited[t] = local_failed[t] = false;
    }
    int n_local() const {
        int k = 0;
        for (int t = 0; t < n_parties; ++t) k += local[t] ? 1 : 0;
        return k;
    }
    static int32_t meet(BusMeet* r, int who, const uint64_t* pub, size_t n_pub, const uint64_t* cap, size_t cap_words, uint64_t* chal, size_t n_chal);
    void fail(int who = -1);  // who >= 0: the table whose prover gave up (counts as arrived when it had not deposited yet)
    void finish_locked(size_t cap_words);  // all local tables are in: exchange with the other shards (if any), release everybody
};
struct BusParty {
    BusMeet* rv;
    int who;
};
int32_t vx_bus_hook(void* party, const uint64_t* pub, size_t n_pub, const uint64_t* cap, size_t cw, uint64_t* chal, size_t n_chal);
// one table of the statement, proven from its own host thread on its own context
struct TableJob {
    vx_ctx* c = nullptr;
    std::thread th;
    int32_t rc = VX_OK;
    std::vector<uint64_t> proof;
    size_t len = 0;
};

// The three tables of a justification -- authority-set commitment (ShaChainAir, sends the chosen signers' keys), Ed25519
// (EdAir) and SHA-512 (Sha512Air) -- as parties first, first + 1, first + 2 of `rv`.  The prover verifies exactly
// floor(2n/3) + 1 signatures (the first signed ones).  `pre` (may be empty) runs on the commitment's thread before anything
// is proven: the native statement checks whose failure must name the error.
struct JustificationTables {
    BusParty party[3];
    vx_chal_hook hooks[3];
    TableJob job[3];  // commitment, Ed25519, SHA-512
    std::vector<uint8_t> chosen;
    size_t n_sig = 0;
};
size_t vx_justification_proof_bound(const vx_stark_config* cfg, size_t n_authorities, int32_t* rc);
int32_t vx_justification_tables_start(vx_ctx* const ctxs[3], const vx_justification* just, const vx_stark_config* cfg, BusMeet* rv, int first,
                                      int32_t (*pre)(vx_ctx*, void*), void* pre_user, JustificationTables* jt, unsigned mask = 7 /* bit t: start table t here */);
// joins the three threads; the first failure with a message of its own is reported on `ctx` (the justification's own rules
// come first); returns VX_OK when all three proofs exist
int32_t vx_justification_tables_join(vx_ctx* ctx, JustificationTables* jt);
// verifier side: expected public inputs and AIR ids of the three tables for the request; the counts are read from the
// proofs' own public inputs (ppub_chain[8] authorities, ppub_ed[0] signatures) and must satisfy signed * 3 > authorities * 2
int32_t vx_justification_expect(const uint64_t* ppub_chain, size_t n_chain, const uint64_t* ppub_ed, size_t n_ed, size_t n_s512, const uint8_t authority_set_hash[32],
                                uint64_t authority_set_id, const uint8_t block_hash[32], uint32_t block_number, uint64_t round, uint64_t spub[10], uint64_t epub[2],
                                uint64_t hpub[15], int air[3], char* err, size_t errlen);
void vx_shared_challenges_host(const uint64_t* const* pubs, const size_t* n_pubs, const uint64_t* const* caps, size_t k, size_t cap_words, uint64_t* out, size_t n_out);

// ---- rotate blob (written by vx_rotate_prove in vx_rotate.hip, read by vx_rotate_verify in vx_verify.hip)
static const uint64_t VX_ROT_MAGIC = 0x3354415458525856ULL;  // "VXRXTAT3"
// magic, set id, block, n_new, header hash (4), set hash (4), new set hash (4), proof lengths: header hash, current-set commitment,
// new-set commitment, Ed25519; parent hash (4); SHA-512 proof length, the precommit's round, start_position, epoch-end proof length
static constexpr size_t VX_ROT_HDR = 28;

static constexpr uint32_t VX_MAX_HEADER_SIZE = 35840;  // consts.rs:16
static inline void be_limbs(const uint8_t h[32], uint64_t out[8]) {
    for (int j = 0; j < 8; ++j)
        out[j] = ((uint64_t)h[4 * j] << 24) | ((uint64_t)h[4 * j + 1] << 16) | ((uint64_t)h[4 * j + 2] << 8) | h[4 * j + 3];
}
