// ShaTreeAir (AIR ids 7 / 8 / 9 for trees of 256 / 512 / 16 leaves): the two SHA-256 Merkle roots HeaderRangeCircuit
// outputs -- state_root_merkle_root and data_root_merkle_root (/root/reference
// circuits/builder/subchain_verification.rs:213-220 for the 8-leaf subtrees of a map job, :268-274 for the reduce
// nodes; native mirror circuits/input/mod.rs:464-528: unhashed 32-byte leaves, node = SHA256(l || r), zero leaves beyond
// the range).  The reference spreads the tree over 2J-1 recursive proofs; here it is ONE table whose leaves arrive from
// the Blake2b AIR (air_blake.cuh) over a logUp bus, so the roots are bound to the very header bytes that were hashed.
// A node takes 128 rows: the DATA compression of l || r (start state IV) and the constant PAD compression of a 64-byte
// message; node g of tree t sits at rows 128 (t N + g) in heap numbering (1 = root, children 2g / 2g+1, leaves N..2N-1,
// slot 0 a dummy).  Compression rows and column layout are ShaChainAir's (air_sha.cuh, 539 columns); everything positional is a
// PERIODIC column; the only other witness is the pair of leaf-enable flags ENL / ENR of a bottom-level node (a disabled
// leaf must be zero and takes nothing from the bus) and their running count CNT.  The flags are NOT the prover's to choose:
// they are boolean, constant over a node, non-increasing in leaf order, and their count over each tree is public input 16 =
// the number of headers of the range (the verifier sets it to target_block - trusted_block), i.e. leaf i is enabled exactly
// when i < n -- every header's two roots MUST be taken from the bus, so a prover can no longer drop a header's sends on the
// Blake2b side together with its receives here.  Row r < 16 of a DATA block receives message word r:
//   inner nodes:                 (tree, child id, r mod 8, word, TAG_WORD)                         from the children's PAD blocks
//   bottom level of both trees:  (leaf, 4 (r mod 8) + q, byte q of the word, tree, TAG_BYTE), q = 0..3   from the header bytes
// and row 63 of a PAD block sends the node's digest (tree, g, j, word_j), except for the root: its digest is public.
// Constraint ORDER is protocol: oracle/sha_tree_air.py restates it independently.
#pragma once
#include <vector>

#include "air_blake.cuh"
#include "air_sha.cuh"

namespace sht {
constexpr int ENL = shc::DG0, ENR = shc::DG0 + 1, CNT = shc::DG0 + 2, COLS = shc::COLS, AUX = 16, N_PERIODIC = 17;
// P_NB / P_BB / P_LASTN sit on the last row of a node (row 63 of its PAD block): the next node is a bottom-level node / this
// node and the next both are / this is the last node of its tree
enum { P_SEL0, P_SEL63, P_SCHED, P_K, P_DATA, P_TREE, P_PWA, P_PBL, P_PBR, P_CID, P_JJ, P_PS, P_ROOT, P_GID, P_NB, P_BB, P_LASTN };
}  // namespace sht

template <int LOGN, int ID_>
struct ShaTreeAirT {
    static constexpr int ID = ID_, COLS = sht::COLS, PUB = 17, PERIODIC = sht::N_PERIODIC, PERIOD_LOG = 8 + LOGN, QUOT_ROWS_PER_LANE = 1, AUX = sht::AUX, CHAL = 4, AUXPUB = 1, EXACT_LOG = 1;
    static constexpr int TREE_SIZE = 1 << LOGN;
    static constexpr int plog(int q) { return q < 4 ? 6 : (q == 4 ? 7 : 8 + LOGN); }  // 4 x 64, 128, then 12 full-period columns

    // one period of every periodic column, back to back (host)
    static void periodic_values(std::vector<uint64_t>& v) {
        using namespace sht;
        const size_t N = TREE_SIZE, n = 256 * N;
        v.assign(4 * 64 + 128 + 12 * n, 0);
        uint64_t* p = v.data();
        p[0] = 1, p[64 + 63] = 1;
        for (int r = 0; r <= 47; ++r) p[128 + r] = 1;
        for (int r = 0; r < 64; ++r) p[192 + r] = shc::K_H[r];
        for (int r = 0; r < 64; ++r) p[256 + r] = 1;
        uint64_t* q = p + 384;  // columns P_TREE .. P_GID, n values each
        for (size_t row = 0; row < n; ++row) {
            const size_t r = row & 63, blk = (row >> 6) & 1, pair = row >> 7, tree = pair / N, g = pair % N;
            const bool msg = blk == 0 && r < 16, left = msg && r < 8, right = msg && r >= 8, bottom = g >= N / 2, inner = g >= 1 && g < N / 2;
            const size_t c = r >= 8 ? 1 : 0;
            const bool send = blk == 1 && r == 63;
            q[(P_TREE - 5) * n + row] = tree;
            q[(P_PWA - 5) * n + row] = msg && inner;
            q[(P_PBL - 5) * n + row] = left && bottom;
            q[(P_PBR - 5) * n + row] = right && bottom;
            q[(P_CID - 5) * n + row] = msg ? (bottom ? 2 * g - N + c : 2 * g + c) : 0;  // a bottom-level child is a leaf: its index; an inner child: its node id
            q[(P_JJ - 5) * n + row] = msg ? (r & 7) : 0;
            q[(P_PS - 5) * n + row] = send && g >= 2;
            q[(P_ROOT - 5) * n + row] = send && g == 1;
            q[(P_GID - 5) * n + row] = send ? g : 0;
            q[(P_NB - 5) * n + row] = send && g + 1 >= N / 2 && g + 1 < N;
            q[(P_BB - 5) * n + row] = send && g >= N / 2 && g + 1 < N;
            q[(P_LASTN - 5) * n + row] = send && g == N - 1;
        }
    }

    template <class F, class Row, class C>
    __host__ __device__ static void eval(const Row& loc, const Row& nxt, const F* per, const F* pub, const F* chal, const F* apub, C& c) {
        using namespace shc;
        using namespace sht;
        const F sel0 = per[P_SEL0], is_data = per[P_DATA];
        const F one = F::from(1);
        auto val = [&](const Row& row, int col0, int nb) -> F {
            F acc = row[col0 + nb - 1];
#pragma unroll 1
            for (int i = nb - 2; i >= 0; --i) acc = acc + acc + row[col0 + i];
            return acc;
        };
        auto window = [&](const Row& row, int p) -> F { return p == 0 ? val(row, W0B, 32) : p == 1 ? val(row, W1B, 32) : p == 14 ? val(row, W14B, 32) : row[WV(p)]; };
        // ---- 1-7. the compression rows (shared with ShaChainAir); a PAD block continues from its DATA block
        sha_compression_constraints<F>(loc, nxt, per, is_data, c);
        // ---- 8. the PAD block's message, the root, zero leaves
#pragma unroll 1
        for (int j = 0; j < 16; ++j) c.constraint(sel0 * (one - is_data) * (window(loc, j) - F::from(pad64(j))));
#pragma unroll 1
        for (int j = 0; j < 8; ++j) c.constraint(per[P_ROOT] * (loc[FFV0 + j] - (pub[j] + per[P_TREE] * (pub[8 + j] - pub[j]))));
        const F w0 = val(loc, W0B, 32);
        c.constraint(per[P_PBL] * (one - loc[ENL]) * w0);
        c.constraint(per[P_PBR] * (one - loc[ENR]) * w0);
        // ---- 8b. the leaf-enable flags are forced: leaf i of either tree is enabled exactly when i < pub[16]
        {
            const F enl = loc[ENL], enr = loc[ENR], end = per[P_SEL63] * (one - is_data), keep = one - end;
            c.constraint(enl * (enl - one));
            c.constraint(enr * (enr - one));
            c.constraint(enr * (one - enl));                                   // left before right
            c.constraint(keep * (nxt[ENL] - enl));                             // constant over the 128 rows of a node
            c.constraint(keep * (nxt[ENR] - enr));
            c.constraint(keep * (nxt[CNT] - loc[CNT]));
            c.constraint(per[P_BB] * nxt[ENL] * (one - enr));                  // non-increasing from node to node
            // the count restarts at slot 0 of a tree and takes in the flags of every bottom-level node
            c.constraint(end * (nxt[CNT] - loc[CNT]) + per[P_LASTN] * loc[CNT] - per[P_NB] * (nxt[ENL] + nxt[ENR]));
            c.constraint(per[P_LASTN] * (loc[CNT] - pub[16]));
        }
        // ---- 9. the bus (logUp): 13 lookups in 7 helper elements of the local row, cyclic running sum
        {
            const X2<F> beta{chal[0], chal[1]}, gamma{chal[2], chal[3]}, g2 = gamma * gamma, g3 = g2 * gamma, g4 = g2 * g2;
            const F en_l = loc[ENL], en_r = loc[ENR], zero = F::from(0);
            const X2<F> tag_w = g4 * F::from(blk::TAG_WORD), tag_b = g4 * F::from(blk::TAG_BYTE);
            // lookup q: 0 = word receive, 1..4 = byte receives, 5..12 = digest sends
            auto mult = [&](int q) -> F {
                if (q == 0) return zero - per[P_PWA];
                if (q <= 4) return zero - (per[P_PBL] * en_l + per[P_PBR] * en_r);
                return per[P_PS];
            };
            auto denom = [&](int q) -> X2<F> {
                if (q == 0) return beta + per[P_TREE] + gamma * per[P_CID] + g2 * per[P_JJ] + g3 * w0 + tag_w;
                if (q <= 4) return beta + per[P_CID] + gamma * (per[P_JJ] * F::from(4) + F::from((uint64_t)(q - 1))) + g2 * val(loc, W0B + 24 - 8 * (q - 1), 8) + g3 * per[P_TREE] + tag_b;
                return beta + per[P_TREE] + gamma * per[P_GID] + g2 * F::from((uint64_t)(q - 5)) + g3 * loc[FFV0 + q - 5] + tag_w;
            };
            X2<F> hsum{zero, zero};
#pragma unroll 1
            for (int e = 0; e < 7; ++e) {
                const X2<F> h{loc[COLS + 2 * e], loc[COLS + 2 * e + 1]};
                const X2<F> du = denom(2 * e);
                if (e < 6) {
                    const X2<F> dv = denom(2 * e + 1);
                    c.constraint_x2(h * du * dv - dv * mult(2 * e) - du * mult(2 * e + 1));
                } else c.constraint_x2(h * du - mult(2 * e));
                hsum = hsum + h;
            }
            const X2<F> z{loc[COLS + 14], loc[COLS + 15]}, zn{nxt[COLS + 14], nxt[COLS + 15]};
            c.constraint_x2(zn - z - hsum + X2<F>{apub[0], apub[1]});
        }
    }
};
using ShaTreeAir256 = ShaTreeAirT<8, 7>;
using ShaTreeAir512 = ShaTreeAirT<9, 8>;
using ShaTreeAir16 = ShaTreeAirT<4, 9>;
