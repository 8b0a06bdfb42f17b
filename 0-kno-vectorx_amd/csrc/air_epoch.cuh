// EpochEndAir (AIR id 15): verify_epoch_end_header in-proof (/root/reference circuits/builder/rotate.rs:74-174 verify_prefix,
// :176-276 verify_epoch_end_header): the bytes of the epoch-end header from start_position + 1 on are the consensus flag 4, the
// engine id "FRNK", a SCALE compact length, the scheduled-change flag 1, the compact number n of new authorities, n records
// (32-byte key, weight 1 as u64 LE) and a zero u32 delay -- and those keys are the new authority set.  The bytes come over the
// logUp bus from the Blake2b table that hashes the header (air_blake.cuh, bus mode 2: tuples (0, k, byte, 1), TAG_BYTE), the keys
// leave as the TAG_KEY tuples ShaChainAir takes in its receive mode, so the new set's commitment is the commitment of these
// header bytes.  512 rows: row 0 the prefix (P = 6 + len1 + len2 bytes), rows 1..n the validators (k = P + 40 (i - 1) + j), row
// n + 1 the delay.  44 byte cells (range: they equal message bytes the Blake2b table range-checks), row flags V / DL, six bits Q
// of the first length byte >> 2.  Public inputs: n, bus_on, len1 one-hot (1, 2, 4, 5 bytes), len2 one-hot.
// Constraint ORDER is protocol: oracle/epoch_air.py restates it independently.
#pragma once
#include <vector>

#include "air.cuh"
#include "air_blake.cuh"
#include "air_ed.cuh"

namespace epo {
constexpr int LOG_N = 9, NB = 44, V = 44, DL = 45, Q0 = 46, COLS = 52, N_HELP = 23, AUX = 2 * N_HELP;
VX_HD constexpr int len_of(int a) { return a == 0 ? 1 : a == 1 ? 2 : a == 2 ? 4 : 5; }
}  // namespace epo

struct EpochEndAir {
    static constexpr int ID = 15, COLS = epo::COLS, PUB = 10, PERIODIC = 2, PERIOD_LOG = epo::LOG_N, QUOT_ROWS_PER_LANE = 1, AUX = epo::AUX, CHAL = 4, AUXPUB = 1, EXACT_LOG = 1;
    static constexpr int plog(int) { return epo::LOG_N; }
    static void periodic_values(std::vector<uint64_t>& v) {
        const size_t n = (size_t)1 << epo::LOG_N;
        v.assign(2 * n, 0);
        v[0] = 1;
        for (size_t i = 1; i < n; ++i) v[n + i] = i - 1;  // record index: one column, so that byte positions stay of degree 1
    }

    template <class F, class Row, class Cn>
    __host__ __device__ static void eval(const Row& loc, const Row& nxt, const F* per, const F* pub, const F* chal, const F* apub, Cn& c) {
        using namespace epo;
        const F one = F::from(1), zero = F::from(0), r0 = per[0], rec = per[1], n_auth = pub[0], on = pub[1];
        const F k8 = F::from(256), k16 = F::from(65536), k24 = F::from(1ULL << 24);
        // ---- 1. row flags
        {
            const int cols[8] = {V, DL, Q0, Q0 + 1, Q0 + 2, Q0 + 3, Q0 + 4, Q0 + 5};
#pragma unroll 1
            for (int q = 0; q < 8; ++q) {
                const F x = loc[cols[q]];
                c.constraint(x * (x - one));
            }
        }
        const F v = loc[V], dl = loc[DL];
        c.constraint(r0 * v);
        c.constraint(r0 * dl);
        c.constraint(r0 * (nxt[V] - one));              // row 1 is a validator
        c.constraint(nxt[DL] - v * (one - nxt[V]));     // the delay row follows the last validator
        c.constraint((one - r0) * nxt[V] * (one - v));  // validators are rows 1..n
        c.constraint(dl * (rec - n_auth));              // ... and n is the public count
        // ---- 2. validator and delay rows
        c.constraint(v * (loc[32] - one));
#pragma unroll 1
        for (int j = 33; j < 40; ++j) c.constraint(v * loc[j]);
#pragma unroll 1
        for (int j = 0; j < 4; ++j) c.constraint(dl * loc[j]);
        // ---- 3. the prefix (row 0): flag, engine id, compact length (any value, well-formed), scheduled change, compact n
        {
            const uint64_t want[5] = {4, 70, 82, 78, 75};
#pragma unroll 1
            for (int j = 0; j < 5; ++j) c.constraint(r0 * (loc[j] - F::from(want[j])));
        }
        const F* l1 = pub + 2;
        const F* l2 = pub + 6;
        {
            F q = loc[Q0 + 5];
#pragma unroll 1
            for (int i = 4; i >= 0; --i) q = q + q + loc[Q0 + i];
            c.constraint(r0 * (loc[5] - q * F::from(4) - (l1[1] + l1[2] * F::from(2) + l1[3] * F::from(3))));
            c.constraint(r0 * l1[3] * q);
            F acc = zero;
#pragma unroll 1
            for (int a = 0; a < 4; ++a) acc = acc + l1[a] * (loc[5 + len_of(a)] - one);
            c.constraint(r0 * acc);
            F acc2 = zero, acc3 = zero;
#pragma unroll 1
            for (int a = 0; a < 4; ++a) {
                const int o = 6 + len_of(a);
                const F b0 = loc[o], b1 = loc[o + 1], b2 = loc[o + 2], b3 = loc[o + 3], b4 = loc[o + 4], four = F::from(4);
                const F dec[4] = {b0 - n_auth * four, b0 + b1 * k8 - n_auth * four - one, b0 + b1 * k8 + b2 * k16 + b3 * k24 - n_auth * four - F::from(2),
                                  b1 + b2 * k8 + b3 * k16 + b4 * k24 - n_auth};
#pragma unroll 1
                for (int b = 0; b < 4; ++b) acc2 = acc2 + l1[a] * l2[b] * dec[b];
                acc3 = acc3 + l1[a] * l2[3] * (b0 - F::from(3));
            }
            c.constraint(r0 * acc2);
            c.constraint(r0 * acc3);
        }
        // ---- 4. the bus: 40 byte receives, 4 key sends, two lookups per helper
        {
            const X2<F> beta{chal[0], chal[1]}, gamma{chal[2], chal[3]}, g2 = gamma * gamma, g3 = g2 * gamma, g4 = g2 * g2;
            F l1len = zero, l2len = zero;
#pragma unroll 1
            for (int a = 0; a < 4; ++a) l1len = l1len + l1[a] * F::from((uint64_t)len_of(a)), l2len = l2len + l2[a] * F::from((uint64_t)len_of(a));
            const F plen = l1len + l2len + F::from(6), kbase = (one - r0) * plen + rec * F::from(40);
            const X2<F> bbase = beta + g3 + g4 * F::from(blk::TAG_BYTE);  // (0, k, byte, tree 1)
            auto m_byte = [&](int j) -> F {
                F pm = zero;  // [j < P] on the prefix row
#pragma unroll 1
                for (int a = 0; a < 4; ++a)
#pragma unroll 1
                    for (int b = 0; b < 4; ++b)
                        if (j < 6 + len_of(a) + len_of(b)) pm = pm + l1[a] * l2[b];
                F m = v + r0 * pm;
                if (j < 4) m = m + dl;
                return zero - m * on;
            };
            auto d_byte = [&](int j) -> X2<F> { return bbase + gamma * (kbase + F::from((uint64_t)j)) + g2 * loc[j]; };
            auto d_key = [&](int q) -> X2<F> {
                const F la = loc[8 * q] + loc[8 * q + 1] * k8 + (loc[8 * q + 2] + loc[8 * q + 3] * k8) * k16;
                const F lb = loc[8 * q + 4] + loc[8 * q + 5] * k8 + (loc[8 * q + 6] + loc[8 * q + 7] * k8) * k16;
                return beta + (rec * F::from(4) + F::from((uint64_t)q)) + gamma * la + g2 * lb + g4 * F::from(edc::TAG_KEY);
            };
            X2<F> hsum{zero, zero};
#pragma unroll 1
            for (int e = 0; e < 20; ++e) {
                const X2<F> du = d_byte(2 * e), dv = d_byte(2 * e + 1);
                const X2<F> h{loc[COLS + 2 * e], loc[COLS + 2 * e + 1]};
                c.constraint_x2(h * du * dv - dv * m_byte(2 * e) - du * m_byte(2 * e + 1));
                hsum = hsum + h;
            }
            const F mk = v * on;
#pragma unroll 1
            for (int e = 0; e < 2; ++e) {
                const X2<F> du = d_key(2 * e), dv = d_key(2 * e + 1);
                const X2<F> h{loc[COLS + 2 * (20 + e)], loc[COLS + 2 * (20 + e) + 1]};
                c.constraint_x2(h * du * dv - dv * mk - du * mk);
                hsum = hsum + h;
            }
            const X2<F> z{loc[COLS + 2 * (N_HELP - 1)], loc[COLS + 2 * (N_HELP - 1) + 1]}, zn{nxt[COLS + 2 * (N_HELP - 1)], nxt[COLS + 2 * (N_HELP - 1) + 1]};
            c.constraint_x2(zn - z - hsum + X2<F>{apub[0], apub[1]});
        }
    }
};
