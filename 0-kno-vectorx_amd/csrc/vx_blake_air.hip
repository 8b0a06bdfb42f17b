// K8: witness/trace generation for BlakeChainAir on the GPU, and the header_range prove entry.
//   kernel A  k_blake_chain: one lane per header -- digest + the chaining value before every
//             128-byte chunk (the only sequential part of BLAKE2b);
//   kernel B  k_blake_trace: one lane per TRACE ROW (block b, r = row mod 16): recomputes
//             the <= 12 rounds it needs from the chunk's chaining value and writes the row
//             packed as 96 words; k_blake_expand unpacks words into the 729 byte / limb / bit
//             cells, a block writing 16 KB runs of one column at a time;
//   kernel C  k_blake_aux: the logUp helper columns once the lookup challenges are known.
// Replaces the curta Blake2b witness generation behind hash_encoded_header
// (/root/reference circuits/builder/header.rs:14-19) for the synthetic header chain.
#include <string.h>

#include "air_blake.cuh"
#include <condition_variable>
#include <mutex>
#include <thread>

#include "vx_bus.h"
#include "vx_internal.h"


// SCALE compact mode of a u32 (decoder.rs:39-92): 1 / 2 / 4 / 5 bytes
static uint8_t compact_mode(uint32_t v) { return v < (1u << 6) ? 0 : v < (1u << 14) ? 1 : v < (1u << 30) ? 2 : 3; }

struct BlockDesc {
    uint64_t msg_off;  // byte offset of the 128-byte chunk in the headers buffer; ~0 = padding block
    uint32_t t, inc;
    uint32_t D[8];     // digest register before this block
    uint8_t fin, first, act, mode;  // mode: SCALE compact mode (0..3) of the header's block number
    uint32_t num;      // block number of the header this chunk belongs to
    uint32_t size;     // length of the whole message (< 2^24)
};

__device__ __forceinline__ uint64_t b_rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

__device__ __forceinline__ void carries(uint64_t o1, uint64_t o2, uint64_t o3, uint8_t* out) {
    uint64_t lo = (o1 & 0xFFFFFFFFULL) + (o2 & 0xFFFFFFFFULL) + (o3 & 0xFFFFFFFFULL);
    uint64_t klo = lo >> 32;
    uint64_t hi = (o1 >> 32) + (o2 >> 32) + (o3 >> 32) + klo;
    out[0] = (uint8_t)klo;
    out[1] = (uint8_t)(hi >> 32);
}
__device__ __forceinline__ void g_mix(uint64_t* v, int ia, int ib, int ic, int id, uint64_t x, uint64_t y) {
    uint64_t a = v[ia], b = v[ib], c = v[ic], d = v[id];
    uint64_t a1 = a + b + x, d1 = b_rotr(d ^ a1, 32), c1 = c + d1, b1 = b_rotr(b ^ c1, 24);
    uint64_t a2 = a1 + b1 + y, d2 = b_rotr(d1 ^ a2, 16), c2 = c1 + d2, b2 = b_rotr(b1 ^ c2, 63);
    v[ia] = a2, v[ib] = b2, v[ic] = c2, v[id] = d2;
}
__device__ void blake_round(uint64_t* v, const uint64_t* m, int round) {
    const uint8_t* s = blk::ORDER[round + 1];  // ORDER[r] for r = 1..12 is sigma[r-1]
    for (int k = 0; k < 4; ++k) g_mix(v, k, 4 + k, 8 + k, 12 + k, m[s[2 * k]], m[s[2 * k + 1]]);
    for (int j = 0; j < 4; ++j) g_mix(v, j, 4 + (j + 1) % 4, 8 + (j + 2) % 4, 12 + (j + 3) % 4, m[s[8 + 2 * j]], m[s[8 + 2 * j + 1]]);
}
__device__ void blake_init_v(uint64_t* v, const uint64_t* h, uint64_t t, bool fin) {
    for (int i = 0; i < 8; ++i) v[i] = h[i], v[8 + i] = blk::IV[i];
    v[12] ^= t;
    if (fin) v[14] = ~v[14];
}

// one lane per header: digest and the chaining value in front of each chunk
__global__ __launch_bounds__(64) void k_blake_chain(const uint8_t* msgs, size_t stride, const uint32_t* sizes, size_t n,
                                                    const uint32_t* block_base, uint64_t* hchain, uint8_t* digests) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t* p = (const uint64_t*)(msgs + i * stride);
    const uint32_t len = sizes[i];
    const uint32_t nchunks = len == 0 ? 1 : (len + 127) / 128;
    uint64_t h[8], m[16], v[16];
    for (int k = 0; k < 8; ++k) h[k] = blk::IV[k];
    h[0] ^= 0x01010020ULL;
    for (uint32_t cidx = 0; cidx < nchunks; ++cidx) {
        uint64_t* hc = hchain + 8 * ((size_t)block_base[i] + cidx);
        for (int k = 0; k < 8; ++k) hc[k] = h[k];
        const bool fin = cidx + 1 == nchunks;
        const uint32_t off = 128 * cidx, rem = fin ? len - off : 128;
        for (int k = 0; k < 16; ++k) {
            uint32_t b = 8 * k;
            uint64_t w = 0;
            if (b < rem) {
                w = p[(off >> 3) + k];
                if (rem - b < 8) w &= (1ULL << (8 * (rem - b))) - 1;
            }
            m[k] = w;
        }
        blake_init_v(v, h, fin ? len : off + 128, fin);
        for (int r = 0; r < 12; ++r) blake_round(v, m, r);
        for (int k = 0; k < 8; ++k) h[k] ^= v[k] ^ v[k + 8];
    }
    uint64_t* d = (uint64_t*)(digests + 32 * i);
    d[0] = h[0], d[1] = h[1], d[2] = h[2], d[3] = h[3];
}

// ---- trace rows ---------------------------------------------------------------------------------------------------
// A row lane that stored its 729 cells one column at a time walked 729 columns 8n bytes apart: the walk's address
// translation, not its bytes, set the time (7.8 ms for 3 GB).  So the row is produced in two steps:
//   k_blake_trace   one lane per trace row (block b, r = row mod 16): recomputes the <= 12 rounds it needs from the
//                   chunk's chaining value and stores the row PACKED as 96 words (stage[w * n + row]); the same lane knows
//                   every XOR its row looks up and bumps the multiplicity histograms of the two tables
//                   (hist[0 .. 2^16) for T1, hist[2^16 .. 2^17) for T2);
//   k_blake_expand  one block per (2048 rows, packed word): every cell is (word >> shift) & mask, and a block writes
//                   16 KB runs of one column at a time.
// Packed words: 8 per G (A1 D1 C1 B1 A2 D2 C2 X; L / T are the low 7 bits / top bit of X's bytes), then
#ifndef VX_HIST_COPIES
#define VX_HIST_COPIES 8
#endif
constexpr int HIST_COPIES = VX_HIST_COPIES;
constexpr int SW_CAR = 64, SW_MS = 65, SW_MB = 81, SW_HL = 82, SW_D = 90, SW_FLAGS = 94, SW_TN = 95, SW_WIN = 96, N_STAGE = 97;
// SW_CAR: 32 carries x 2 bits; SW_MS: the 16 message words in this row's order; SW_D: 4 words of two limbs;
// SW_FLAGS: ACT FIN FIRST CAP FA MDF0 MDF1 MDF3 (bits 0..7), INC (8..15), CNT (16..23), MK (24..31), E (32..39), SZ (40..63); SW_TN: T (low half),
// NUM (high half); SW_WIN: KOF (bits 0..23), TR (bit 32)
struct ExpandEntry {
    uint16_t col;
    uint8_t shift, bits;
};
__global__ __launch_bounds__(256) void k_blake_trace(const uint8_t* msgs, const BlockDesc* descs, const uint64_t* hchain, size_t n_real,
                                                     uint64_t* __restrict__ stage, uint32_t* __restrict__ hist, uint32_t* __restrict__ rc_part, size_t n,
                                                     uint32_t win_off, uint32_t win_len) {
    using namespace blk;
    // the (byte, 0) range-check lookups of every row fall on 256 counters = 8 cache lines: as global atomics they
    // serialise in L2 (the kernel took 7.5 ms for 0.4 GB of stores); they are counted per block in LDS instead and
    // summed by k_blake_mult.  n is a multiple of the block size (n >= 2^16).
    __shared__ uint32_t rc_lds[256];
    rc_lds[threadIdx.x] = 0;
    __syncthreads();
    const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t b = row >> 4;
    const int r = (int)(row & 15);
    const BlockDesc d = descs[b];
    uint64_t h[8], m[16], v[16];
    if (b < n_real) {
        for (int k = 0; k < 8; ++k) h[k] = hchain[8 * b + k];
        const uint64_t* p = (const uint64_t*)(msgs + d.msg_off);
        for (int k = 0; k < 16; ++k) {
            uint32_t bb = 8 * k;
            uint64_t w = 0;
            if (bb < d.inc) {
                w = p[k];
                if (d.inc - bb < 8) w &= (1ULL << (8 * (d.inc - bb))) - 1;
            }
            m[k] = w;
        }
    } else {  // padding block: the 40-byte message D || compact(number) || 0..
        for (int k = 0; k < 8; ++k) h[k] = IV[k];
        h[0] ^= 0x01010020ULL;
        for (int k = 0; k < 16; ++k) m[k] = k < 4 ? ((uint64_t)d.D[2 * k] | ((uint64_t)d.D[2 * k + 1] << 32)) : 0;
        // bytes 32..: SCALE compact of the last block number in its own mode (decoder.rs:39-92)
        m[4] = d.mode == 0 ? 4ULL * d.num : d.mode == 1 ? 4ULL * d.num + 1 : d.mode == 2 ? 4ULL * d.num + 2 : (3ULL | ((uint64_t)d.num << 8));
    }
    auto st = [&](int w) -> uint64_t& { return stage[(size_t)w * n + row]; };
    auto gw = [&](int k, int slot) -> uint64_t& { return st(8 * k + slot); };  // slot 7 = X (L / T)
    // HIST_COPIES private copies of the two histograms, picked by block index: consecutive blocks go to different XCDs (each
    // with its own L2), and counters shared by all of them bounce between the L2s; k_blake_mult adds the copies up
    uint32_t* const hc = hist + (size_t)(blockIdx.x % HIST_COPIES) * 131072;
    auto look1 = [&](uint64_t a, uint64_t bq) {  // 8 byte lookups (a_i, b_i, .) into T1
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&hc[((a >> (8 * j)) & 0xFF) | (((bq >> (8 * j)) & 0xFF) << 8)], 1u);
    };
    auto look2 = [&](uint64_t a, uint64_t bq) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(&hc[65536 + (((a >> (8 * j)) & 0xFF) | (((bq >> (8 * j)) & 0xFF) << 8))], 1u);
    };
    // ---- G area + carries
    blake_init_v(v, h, d.t, d.fin);
    uint64_t carw = 0;
    if (r >= 1 && r <= 12) {
        for (int q = 0; q + 1 < r; ++q) blake_round(v, m, q);
        const uint8_t* sg = ORDER[r];  // sigma of round r - 1
        auto G = [&](int k, uint64_t& va, uint64_t& vb, uint64_t& vc, uint64_t& vd, uint64_t x, uint64_t y) {
            const uint64_t a = va, bq = vb, c = vc, dd = vd;
            const uint64_t a1 = a + bq + x, d1 = b_rotr(dd ^ a1, 32), c1 = c + d1, b1 = b_rotr(bq ^ c1, 24);
            const uint64_t a2 = a1 + b1 + y, d2 = b_rotr(d1 ^ a2, 16), c2 = c1 + d2, b2 = b_rotr(b1 ^ c2, 63);
            gw(k, S_A1) = a1, gw(k, S_D1) = d1, gw(k, S_C1) = c1, gw(k, S_B1) = b1, gw(k, S_A2) = a2, gw(k, S_D2) = d2, gw(k, S_C2) = c2, gw(k, 7) = b1 ^ c2;
            uint8_t car[4];
            carries(a, bq, x, car);
            carries(a1, b1, y, car + 2);
            carw |= ((uint64_t)car[0] | ((uint64_t)car[1] << 2) | ((uint64_t)car[2] << 4) | ((uint64_t)car[3] << 6)) << (8 * k);
            // the row's lookups: (d, A1), (b, C1), (D1, A2) into T1 and (B1, C2) into T2
            look1(dd, a1), look1(bq, c1), look1(d1, a2), look2(b1, c2);
            va = a2, vb = b2, vc = c2, vd = d2;
        };
        G(0, v[0], v[4], v[8], v[12], m[sg[0]], m[sg[1]]);
        G(1, v[1], v[5], v[9], v[13], m[sg[2]], m[sg[3]]);
        G(2, v[2], v[6], v[10], v[14], m[sg[4]], m[sg[5]]);
        G(3, v[3], v[7], v[11], v[15], m[sg[6]], m[sg[7]]);
        G(4, v[0], v[5], v[10], v[15], m[sg[8]], m[sg[9]]);
        G(5, v[1], v[6], v[11], v[12], m[sg[10]], m[sg[11]]);
        G(6, v[2], v[7], v[8], v[13], m[sg[12]], m[sg[13]]);
        G(7, v[3], v[4], v[9], v[14], m[sg[14]], m[sg[15]]);
    } else {
        for (int w = 0; w < 64; ++w) st(w) = 0;  // (a lane's later store to the same address wins)
        if (r == 0) {
            for (int w = 0; w < 16; ++w) {
                const int mm = w & 3;
                if (w < 4) gw(4 + w, S_A2) = v[w];
                else if (w < 8) gw(4 + (mm + 3) % 4, 7) = b_rotr(v[w], 1);  // 2 L[j] + T[j-1] = byte j of v[w]
                else if (w < 12) gw(4 + (mm + 2) % 4, S_C2) = v[w];
                else gw(4 + (mm + 1) % 4, S_D2) = v[w];
            }
        } else {
            for (int q = 0; q < 12; ++q) blake_round(v, m, q);
            if (r <= 14)
                for (int w = 0; w < 8; ++w) {
                    const uint64_t u = v[w] ^ v[8 + w];
                    const uint64_t x = r == 13 ? v[w] : u, y = r == 13 ? v[8 + w] : h[w];
                    gw(w, S_D1) = x, gw(w, S_A2) = y, gw(w, S_D2) = b_rotr(x ^ y, 16);
                    look1(x, y);
                }
        }
    }
    st(SW_CAR) = carw;
    uint64_t h_out[8];
    if (r >= 13)
        for (int k = 0; k < 8; ++k) h_out[k] = h[k] ^ v[k] ^ v[k + 8];
    // ---- message schedule + bytes of natural word r (range checked as (byte, 0, byte) in T1)
    for (int s = 0; s < 16; ++s) st(SW_MS + s) = m[ORDER[r][s]];
    st(SW_MB) = m[r];
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&rc_lds[(m[r] >> (8 * j)) & 0xFF], 1u);
    // ---- H register, digest register
    for (int w = 0; w < 8; ++w) st(SW_HL + w) = r <= 13 ? h[w] : (r == 14 ? h_out[w] : (d.fin ? (w == 0 ? IV[0] ^ 0x01010020ULL : IV[w]) : h_out[w]));
    const bool cap = d.act && d.fin;
    for (int j = 0; j < 4; ++j) st(SW_D + j) = (r == 15 && cap) ? h_out[j] : ((uint64_t)d.D[2 * j] | ((uint64_t)d.D[2 * j + 1] << 32));
    // ---- flags, counters
    uint64_t mk = 0;
    for (int bq = 0; bq < 8; ++bq) mk |= (uint64_t)((uint32_t)(8 * r + bq) < d.inc ? 1 : 0) << bq;
    const uint64_t cnt = d.inc < (uint32_t)(8 * (r + 1)) ? d.inc : (uint32_t)(8 * (r + 1));
    // the row's window: rows 4..8 of a first chunk can hold state-root bytes (right behind the compact number, decoder.rs:121-128),
    // every other row data-root bytes (the last 32 bytes of the message, :132-149).  E[b]: byte 8r + b of this chunk lies in the
    // window's 32 bytes of an active message
    const bool srw = d.first && r >= 4 && r <= 8;
    // win_len > 0 (rotate, bus mode 2): outside those rows the window is bytes [win_off, win_off + win_len) of the message instead
    const uint32_t clen = d.mode == 0 ? 1 : d.mode == 1 ? 2 : d.mode == 2 ? 4 : 5, kof = srw ? 32 + clen : (win_len ? win_off : d.size - 32);
    const uint32_t wlen = (win_len && !srw) ? win_len : 32;
    uint64_t eb = 0;
    for (int bq = 0; bq < 8; ++bq) {
        const uint32_t pos = d.t - d.inc + 8 * r + bq;
        eb |= (uint64_t)((d.act && pos >= kof && pos < kof + wlen && (!win_len || (pos < d.size && !srw))) ? 1 : 0) << bq;
    }
    const uint64_t mdf = d.first ? (uint64_t)(d.mode == 0) | ((uint64_t)(d.mode == 1) << 1) | ((uint64_t)(d.mode == 3) << 2) : 0;
    st(SW_FLAGS) = (uint64_t)d.act | ((uint64_t)d.fin << 1) | ((uint64_t)d.first << 2) | ((uint64_t)(cap ? 1 : 0) << 3) | ((uint64_t)((d.first && d.act) ? 1 : 0) << 4) |
                   (mdf << 5) | ((uint64_t)d.inc << 8) | (cnt << 16) | (mk << 24) | (eb << 32) | ((uint64_t)d.size << 40);
    st(SW_TN) = (uint64_t)d.t | ((uint64_t)d.num << 32);
    st(SW_WIN) = (uint64_t)kof | ((uint64_t)(srw ? 0 : 1) << 32);
    __syncthreads();
    rc_part[(size_t)blockIdx.x * 256 + threadIdx.x] = rc_lds[threadIdx.x];
}
constexpr int EXP_RPL = 8;  // rows per lane: a block writes 8 x 2 KB = 16 KB of a column per visit
__global__ __launch_bounds__(256) void k_blake_expand(const uint64_t* __restrict__ stage, uint64_t* __restrict__ tr, size_t n, const ExpandEntry* __restrict__ ent,
                                                      const uint32_t* __restrict__ off) {
    const int w = blockIdx.y;
    const size_t row0 = (size_t)blockIdx.x * (256 * EXP_RPL) + threadIdx.x;
    uint64_t x[EXP_RPL];
#pragma unroll
    for (int rr = 0; rr < EXP_RPL; ++rr) {
        const size_t row = row0 + 256 * (size_t)rr;
        x[rr] = row < n ? stage[(size_t)w * n + row] : 0;
    }
    for (uint32_t e = off[w]; e < off[w + 1]; ++e) {
        const ExpandEntry E = ent[e];
        const uint64_t mask = E.bits == 64 ? ~0ULL : ((1ULL << E.bits) - 1);
        uint64_t* c = tr + (size_t)E.col * n;
#pragma unroll
        for (int rr = 0; rr < EXP_RPL; ++rr) {
            const size_t row = row0 + 256 * (size_t)rr;
            if (row < n) c[row] = (x[rr] >> E.shift) & mask;
        }
    }
}
// cell <- (packed word, shift, width) for every main column except the two multiplicities
static void blake_expand_table(std::vector<ExpandEntry>& ent, std::vector<uint32_t>& off) {
    using namespace blk;
    std::vector<std::vector<ExpandEntry>> per(N_STAGE);
    auto add = [&](int w, int col, int shift, int bits) { per[w].push_back(ExpandEntry{(uint16_t)col, (uint8_t)shift, (uint8_t)bits}); };
    for (int k = 0; k < 8; ++k) {
        for (int slot = 0; slot < 7; ++slot)
            for (int j = 0; j < 8; ++j) add(8 * k + slot, GC(k, slot, j), 8 * j, 8);
        for (int j = 0; j < 8; ++j) add(8 * k + 7, GC(k, S_L, j), 8 * j, 7), add(8 * k + 7, GC(k, S_T, j), 8 * j + 7, 1);
        for (int q = 0; q < 4; ++q) add(SW_CAR, CAR(k, q), 8 * k + 2 * q, 2);
    }
    for (int s = 0; s < 16; ++s) add(SW_MS + s, MS(s, 0), 0, 32), add(SW_MS + s, MS(s, 1), 32, 32);
    for (int j = 0; j < 8; ++j) add(SW_MB, MB0 + j, 8 * j, 8);
    for (int w = 0; w < 8; ++w) add(SW_HL + w, HL(w, 0), 0, 32), add(SW_HL + w, HL(w, 1), 32, 32);
    for (int j = 0; j < 4; ++j) add(SW_D + j, D0 + 2 * j, 0, 32), add(SW_D + j, D0 + 2 * j + 1, 32, 32);
    add(SW_FLAGS, ACT, 0, 1), add(SW_FLAGS, FIN, 1, 1), add(SW_FLAGS, FIRST, 2, 1), add(SW_FLAGS, CAP, 3, 1), add(SW_FLAGS, FA, 4, 1);
    add(SW_FLAGS, INC, 8, 8), add(SW_FLAGS, CNT, 16, 8);
    for (int i = 0; i < 8; ++i) add(SW_FLAGS, IB0 + i, 8 + i, 1), add(SW_FLAGS, MK0 + i, 24 + i, 1), add(SW_FLAGS, E0 + i, 32 + i, 1);
    add(SW_FLAGS, SZ, 40, 24);
    add(SW_FLAGS, MDF0, 5, 1), add(SW_FLAGS, MDF1, 6, 1), add(SW_FLAGS, MDF3, 7, 1);
    add(SW_WIN, KOF, 0, 24), add(SW_WIN, TR, 32, 1);
    add(SW_TN, T, 0, 32), add(SW_TN, NUM, 32, 32);
    for (int i = 0; i < 32; ++i) add(SW_TN, TB0 + i, i, 1);
    off.assign(1, 0);
    for (auto& v : per) {
        ent.insert(ent.end(), v.begin(), v.end());
        off.push_back((uint32_t)ent.size());
    }
}
// the multiplicity columns: all counts in the first copy of the periodic tables
__global__ __launch_bounds__(256) void k_blake_mult(const uint32_t* hist, const uint32_t* rc_part, uint64_t* tr, size_t n) {
    const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (row >= n) return;
    uint64_t m1 = 0, m2 = 0;
    if (row < 65536)
        for (int cpy = 0; cpy < HIST_COPIES; ++cpy) m1 += hist[(size_t)cpy * 131072 + row], m2 += hist[(size_t)cpy * 131072 + 65536 + row];
    if (row < 256)  // table rows (a, b = 0): plus the per-block range-check counts
        for (size_t b = 0; b < n / 256; ++b) m1 += rc_part[b * 256 + row];
    tr[(size_t)blk::M1 * n + row] = m1;
    tr[(size_t)blk::M2 * n + row] = m2;
}

// ---- auxiliary columns (logUp) --------------------------------------------------------------------------------------
// k_blake_aux: one lane per (row i, unit u); i is the "next" row of the pair (i-1, i).  Units 0..7 = the 16 helper
// elements of G number u (two lookups each, h = m (1/D_u + 1/D_v); the four pairs of a group share ONE extension-field
// inversion, Montgomery's trick); unit 8 = the 4 message-byte range-check helpers and the table helper of row i.  Every
// unit leaves the sum of its helpers in part[u]; k_blake_aux_z adds them up into the running-sum increment
// Z(i) - Z(i-1) = sum_e h_e(i) - ht(i-1), stored at row i-1 and turned into Z by an exclusive scan.
struct AuxArgs {
    const uint64_t* tr;
    uint64_t* aux;
    uint64_t* part;  // [9][2][n]
    size_t n;
    gl2 beta, gamma;
    uint64_t first_number, tree_size, bus_on;  // the block number of leaf 0 (public input 18 in bus mode 1, else 16), public inputs 18, 19
};
#ifndef VX_AUX_WAVES
#define VX_AUX_WAVES 4  // measured: 2 -> 6.7 ms, 3 -> 6.4, 4 -> 4.9 (64 VGPRs + scratch: the kernel lives on occupancy hiding its dependent multiply chains)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VX_AUX_WAVES, VX_AUX_WAVES))) void k_blake_aux(AuxArgs a) {
    using namespace blk;
    const size_t n = a.n, i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const int unit = blockIdx.y;
    if (i >= n) return;
    const size_t ip = (i + n - 1) & (n - 1);
    const int rl = (int)(ip & 15);  // row index (mod 16) of the LOCAL row: selectors are taken there
    const bool g_on = rl <= 11, m3 = rl <= 13;
    const gl2 beta = a.beta, gamma = a.gamma, g2 = gl2_mul(gamma, gamma);
    auto N = [&](int col) -> uint64_t { return a.tr[(size_t)col * n + i]; };
    auto L = [&](int col) -> uint64_t { return a.tr[(size_t)col * n + ip]; };
    auto fp1 = [&](uint64_t x, uint64_t y, uint64_t z) -> gl2 {
        gl2 d = gl2_add(beta, gl2_add(gl2_scale(gamma, y), gl2_scale(g2, z)));
        d.a = gl_add(d.a, x);
        return d;
    };
    auto store = [&](int e, gl2 h) {
        a.aux[(size_t)(2 * e) * n + i] = h.a;
        a.aux[(size_t)(2 * e + 1) * n + i] = h.b;
    };
    gl2 hsum{0, 0};
    // four pairs (p_q = D_u D_v, s_q = D_u + D_v) -> h_q = s_q / p_q with one inversion
    auto four = [&](gl2* p, gl2* s, int e0) {
        const gl2 c1 = gl2_mul(p[0], p[1]), c2 = gl2_mul(c1, p[2]), c3 = gl2_mul(c2, p[3]);
        gl2 inv = gl2_inv(c3);
        const gl2 i3 = gl2_mul(inv, c2);
        inv = gl2_mul(inv, p[3]);
        const gl2 i2 = gl2_mul(inv, c1);
        inv = gl2_mul(inv, p[2]);
        const gl2 i1 = gl2_mul(inv, p[0]), i0 = gl2_mul(inv, p[1]);
        const gl2 iq[4] = {i0, i1, i2, i3};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const gl2 h = gl2_mul(s[q], iq[q]);
            store(e0 + q, h);
            hsum = gl2_add(hsum, h);
        }
    };
    if (unit < 8) {
        const int k = unit;
        const gl2 g3 = gl2_mul(g2, gamma), bt2 = gl2_add(beta, gl2_mul(g2, g2));
        auto out_byte_loc = [&](int w, int j) -> uint64_t {
            const int m = w & 3;
            if (w < 8) {  // only b (w = 4..7) and d (w = 12..15) operands enter a lookup
                const int kk = 4 + (m + 3) % 4;
                return 2 * L(GC(kk, S_L, j)) + L(GC(kk, S_T, (j + 7) & 7));
            }
            return L(GC(4 + (m + 1) % 4, S_D2, j));
        };
        auto in_byte = [&](int op, int j) -> uint64_t {
            if (k < 4) return out_byte_loc(4 * op + k, j);
            const int j0 = k - 4;
            if (op == 1) {
                const int kb = (j0 + 1) & 3;
                return 2 * N(GC(kb, S_L, j)) + N(GC(kb, S_T, (j + 7) & 7));
            }
            return N(GC((j0 + 3) & 3, S_D2, j));
        };
        auto denoms = [&](int grp, gl2* dd) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (grp == 0) dd[q] = fp1(in_byte(3, q), N(GC(k, S_A1, q)), N(GC(k, S_D1, (q + 4) & 7)));
                else if (grp == 1) dd[q] = fp1(in_byte(1, q), N(GC(k, S_C1, q)), N(GC(k, S_B1, (q + 5) & 7)));
                else if (grp == 2) dd[q] = fp1(N(GC(k, S_D1, q)), N(GC(k, S_A2, q)), N(GC(k, S_D2, (q + 6) & 7)));
                else {
                    gl2 d = gl2_add(bt2, gl2_add(gl2_scale(gamma, N(GC(k, S_C2, q))), gl2_add(gl2_scale(g2, N(GC(k, S_L, q))), gl2_scale(g3, N(GC(k, S_T, q))))));
                    d.a = gl_add(d.a, N(GC(k, S_B1, q)));
                    dd[q] = d;
                }
            }
        };
        // ONE extension-field inversion per lane (Montgomery's trick on two levels): pass 1 multiplies the 8 denominators
        // of every active group into c[grp]; the four c are inverted together; pass 2 recomputes the denominators (the
        // cells come back from cache) and unwinds each group with its 1 / c[grp].
        gl2 cg[4], tot{1, 0};
#pragma unroll 1
        for (int grp = 0; grp < 4; ++grp) {
            cg[grp] = gl2{1, 0};
            if (grp == 2 ? m3 : g_on) {
                gl2 dd[8];
                denoms(grp, dd);
                gl2 c = gl2_mul(dd[0], dd[1]);
#pragma unroll
                for (int q = 2; q < 8; ++q) c = gl2_mul(c, dd[q]);
                cg[grp] = c;
            }
        }
        // prefix products of cg, one inversion, suffix unwinding -> icg[grp] = 1 / cg[grp]
        const gl2 p01 = gl2_mul(cg[0], cg[1]), p012 = gl2_mul(p01, cg[2]);
        tot = gl2_mul(p012, cg[3]);
        gl2 inv = m3 ? gl2_inv(tot) : gl2{1, 0};
        gl2 icg[4];
        icg[3] = gl2_mul(inv, p012), inv = gl2_mul(inv, cg[3]);
        icg[2] = gl2_mul(inv, p01), inv = gl2_mul(inv, cg[2]);
        icg[1] = gl2_mul(inv, cg[0]), icg[0] = gl2_mul(inv, cg[1]);
#pragma unroll 1
        for (int grp = 0; grp < 4; ++grp) {
            const int e0 = (k * 4 + grp) * 4;
            if (grp == 2 ? m3 : g_on) {
                gl2 dd[8], p[4];
                denoms(grp, dd);
#pragma unroll
                for (int q = 0; q < 4; ++q) p[q] = gl2_mul(dd[2 * q], dd[2 * q + 1]);
                // 1 / p[q] = icg * (product of the other three pair products)
                const gl2 p01_ = gl2_mul(p[0], p[1]), p23_ = gl2_mul(p[2], p[3]);
                const gl2 a01 = gl2_mul(icg[grp], p23_), a23 = gl2_mul(icg[grp], p01_);  // 1 / (p0 p1), 1 / (p2 p3)
                const gl2 ip[4] = {gl2_mul(a01, p[1]), gl2_mul(a01, p[0]), gl2_mul(a23, p[3]), gl2_mul(a23, p[2])};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const gl2 h = gl2_mul(gl2_add(dd[2 * q], dd[2 * q + 1]), ip[q]);
                    store(e0 + q, h);
                    hsum = gl2_add(hsum, h);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) store(e0 + q, gl2{0, 0});
            }
        }
    } else {
        gl2 p[4], s[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t b0 = N(MB0 + 2 * q), b1 = N(MB0 + 2 * q + 1);
            const gl2 du = fp1(b0, 0, b0), dv = fp1(b1, 0, b1);
            p[q] = gl2_mul(du, dv), s[q] = gl2_add(du, dv);
        }
        four(p, s, HM0);
        // bus sends of this row (it is the "next" row of its pair; r = its index in the block): byte b under E[b] as (leaf, position
        // in the root, byte, tree) -- state root (tree 0) on rows 4..8 of a first chunk, data root (tree 1) elsewhere
        {
            const gl2 g3 = gl2_mul(g2, gamma), g4 = gl2_mul(g2, g2);
            const int r = (int)(i & 15);
            const uint64_t leaf = gl_sub(N(NUM), a.first_number);
            const uint64_t pos0 = gl_sub(gl_add(gl_sub(N(T), N(INC)), (uint64_t)(8 * r)), N(KOF));
            const gl2 bbase = gl2_add(beta, gl2_add(gl2{leaf, 0}, gl2_add(gl2_scale(g3, N(TR)), gl2_scale(g4, TAG_BYTE))));
#pragma unroll 1
            for (int pair = 0; pair < 4; ++pair) {
                gl2 h{0, 0};
                const bool live = a.bus_on && N(ACT);
                const uint64_t e0 = live ? N(E0 + 2 * pair) : 0, e1 = live ? N(E0 + 2 * pair + 1) : 0;
                if (e0 | e1) {
                    const gl2 du = gl2_add(bbase, gl2_add(gl2_scale(gamma, gl_add(pos0, 2 * pair)), gl2_scale(g2, N(MB0 + 2 * pair))));
                    const gl2 dv = gl2_add(bbase, gl2_add(gl2_scale(gamma, gl_add(pos0, 2 * pair + 1)), gl2_scale(g2, N(MB0 + 2 * pair + 1))));
                    h = gl2_mul(gl2_add(gl2_scale(dv, e0), gl2_scale(du, e1)), gl2_inv(gl2_mul(du, dv)));
                }
                store(HB0 + pair, h);
                hsum = gl2_add(hsum, h);
            }
        }
        // table helper of this row: ht = M1 / D_t1 + M2 / D_t2
        const uint64_t m1 = N(M1), m2 = N(M2);
        gl2 ht{0, 0};
        if (m1 | m2) {
            const gl2 g3 = gl2_mul(g2, gamma), bt2 = gl2_add(beta, gl2_mul(g2, g2));
            const uint64_t ti = i & 65535, ta = ti & 255, tb = ti >> 8, x = ta ^ tb;
            const gl2 d1 = fp1(ta, tb, x);
            gl2 d2 = gl2_add(bt2, gl2_add(gl2_scale(gamma, tb), gl2_add(gl2_scale(g2, x & 127), gl2_scale(g3, x >> 7))));
            d2.a = gl_add(d2.a, ta);
            ht = gl2_mul(gl2_add(gl2_scale(d2, m1), gl2_scale(d1, m2)), gl2_inv(gl2_mul(d1, d2)));
        }
        store(HT, ht);
    }
    a.part[(size_t)(2 * unit) * n + i] = hsum.a;
    a.part[(size_t)(2 * unit + 1) * n + i] = hsum.b;
}
__global__ __launch_bounds__(256) void k_blake_aux_z(const uint64_t* part, uint64_t* aux, size_t n) {
    using namespace blk;
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t ip = (i + n - 1) & (n - 1);
    uint64_t sa = 0, sb = 0;
#pragma unroll
    for (int u = 0; u < 9; ++u) sa = gl_add(sa, part[(size_t)(2 * u) * n + i]), sb = gl_add(sb, part[(size_t)(2 * u + 1) * n + i]);
    aux[(size_t)(2 * ZZ) * n + ip] = gl_sub(sa, aux[(size_t)(2 * HT) * n + ip]);
    aux[(size_t)(2 * ZZ + 1) * n + ip] = gl_sub(sb, aux[(size_t)(2 * HT + 1) * n + ip]);
}

// Z(i) = (exclusive prefix sum of the increments)(i) - i * S / n: with the published S / n subtracted from every increment
// the running sum closes cyclically
__global__ __launch_bounds__(256) void k_bus_close(uint64_t* za, uint64_t* zb, size_t n, uint64_t spa, uint64_t spb) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    za[i] = gl_sub(za[i], gl_mul(spa, (uint64_t)i));
    zb[i] = gl_sub(zb[i], gl_mul(spb, (uint64_t)i));
}
int32_t vx_bus_close_dev(vx_ctx* ctx, uint64_t* z_cols, int log_n, uint64_t aux_pub[2]) {
    const size_t n = (size_t)1 << log_n;
    uint64_t tot[2];
    VX_TRY(vx_scan_cols_dev(ctx, z_cols, log_n, 2, tot));
    const uint64_t ninv = glh::inv(n % glh::P);
    aux_pub[0] = glh::mul(tot[0], ninv), aux_pub[1] = glh::mul(tot[1], ninv);
    hipLaunchKernelGGL(k_bus_close, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, z_cols, z_cols + n, n, aux_pub[0], aux_pub[1]);
    VX_HIP(hipGetLastError());
    return VX_OK;
}

int32_t vx_blake_air_gen_aux(vx_ctx* ctx, const uint64_t* trace, int log_n, const uint64_t* chal, const uint64_t* pub, uint64_t* aux, uint64_t* aux_pub) {
    const size_t n = (size_t)1 << log_n;
    uint64_t* part = (uint64_t*)vx_pool_alloc(ctx, 18 * n * 8);
    if (!part) return vx_fail(ctx, VX_ERR_OOM, "blake aux: out of device memory");
    AuxArgs a{trace, aux, part, n, gl2{chal[0], chal[1]}, gl2{chal[2], chal[3]}, pub[19] == 1 ? pub[18] : pub[16], pub[18], pub[19]};  // bus mode 1 counts leaves from the range's first block
    hipLaunchKernelGGL(k_blake_aux, dim3((unsigned)((n + 255) / 256), 9), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_blake_aux_z, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint64_t*)part, aux, n);
    const hipError_t e = hipGetLastError();
    vx_pool_free(ctx, part);  // recycled only by later work on the same stream
    if (e != hipSuccess) return vx_fail(ctx, VX_ERR_DEVICE, "blake aux: %s", hipGetErrorString(e));
    return vx_bus_close_dev(ctx, aux + (size_t)(2 * blk::ZZ) * n, log_n, aux_pub);
}

extern "C" {

int32_t vx_blake_chain_trace(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes, size_t n_headers,
                             const uint8_t trusted_hash[32], uint32_t first_block_number, uint32_t tree_size, uint32_t window_offset, uint32_t window_length,
                             int log_n, vx_buf* trace_out, uint64_t public_inputs_out[20], uint8_t* digests_out) {
    if (!ctx || !headers || !sizes || !trusted_hash || !trace_out || !public_inputs_out) return VX_ERR_ARG;
    VX_CHECK(stride % 128 == 0 && stride > 0, "blake trace: stride %zu must be a positive multiple of 128", stride);
    VX_CHECK(n_headers >= 1 && n_headers * stride <= headers->n * 8, "blake trace: headers exceed the buffer");
    VX_CHECK(log_n >= blk::TABLE_LOG && log_n <= 24, "blake trace: log_n %d out of range [16, 24] (the trace holds one copy of the 2^16-row lookup tables)", log_n);
    VX_CHECK((uint64_t)first_block_number + n_headers <= (1ULL << 32), "blake trace: block numbers %u.. overflow 32 bits", first_block_number);
    VX_CHECK(!window_length || (n_headers == 1 && !tree_size && window_offset >= 72 && (uint64_t)window_offset + window_length < (1u << 24)),
             "blake trace: a byte window (offset %u, length %u) goes with one header, no Merkle tree, and starts behind the state root", window_offset, window_length);
    VX_CHECK(window_length || window_offset == 0 || (tree_size && window_offset < first_block_number), "blake trace: a leaf offset (%u) goes with tree_size != 0", window_offset);
    const size_t n = (size_t)1 << log_n, n_blocks = n >> 4;
    VX_CHECK(trace_out->n >= n * blk::COLS, "blake trace: trace buffer holds %zu < %zu elements", trace_out->n, n * (size_t)blk::COLS);
    std::vector<uint32_t> base(n_headers);
    size_t n_real = 0;
    for (size_t i = 0; i < n_headers; ++i) {
        const uint32_t clen = (uint32_t[]){1, 2, 4, 5}[compact_mode(first_block_number + (uint32_t)i)];  // parent hash + compact number must fit
        VX_CHECK(sizes[i] <= stride && sizes[i] >= 32 + clen && sizes[i] < (1u << 24), "blake trace: header %zu has size %u", i, sizes[i]);
        VX_CHECK(!tree_size || sizes[i] >= 104,
                 "blake trace: header %zu has size %u (shorter than 104 bytes its state root and data root would share trace rows; no Avail header is)", i, sizes[i]);
        base[i] = (uint32_t)n_real;
        n_real += (sizes[i] + 127) / 128;
    }
    VX_CHECK(n_real <= n_blocks, "blake trace: %zu compressions do not fit 2^%d rows (%zu blocks)", n_real, log_n, n_blocks);
    // device scratch: sizes | block_base | digests | hchain | descs
    const size_t w_sizes = (n_headers * 4 + 7) / 8, w_dig = n_headers * 4, w_hc = n_real * 8;
    const size_t w_desc = (n_blocks * sizeof(BlockDesc) + 7) / 8;
    const size_t w_hist = (size_t)HIST_COPIES * 65536 + n / 2;  // HIST_COPIES x two tables x 2^16 uint32 counters + 256 range-check counters per 256-row block
    std::vector<ExpandEntry> ent;
    std::vector<uint32_t> eoff;
    blake_expand_table(ent, eoff);
    VX_CHECK(ent.size() == (size_t)blk::COLS - 2, "blake trace: expansion table covers %zu of %d columns", ent.size(), blk::COLS - 2);
    const size_t w_ent = (ent.size() * sizeof(ExpandEntry) + 7) / 8, w_eoff = (eoff.size() * 4 + 7) / 8, w_stage = (size_t)N_STAGE * n;
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, 2 * w_sizes + w_dig + w_hc + w_desc + w_hist + w_ent + w_eoff + w_stage, &sc));
    uint32_t* d_sizes = (uint32_t*)sc;
    uint32_t* d_base = (uint32_t*)(sc + w_sizes);
    uint8_t* d_dig = (uint8_t*)(sc + 2 * w_sizes);
    uint64_t* d_hc = sc + 2 * w_sizes + w_dig;
    BlockDesc* d_desc = (BlockDesc*)(d_hc + w_hc);
    uint32_t* d_hist = (uint32_t*)(d_hc + w_hc + w_desc);
    ExpandEntry* d_ent = (ExpandEntry*)(d_hc + w_hc + w_desc + w_hist);
    uint32_t* d_eoff = (uint32_t*)(d_hc + w_hc + w_desc + w_hist + w_ent);
    uint64_t* d_stage = d_hc + w_hc + w_desc + w_hist + w_ent + w_eoff;
    VX_HIP(hipMemcpyAsync(d_ent, ent.data(), ent.size() * sizeof(ExpandEntry), hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemcpyAsync(d_eoff, eoff.data(), eoff.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemcpyAsync(d_sizes, sizes, n_headers * 4, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemcpyAsync(d_base, base.data(), n_headers * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_blake_chain, dim3((unsigned)((n_headers + 63) / 64)), dim3(64), 0, ctx->stream, (const uint8_t*)headers->d,
                       stride, (const uint32_t*)d_sizes, n_headers, (const uint32_t*)d_base, d_hc, d_dig);
    VX_HIP(hipGetLastError());
    std::vector<uint8_t> dig(32 * n_headers);
    VX_HIP(hipMemcpyAsync(dig.data(), d_dig, dig.size(), hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    // block descriptors (host): the link rule is checked here too, so a broken chain fails loudly
    std::vector<BlockDesc> descs(n_blocks);
    uint8_t D[32];
    memcpy(D, trusted_hash, 32);
    size_t bi = 0;
    for (size_t i = 0; i < n_headers; ++i) {
        const uint32_t nch = (sizes[i] + 127) / 128;
        for (uint32_t cidx = 0; cidx < nch; ++cidx, ++bi) {
            BlockDesc& d = descs[bi];
            memset(&d, 0, sizeof d);
            d.fin = cidx + 1 == nch;
            d.first = cidx == 0;
            d.act = 1;
            d.inc = d.fin ? sizes[i] - 128 * cidx : 128;
            d.t = d.fin ? sizes[i] : 128 * (cidx + 1);
            d.msg_off = i * stride + 128 * (size_t)cidx;
            d.num = first_block_number + (uint32_t)i;
            d.mode = compact_mode(d.num);
            d.size = sizes[i];
            memcpy(d.D, D, 32);
        }
        memcpy(D, dig.data() + 32 * i, 32);
    }
    for (; bi < n_blocks; ++bi) {
        BlockDesc& d = descs[bi];
        memset(&d, 0, sizeof d);
        d.fin = d.first = 1;
        d.act = 0;
        d.inc = d.t = d.size = 40;
        d.msg_off = ~0ULL;
        d.num = first_block_number + (uint32_t)n_headers - 1;
        d.mode = compact_mode(d.num);
        memcpy(d.D, D, 32);
    }
    VX_HIP(hipMemcpyAsync(d_desc, descs.data(), n_blocks * sizeof(BlockDesc), hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipMemsetAsync(d_hist, 0, (size_t)HIST_COPIES * 65536 * 8, ctx->stream));
    hipLaunchKernelGGL(k_blake_trace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint8_t*)headers->d,
                       (const BlockDesc*)d_desc, (const uint64_t*)d_hc, n_real, d_stage, d_hist, d_hist + (size_t)HIST_COPIES * 131072, n, window_offset, window_length);
    VX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_blake_expand, dim3((unsigned)((n + 256 * EXP_RPL - 1) / (256 * EXP_RPL)), N_STAGE), dim3(256), 0, ctx->stream,
                       (const uint64_t*)d_stage, trace_out->d, n, (const ExpandEntry*)d_ent, (const uint32_t*)d_eoff);
    VX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_blake_mult, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_hist, (const uint32_t*)(d_hist + (size_t)HIST_COPIES * 131072), trace_out->d, n);
    VX_HIP(hipGetLastError());
    VX_HIP(hipStreamSynchronize(ctx->stream));  // descs must outlive the kernel
    for (int j = 0; j < 8; ++j) {
        uint32_t a, b;
        memcpy(&a, trusted_hash + 4 * j, 4);
        memcpy(&b, D + 4 * j, 4);
        public_inputs_out[j] = a;
        public_inputs_out[8 + j] = b;
    }
    public_inputs_out[16] = first_block_number;
    public_inputs_out[17] = first_block_number + (uint64_t)n_headers - 1;
    // bus mode 1: the block number of leaf 0 = this table's first block minus the headers of the range before it (window_offset
    // doubles as that count when window_length = 0: a map segment of a longer range; 0 = the table starts the range)
    public_inputs_out[18] = window_length ? window_offset : (tree_size ? (uint64_t)first_block_number - window_offset : 0);
    public_inputs_out[19] = window_length ? 2 : (tree_size ? 1 : 0);  // bus mode: 1 the state / data roots go to a Merkle AIR, 2 a window of message bytes
    if (digests_out) memcpy(digests_out, dig.data(), dig.size());
    return VX_OK;
}

static const uint64_t VX_HR_MAGIC = VX_HR_BLOB_MAGIC;  // "HRRANGE5" (include/vx.h)
// magic, max_headers, trusted, target, out96 (12), proof lengths: hash chain, authority-set commitment, Merkle, Ed25519, SHA-512; the precommit's round
static const size_t VX_HR_HDR = VX_HR_BLOB_FIXED_WORDS + 1;  // one segment

static int sha_log_n(size_t n_keys) {
    int log_n = 6;
    while (((size_t)1 << log_n) < 64 * (2 * n_keys - 1)) ++log_n;
    return log_n;
}
static int tree_air_id(uint32_t max_headers) { return max_headers == 256 ? 7 : max_headers == 512 ? 8 : max_headers == 16 ? 9 : 0; }
static int tree_log_n(uint32_t max_headers) {
    int l = 8;
    while ((1u << (l - 8)) < max_headers) ++l;
    return l;
}
// the EdDSA tables by the number of signatures they verify: 256 rows per signature (one slot stays idle) / 164 rows per hash.
// The prover needs floor(2n/3) + 1 of the n authorities (justification.rs:164-186), so it verifies exactly that many.
static size_t sig_quorum(size_t n_auth) { return 2 * n_auth / 3 + 1; }
static int ed_log_n(size_t n_sig) { return n_sig <= 255 ? 16 : 17; }
static int ed_air_id(size_t n_sig) { return n_sig <= 255 ? VX_AIR_ED25519_16 : VX_AIR_ED25519; }
static int s512_log_n(size_t n_sig) { return n_sig <= 6 ? 10 : n_sig <= 204 ? 15 : 16; }
static int s512_air_id(size_t n_sig) { return n_sig <= 6 ? VX_AIR_SHA512_10 : n_sig <= 204 ? VX_AIR_SHA512_15 : VX_AIR_SHA512; }

int32_t vx_header_range_proof_bound(const vx_stark_config* cfg, size_t n_chunks, size_t n_authorities, size_t* n_words) {
    if (!cfg || !n_words || n_chunks == 0) return VX_ERR_ARG;
    int log_n = blk::TABLE_LOG;
    while (((size_t)1 << log_n) < 16 * n_chunks) ++log_n;
    size_t w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0;
    int32_t rc = vx_stark_proof_bound(VX_AIR_BLAKE_CHAIN, cfg, log_n, &w1);
    if (rc == VX_OK && n_authorities) rc = vx_stark_proof_bound(VX_AIR_SHA_CHAIN, cfg, sha_log_n(n_authorities), &w2);
    if (rc == VX_OK) rc = vx_stark_proof_bound(8, cfg, tree_log_n(512), &w3);  // the largest Merkle AIR (the request's max_headers is not known here)
    if (rc == VX_OK && n_authorities) rc = vx_stark_proof_bound(ed_air_id(sig_quorum(n_authorities)), cfg, ed_log_n(sig_quorum(n_authorities)), &w4);
    if (rc == VX_OK && n_authorities) rc = vx_stark_proof_bound(s512_air_id(sig_quorum(n_authorities)), cfg, s512_log_n(sig_quorum(n_authorities)), &w5);
    *n_words = w1 + w2 + w3 + w4 + w5 + VX_HR_HDR;
    return rc;
}

}  // extern "C"

void BusMeet::finish_locked(size_t cap_words) {
    if (done) return;
    if (xch && xch->fn) {
        // slot t: [state (0 absent, 1 present, 2 failed), n_pub, pub[MAX_PUB], cap[cap_words]]; the shards' arrays are summed
        const size_t slot = 2 + MAX_PUB + cap_words;
        std::vector<uint64_t> w((size_t)n_parties * slot, 0);
        for (int t = 0; t < n_parties; ++t)
            if (local[t]) {
                uint64_t* q = w.data() + (size_t)t * slot;
                if (deposited[t] && pub[t].size() <= (size_t)MAX_PUB && cap[t].size() == cap_words) {
                    q[0] = 1, q[1] = pub[t].size();
                    memcpy(q + 2, pub[t].data(), pub[t].size() * 8);
                    memcpy(q + 2 + MAX_PUB, cap[t].data(), cap_words * 8);
                } else q[0] = 2;
            }
        if (xch->fn(xch->user, w.data(), w.size()) != 0) failed = true;
        for (int t = 0; t < n_parties && !failed; ++t) {
            const uint64_t* q = w.data() + (size_t)t * slot;
            if (q[0] != 1 || q[1] > (uint64_t)MAX_PUB) failed = true;  // a table nobody proved, one proved twice, or one whose prover gave up
            else if (!local[t]) pub[t].assign(q + 2, q + 2 + q[1]), cap[t].assign(q + 2 + MAX_PUB, q + 2 + MAX_PUB + cap_words);
        }
    }
    done = true;
    cv.notify_all();
}
int32_t BusMeet::meet(BusMeet* r, int who, const uint64_t* pub, size_t n_pub, const uint64_t* cap, size_t cap_words, uint64_t* chal, size_t n_chal) {
    std::unique_lock<std::mutex> lk(r->m);
    r->pub[who].assign(pub, pub + n_pub);
    r->cap[who].assign(cap, cap + cap_words);
    r->deposited[who] = true;
    ++r->arrived;
    r->capw = cap_words;
    if (r->arrived == r->n_local()) r->finish_locked(cap_words);
    r->cv.wait(lk, [&] { return r->done || (r->failed && !r->xch); });
    if (r->failed) return VX_ERR_STATEMENT;  // another table's prover gave up
    const uint64_t *pubs[MAX], *caps[MAX];
    size_t ns[MAX];
    for (int t = 0; t < r->n_parties; ++t) pubs[t] = r->pub[t].data(), ns[t] = r->pub[t].size(), caps[t] = r->cap[t].data();
    uint64_t c[4];
    vx_shared_challenges_n(pubs, ns, caps, (size_t)r->n_parties, cap_words, c, 4);
    for (size_t q = 0; q < n_chal && q < 4; ++q) chal[q] = c[q];
    return VX_OK;
}
void BusMeet::fail(int who) {
    std::lock_guard<std::mutex> lk(m);
    failed = true;
    if (xch && who >= 0 && who < n_parties && local[who] && !deposited[who] && !done) {
        // a sharded proof: the other shards are (or will be) inside the exchange -- this table arrives as a failure marker
        deposited[who] = false;
        ++arrived;
        local_failed[who] = true;
        if (arrived == n_local()) finish_locked(capw);
    }
    cv.notify_all();
}
int32_t vx_bus_hook(void* u, const uint64_t* pub, size_t n_pub, const uint64_t* cap, size_t cw, uint64_t* chal, size_t n_chal) {
    BusParty* p = (BusParty*)u;
    return BusMeet::meet(p->rv, p->who, pub, n_pub, cap, cw, chal, n_chal);
}

size_t vx_justification_proof_bound(const vx_stark_config* cfg, size_t n_authorities, int32_t* rc_out) {
    size_t w2 = 0, w4 = 0, w5 = 0;
    int32_t rc = vx_stark_proof_bound(VX_AIR_SHA_CHAIN, cfg, sha_log_n(n_authorities), &w2);
    if (rc == VX_OK) rc = vx_stark_proof_bound(ed_air_id(sig_quorum(n_authorities)), cfg, ed_log_n(sig_quorum(n_authorities)), &w4);
    if (rc == VX_OK) rc = vx_stark_proof_bound(s512_air_id(sig_quorum(n_authorities)), cfg, s512_log_n(sig_quorum(n_authorities)), &w5);
    *rc_out = rc;
    return w2 + w4 + w5;
}

int32_t vx_justification_tables_start(vx_ctx* const ctxs[3], const vx_justification* just, const vx_stark_config* cfg, BusMeet* rv, int first,
                                      int32_t (*pre)(vx_ctx*, void*), void* pre_user, JustificationTables* jt, unsigned mask) {
    // the signatures the proof verifies: the first floor(2n/3) + 1 signed authorities (more would only cost rows); when fewer
    // signed, all of them -- the native threshold check refuses the justification before anything is proven
    jt->chosen.assign(just->num_authorities, 0);
    jt->n_sig = 0;
    for (size_t i = 0; i < just->num_authorities && jt->n_sig < sig_quorum(just->num_authorities); ++i)
        if (just->validator_signed[i]) jt->chosen[i] = 1, ++jt->n_sig;
    for (int t = 0; t < 3; ++t) jt->party[t] = {rv, first + t}, jt->hooks[t] = {vx_bus_hook, &jt->party[t]}, jt->job[t].c = ctxs[t];
    auto prove_chain = [=](vx_ctx* c, TableJob& j) -> int32_t {
        if (pre) VX_TRY(pre(c, pre_user));
        const int sl = sha_log_n(just->num_authorities);
        size_t bound = 0;
        VX_TRY(vx_stark_proof_bound(VX_AIR_SHA_CHAIN, cfg, sl, &bound));
        j.proof.resize(bound);
        vx_buf* st = nullptr;
        VX_TRY(vx_alloc(c, ((size_t)VX_SHA_AIR_COLS) << sl, &st));
        uint64_t spub[10];
        uint8_t com[32];
        int32_t r = vx_sha_chain_trace_dev(c, just->pubkeys, just->num_authorities, jt->chosen.data(), 1, sl, st->d, spub, com);
        if (r == VX_OK && memcmp(com, just->authority_set_hash, 32) != 0) r = vx_fail(c, VX_ERR_STATEMENT, "authority-set commitment mismatch");
        if (r == VX_OK) r = vx_stark_prove_impl(c, VX_AIR_SHA_CHAIN, cfg, st->d, st->n, /*consume_trace=*/0, sl, spub, 10, j.proof.data(), j.proof.size(), &j.len, &jt->hooks[0]);
        (void)vx_free(c, st);
        return r;
    };
    auto prove_ed = [=](vx_ctx* c, TableJob& j) -> int32_t {
        const int el = ed_log_n(jt->n_sig), id = ed_air_id(jt->n_sig);
        size_t bound = 0;
        VX_TRY(vx_stark_proof_bound(id, cfg, el, &bound));
        j.proof.resize(bound);
        vx_buf* et = nullptr;
        VX_TRY(vx_alloc(c, ((size_t)VX_ED_AIR_COLS) << el, &et));
        uint64_t epub[2];
        int32_t r = vx_ed_trace_dev(c, just->pubkeys, just->signatures, just->precommit, 53, jt->chosen.data(), just->num_authorities, el, 1, et->d, epub);
        if (r == VX_OK) r = vx_stark_prove_impl(c, id, cfg, et->d, et->n, /*consume_trace=*/0, el, epub, 2, j.proof.data(), j.proof.size(), &j.len, &jt->hooks[1]);
        (void)vx_free(c, et);
        return r;
    };
    auto prove_s512 = [=](vx_ctx* c, TableJob& j) -> int32_t {
        const int hl = s512_log_n(jt->n_sig), id = s512_air_id(jt->n_sig);
        size_t bound = 0;
        VX_TRY(vx_stark_proof_bound(id, cfg, hl, &bound));
        j.proof.resize(bound);
        vx_buf* ht = nullptr;
        VX_TRY(vx_alloc(c, ((size_t)VX_SHA512_AIR_COLS) << hl, &ht));
        uint64_t hpub[15];
        int32_t r = vx_sha512_trace_dev(c, just->pubkeys, just->signatures, just->precommit, jt->chosen.data(), just->num_authorities, hl, 1, ht->d, hpub);
        if (r == VX_OK) r = vx_stark_prove_impl(c, id, cfg, ht->d, ht->n, /*consume_trace=*/0, hl, hpub, 15, j.proof.data(), j.proof.size(), &j.len, &jt->hooks[2]);
        (void)vx_free(c, ht);
        return r;
    };
    for (int t = 0; t < 3; ++t) {
        if (!(mask >> t & 1)) continue;  // another shard's table
        TableJob* j = &jt->job[t];
        try {
            j->th = std::thread([=] {
                (void)hipSetDevice(j->c->device);
                j->rc = t == 0 ? prove_chain(j->c, *j) : t == 1 ? prove_ed(j->c, *j) : prove_s512(j->c, *j);
                if (j->rc != VX_OK) rv->fail(first + t);  // do not leave the other provers waiting at their hooks
            });
        } catch (...) {
            for (int u = t; u < 3; ++u)
                if (mask >> u & 1) rv->fail(first + u);
            return VX_ERR_DEVICE;
        }
    }
    return VX_OK;
}

int32_t vx_justification_tables_join(vx_ctx* ctx, JustificationTables* jt) {
    for (int t = 0; t < 3; ++t)
        if (jt->job[t].th.joinable()) jt->job[t].th.join();
    // a prover released from the rendezvous by somebody else's failure reports VX_ERR_STATEMENT without a message
    for (int t = 0; t < 3; ++t)
        if (jt->job[t].rc != VX_OK && vx_last_error(jt->job[t].c)[0]) return vx_fail(ctx, jt->job[t].rc, "%s", vx_last_error(jt->job[t].c));
    for (int t = 0; t < 3; ++t)
        if (jt->job[t].rc != VX_OK) return vx_fail(ctx, jt->job[t].rc, "justification table %d failed", t);
    return VX_OK;
}

// Map segments: contiguous runs of headers with (nearly) equal numbers of Blake2b compressions.  bounds[s] .. bounds[s + 1]
static void hr_segments(const uint32_t* sizes, size_t n, uint32_t S, std::vector<size_t>& bounds, std::vector<size_t>& chunks) {
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) total += (sizes[i] + 127) / 128;
    bounds.assign(1, 0), chunks.clear();
    size_t acc = 0, seg = 0;
    for (size_t i = 0; i < n; ++i) {
        acc += (sizes[i] + 127) / 128, seg += (sizes[i] + 127) / 128;
        const size_t s = bounds.size();  // the segment being filled is s - 1
        const size_t left_h = n - 1 - i, left_s = S - s;  // headers / segments still to come
        if (s < S && (acc * S >= s * total || left_h == left_s) && left_h >= left_s) bounds.push_back(i + 1), chunks.push_back(seg), seg = 0;
    }
    bounds.push_back(n), chunks.push_back(seg);
}
static int blake_log_n(size_t chunks) {
    int log_n = blk::TABLE_LOG;  // at least one copy of the lookup tables
    while (((size_t)1 << log_n) < 16 * chunks) ++log_n;
    return log_n;
}

static int32_t hr_prove(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes, size_t n_fetched, uint32_t max_headers, uint32_t trusted_block,
                        const uint8_t trusted_hash[32], uint32_t target_block, const vx_justification* just, const vx_stark_config* cfg, uint32_t S, uint32_t shard,
                        uint32_t n_shards, const vx_hr_exchange* xch, uint8_t out96[96], uint64_t* proof_out, size_t proof_cap, size_t* proof_len) {
    if (!ctx || !cfg || !proof_len || !out96) return VX_ERR_ARG;
    const int tree_id = tree_air_id(max_headers);
    VX_CHECK(tree_id, "header_range: max_headers %u has no Merkle AIR (16, 256 or 512)", max_headers);
    VX_CHECK(!just || (just->num_authorities >= 1 && just->num_authorities <= 512), "header_range: %u authorities (the EdDSA table holds 512)", just ? just->num_authorities : 0);
    VX_CHECK(S >= 1 && S <= VX_HR_MAX_SEGMENTS && S <= n_fetched, "header_range: %u map segments for %zu headers (1 .. %d, at most one per header)", S, n_fetched, (int)VX_HR_MAX_SEGMENTS);
    VX_CHECK(n_shards >= 1 && shard < n_shards && (n_shards == 1 || (xch && xch->fn)), "header_range: shard %u of %u (an exchange function goes with more than one shard)", shard, n_shards);
    VX_CHECK(cfg->cap_height >= 0 && cfg->cap_height <= 8, "header_range: cap_height %d", cfg->cap_height);
    // 1. statement + public outputs (map/reduce chain rules, Merkle roots)
    VX_TRY(vx_verify_subchain(ctx, headers, stride, sizes, n_fetched, max_headers, trusted_block, trusted_hash, target_block, out96));
    const size_t HDR = VX_HR_BLOB_FIXED_WORDS + S;
    const int tl = tree_log_n(max_headers);
    // 2. the tables, all on ONE logUp bus under shared lookup challenges (BusMeet), in bus order:
    //    0 .. S-1  BlakeChainAir   map segment s: every compression of its headers; sends state-root and data-root bytes
    //    S         ShaTreeAir      the two SHA-256 Merkle trees over exactly those roots (subchain_verification.rs:213-220, 268-274)
    //   with a justification (header_range.rs:49-54 -> justification.rs:195-257):
    //    S+1       ShaChainAir     the authority-set commitment (justification.rs:127-162); sends the keys of the signed authorities
    //    S+2       EdAir           [S]B = R + [h]A for every signed authority (:229-243); receives the keys, exchanges R || A / H with
    //    S+3       Sha512Air       H = SHA-512(R || A || precommit)
    //   table t is proven here when t mod n_shards == shard
    const int n_tables = (int)S + 1 + (just ? 3 : 0);
    auto mine = [&](int t) { return (uint32_t)t % n_shards == shard; };
    BusMeet rv;
    rv.n_parties = n_tables;
    rv.capw = (size_t)4 << cfg->cap_height;
    for (int t = 0; t < n_tables; ++t) rv.local[t] = mine(t);
    rv.xch = n_shards > 1 ? xch : nullptr;
    std::vector<size_t> bounds, seg_chunks;
    hr_segments(sizes, n_fetched, S, bounds, seg_chunks);
    // the hash every segment starts from: the Blake2b-256 digest of the header before it
    std::vector<uint8_t> digests;
    if (S > 1) {
        digests.resize(32 * n_fetched);
        VX_TRY(vx_blake2b_256_batch(ctx, headers, stride, sizes, n_fetched, digests.data()));
    }
    // the leaves of the two Merkle trees: decode_header on the GPU (all four compact modes)
    std::vector<uint32_t> numbers(n_fetched);
    std::vector<uint8_t> modes(n_fetched), oks(n_fetched), parents(32 * n_fetched), sroots(32 * n_fetched), droots(32 * n_fetched);
    if (mine((int)S)) VX_TRY(vx_decode_header_batch(ctx, headers, stride, sizes, n_fetched, numbers.data(), modes.data(), oks.data(), parents.data(), sroots.data(), droots.data()));
    // one context per local table: the first local segment runs on `ctx` from this thread, everything else on a chain of side contexts
    std::vector<BusParty> party(n_tables);
    std::vector<vx_chal_hook> hooks(n_tables);
    for (int t = 0; t < n_tables; ++t) party[t] = {&rv, t}, hooks[t] = {vx_bus_hook, &party[t]};
    int main_seg = -1;
    for (int t = 0; t < (int)S && main_seg < 0; ++t)
        if (mine(t)) main_seg = t;
    std::vector<vx_ctx*> tctx(n_tables, nullptr);
    {
        vx_ctx* c = ctx;
        for (int t = 0; t < n_tables; ++t) {
            if (!mine(t)) continue;
            if (t == main_seg) tctx[t] = ctx;
            else {
                c = c ? vx_side_ctx(c) : nullptr;
                VX_CHECK(c, "header_range: no side context for every table (the provers meet at their challenge hooks, each on its own context)");
                tctx[t] = c;
            }
        }
    }
    std::vector<TableJob> seg(S);
    auto prove_segment = [&](int s, vx_ctx* c, TableJob& j) -> int32_t {
        const size_t a = bounds[s], b = bounds[s + 1];
        const int log_n = blake_log_n(seg_chunks[s]);
        size_t bound = 0;
        VX_TRY(vx_stark_proof_bound(VX_AIR_BLAKE_CHAIN, cfg, log_n, &bound));
        j.proof.resize(bound);
        vx_buf* trace = nullptr;
        VX_TRY(vx_alloc(c, ((size_t)blk::COLS) << log_n, &trace));
        vx_buf view{headers->d + a * stride / 8, headers->n - a * stride / 8};
        uint64_t pub[20];
        int32_t r = vx_blake_chain_trace(c, &view, stride, sizes + a, b - a, a ? digests.data() + 32 * (a - 1) : trusted_hash, trusted_block + 1 + (uint32_t)a, max_headers,
                                         (uint32_t)a, 0, log_n, trace, pub, nullptr);
        if (r == VX_OK && b == n_fetched) {
            uint8_t tgt[32];
            for (int q = 0; q < 8; ++q) {
                uint32_t l = (uint32_t)pub[8 + q];
                memcpy(tgt + 4 * q, &l, 4);
            }
            if (memcmp(tgt, out96, 32) != 0) r = vx_fail(c, VX_ERR_STATEMENT, "header_range: chain digest differs from the subchain target hash");
        }
        if (r == VX_OK) r = vx_stark_prove_impl(c, VX_AIR_BLAKE_CHAIN, cfg, trace->d, trace->n, /*consume_trace=*/1, log_n, pub, 20, j.proof.data(), j.proof.size(), &j.len, &hooks[s]);
        (void)vx_free(c, trace);
        return r;
    };
    TableJob tree;
    uint64_t tpub[17];
    auto prove_tree = [&](vx_ctx* c, TableJob& j) -> int32_t {
        vx_buf* tt = nullptr;
        int32_t r = vx_alloc(c, ((size_t)VX_SHA_TREE_AIR_COLS) << tl, &tt);
        if (r == VX_OK) r = vx_sha_tree_trace_dev(c, sroots.data(), droots.data(), n_fetched, tl - 8, tt->d, tpub);
        if (r == VX_OK) {
            uint8_t roots[64];
            for (int q = 0; q < 16; ++q)
                for (int b = 0; b < 4; ++b) roots[4 * q + b] = (uint8_t)(tpub[q] >> (24 - 8 * b));
            if (memcmp(roots, out96 + 32, 64) != 0) r = vx_fail(c, VX_ERR_STATEMENT, "header_range: Merkle AIR roots differ from the subchain roots");
        }
        size_t bound = 0;
        if (r == VX_OK) r = vx_stark_proof_bound(tree_id, cfg, tl, &bound);
        if (r == VX_OK) {
            j.proof.resize(bound);
            r = vx_stark_prove_impl(c, tree_id, cfg, tt->d, tt->n, /*consume_trace=*/0, tl, tpub, 17, j.proof.data(), j.proof.size(), &j.len, &hooks[S]);
        }
        if (tt) (void)vx_free(c, tt);
        return r;
    };
    // the target header is justified by > 2/3 of the committed authority set: every rule natively first (error behaviour of
    // the reference's hint, justification.rs:29-83) -- on the commitment table's thread, or here when that table is another shard's
    struct PreArgs {
        const vx_justification* just;
        uint32_t target_block;
        const uint8_t* target_hash;
    } pre_args{just, target_block, out96};
    auto pre = [](vx_ctx* c, void* u) -> int32_t {
        const PreArgs* a = (const PreArgs*)u;
        return vx_verify_simple_justification(c, a->target_block, a->target_hash, a->just->authority_set_id, a->just->authority_set_hash, a->just->precommit,
                                              a->just->pubkeys, a->just->signatures, a->just->validator_signed, a->just->num_authorities, a->just->max_authorities);
    };
    int32_t rc = VX_OK;
    if (just && !mine((int)S + 1)) rc = pre(ctx, &pre_args);
    if (rc != VX_OK) {  // nothing has been started yet; the other shards see this shard's tables fail
        for (int t = 0; t < n_tables; ++t)
            if (mine(t)) rv.fail(t);
        return rc;
    }
    JustificationTables jt;
    auto start = [&](TableJob& j, int t, auto&& fn) {
        j.c = tctx[t];
        try {
            j.th = std::thread([&j, &rv, t, fn] {
                (void)hipSetDevice(j.c->device);
                j.rc = fn(j.c, j);
                if (j.rc != VX_OK) rv.fail(t);  // do not leave the other provers waiting at their hooks
            });
        } catch (...) {
            j.rc = VX_ERR_DEVICE;
            rv.fail(t);
        }
    };
    for (int s = 0; s < (int)S; ++s)
        if (mine(s) && s != main_seg) start(seg[s], s, [&, s](vx_ctx* c, TableJob& j) { return prove_segment(s, c, j); });
    if (mine((int)S)) start(tree, (int)S, [&](vx_ctx* c, TableJob& j) { return prove_tree(c, j); });
    if (just) {
        vx_ctx* jc[3] = {tctx[S + 1], tctx[S + 2], tctx[S + 3]};
        const unsigned mask = (mine((int)S + 1) ? 1u : 0) | (mine((int)S + 2) ? 2u : 0) | (mine((int)S + 3) ? 4u : 0);
        rc = vx_justification_tables_start(jc, just, cfg, &rv, (int)S + 1, pre, &pre_args, &jt, mask);
        if (rc != VX_OK) (void)vx_fail(ctx, rc, "header_range: no host thread for the justification tables");
    }
    if (main_seg >= 0) {
        seg[main_seg].c = ctx;
        if (rc == VX_OK) seg[main_seg].rc = prove_segment(main_seg, ctx, seg[main_seg]);
        else seg[main_seg].rc = rc;
        if (seg[main_seg].rc != VX_OK) rv.fail(main_seg);
    }
    for (int s = 0; s < (int)S; ++s)
        if (seg[s].th.joinable()) seg[s].th.join();
    if (tree.th.joinable()) tree.th.join();
    const int32_t rc_just = just ? vx_justification_tables_join(ctx, &jt) : VX_OK;  // (the justification's own rules name the error first)
    if (rc == VX_OK) {
        if (rc_just != VX_OK) rc = rc_just;
        else if (tree.rc != VX_OK) rc = vx_fail(ctx, tree.rc, "%s", vx_last_error(tree.c));
        else
            for (int s = 0; s < (int)S && rc == VX_OK; ++s)
                if (seg[s].rc != VX_OK) rc = s == main_seg ? seg[s].rc : vx_fail(ctx, seg[s].rc, "%s", vx_last_error(seg[s].c));
    }
    if (rc != VX_OK) return rc;
    size_t total = HDR + tree.len + (just ? jt.job[0].len + jt.job[1].len + jt.job[2].len : 0);
    for (int s = 0; s < (int)S; ++s) total += seg[s].len;
    *proof_len = total;
    if (!proof_out || proof_cap < total) return vx_fail(ctx, VX_ERR_BUFSZ, "header_range: proof needs %zu words, buffer has %zu", total, proof_cap);
    size_t off = HDR;
    for (int s = 0; s < (int)S; ++s) memcpy(proof_out + off, seg[s].proof.data(), seg[s].len * 8), off += seg[s].len;
    const TableJob* order[4] = {just ? &jt.job[0] : nullptr, &tree, just ? &jt.job[1] : nullptr, just ? &jt.job[2] : nullptr};  // commitment, Merkle, Ed25519, SHA-512
    for (int q = 0; q < 4; ++q)
        if (order[q]) memcpy(proof_out + off, order[q]->proof.data(), order[q]->len * 8), off += order[q]->len;
    proof_out[0] = VX_HR_MAGIC;
    proof_out[1] = max_headers;
    proof_out[2] = trusted_block;
    proof_out[3] = target_block;
    memcpy(proof_out + 4, out96, 96);
    proof_out[16] = S;
    proof_out[17] = just ? jt.job[0].len : 0;
    proof_out[18] = tree.len;
    proof_out[19] = just ? jt.job[1].len : 0;
    proof_out[20] = just ? jt.job[2].len : 0;
    uint64_t round = 0;
    if (just) memcpy(&round, just->precommit + 37, 8);  // 0x01 || hash 32 || block 4 || round 8 || set id 8 (decoder.rs:159-200)
    proof_out[21] = round;
    for (int s = 0; s < (int)S; ++s) proof_out[VX_HR_BLOB_FIXED_WORDS + s] = seg[s].len;
    return VX_OK;
}

extern "C" {

int32_t vx_header_range_prove(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes, size_t n_fetched,
                              uint32_t max_headers, uint32_t trusted_block, const uint8_t trusted_hash[32], uint32_t target_block,
                              const vx_justification* just, const vx_stark_config* cfg, uint8_t out96[96], uint64_t* proof_out,
                              size_t proof_cap, size_t* proof_len) {
    return hr_prove(ctx, headers, stride, sizes, n_fetched, max_headers, trusted_block, trusted_hash, target_block, just, cfg, 1, 0, 1, nullptr, out96, proof_out, proof_cap, proof_len);
}
int32_t vx_header_range_prove_ex(vx_ctx* ctx, const vx_buf* headers, size_t stride, const uint32_t* sizes, size_t n_fetched, uint32_t max_headers, uint32_t trusted_block,
                                 const uint8_t trusted_hash[32], uint32_t target_block, const vx_justification* just, const vx_stark_config* cfg, uint32_t n_segments,
                                 uint32_t shard, uint32_t n_shards, const vx_hr_exchange* exchange, uint8_t out96[96], uint64_t* proof_out, size_t proof_cap, size_t* proof_len) {
    return hr_prove(ctx, headers, stride, sizes, n_fetched, max_headers, trusted_block, trusted_hash, target_block, just, cfg, n_segments, shard, n_shards, exchange, out96,
                    proof_out, proof_cap, proof_len);
}
int32_t vx_header_range_proof_bound_ex(const vx_stark_config* cfg, size_t n_chunks, size_t n_authorities, uint32_t n_segments, size_t* n_words) {
    if (!cfg || !n_words || n_chunks == 0 || n_segments < 1 || n_segments > VX_HR_MAX_SEGMENTS) return VX_ERR_ARG;
    size_t one = 0;
    // a segment is never taller than the unsegmented table: S copies of that bound (generous), plus the longer header
    const int32_t rc = vx_header_range_proof_bound(cfg, n_chunks, n_authorities, &one);
    size_t w1 = 0;
    int32_t rc2 = vx_stark_proof_bound(VX_AIR_BLAKE_CHAIN, cfg, blake_log_n(n_chunks), &w1);
    *n_words = one + (size_t)(n_segments - 1) * w1 + n_segments;
    return rc != VX_OK ? rc : rc2;
}
// host only: the blobs of the shards of one proof -> the request's blob
int32_t vx_header_range_merge(const uint64_t** blobs, const size_t* lens, size_t n_blobs, uint64_t* out, size_t out_cap, size_t* out_len, char* err, size_t errlen) {
    auto bad = [&](const char* why) {
        if (err && errlen) snprintf(err, errlen, "%s", why);
        return (int32_t)VX_ERR_STATEMENT;
    };
    if (!blobs || !lens || !out_len || n_blobs == 0) return VX_ERR_ARG;
    for (size_t k = 0; k < n_blobs; ++k)
        if (!blobs[k] || lens[k] <= VX_HR_BLOB_FIXED_WORDS || blobs[k][0] != VX_HR_MAGIC) return bad("merge: not a header_range blob");
    const uint64_t S = blobs[0][16];
    if (S < 1 || S > VX_HR_MAX_SEGMENTS) return bad("merge: bad segment count");
    const size_t HDR = VX_HR_BLOB_FIXED_WORDS + S, NT = S + 4;
    // table lengths of blob k in blob order: segments, commitment, Merkle, Ed25519, SHA-512
    auto tlen = [&](size_t k, size_t t) -> uint64_t { return t < S ? blobs[k][VX_HR_BLOB_FIXED_WORDS + t] : blobs[k][17 + (t - S)]; };
    std::vector<int> src(NT, -1);
    std::vector<std::vector<size_t>> offs(n_blobs, std::vector<size_t>(NT, 0));
    for (size_t k = 0; k < n_blobs; ++k) {
        if (lens[k] < HDR || blobs[k][16] != S || memcmp(blobs[k], blobs[0], 16 * 8) != 0 || blobs[k][21] != blobs[0][21]) return bad("merge: the blobs are not shards of one proof");
        size_t off = HDR;
        for (size_t t = 0; t < NT; ++t) {
            const uint64_t l = tlen(k, t);
            if (l > lens[k] - off) return bad("merge: blob lengths are inconsistent");
            offs[k][t] = off, off += l;
            if (l) {
                if (src[t] >= 0) return bad("merge: a table is present in two shards");
                src[t] = (int)k;
            }
        }
        if (off != lens[k]) return bad("merge: blob lengths are inconsistent");
    }
    size_t total = HDR;
    for (size_t t = 0; t < NT; ++t)
        if (src[t] >= 0) total += tlen(src[t], t);
    *out_len = total;
    if (!out || out_cap < total) return VX_ERR_BUFSZ;
    memcpy(out, blobs[0], HDR * 8);
    size_t off = HDR;
    for (size_t t = 0; t < NT; ++t) {
        const uint64_t l = src[t] >= 0 ? tlen(src[t], t) : 0;
        if (t < S) out[VX_HR_BLOB_FIXED_WORDS + t] = l;
        else out[17 + (t - S)] = l;
        if (l) memcpy(out + off, blobs[src[t]] + offs[src[t]][t], l * 8), off += l;
    }
    return VX_OK;
}
}
