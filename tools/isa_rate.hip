// Issue-rate microbenchmark for the integer instructions the Goldilocks kernels are made of (gfx950).
// Each kernel runs ITER x 64 independent instructions of one kind per wave; cycles per instruction per SIMD =
// elapsed * clock / (ITER * 64 * waves_per_simd).   hipcc --offload-arch=gfx950 -O3 -o isa_rate isa_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define REP8(x) x x x x x x x x
#define BODY(name, ins)                                                                                  \
    __global__ __launch_bounds__(256) void name(uint64_t* out, uint32_t seed) {                         \
        uint32_t a = threadIdx.x + seed, b = a * 7 + 1, c = b ^ 0x55;                                    \
        uint64_t r0 = a, r1 = b, r2 = c, r3 = a + b, r4 = 5, r5 = 6, r6 = 7, r7 = 8;                     \
        uint32_t w0 = a, w1 = b, w2 = c, w3 = 3, w4 = 4, w5 = 5, w6 = 6, w7 = 7;                         \
        for (int i = 0; i < ITER; ++i) { REP8(ins) }                                                     \
        out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7; \
    }
#define R8(op, fmt) \
    asm volatile(op " %0, " fmt : "+v"(r0) : "v"(a), "v"(b)); asm volatile(op " %0, " fmt : "+v"(r1) : "v"(a), "v"(b)); \
    asm volatile(op " %0, " fmt : "+v"(r2) : "v"(a), "v"(b)); asm volatile(op " %0, " fmt : "+v"(r3) : "v"(a), "v"(b)); \
    asm volatile(op " %0, " fmt : "+v"(r4) : "v"(a), "v"(b)); asm volatile(op " %0, " fmt : "+v"(r5) : "v"(a), "v"(b)); \
    asm volatile(op " %0, " fmt : "+v"(r6) : "v"(a), "v"(b)); asm volatile(op " %0, " fmt : "+v"(r7) : "v"(a), "v"(b));
#define W8(op, fmt) \
    asm volatile(op " %0, " fmt : "+v"(w0) : "v"(a), "v"(b)); asm volatile(op " %0, " fmt : "+v"(w1) : "v"(a), "v"(b)); \
    asm volatile(op " %0, " fmt : "+v"(w2) : "v"(a), "v"(b)); asm volatile(op " %0, " fmt : "+v"(w3) : "v"(a), "v"(b)); \
    asm volatile(op " %0, " fmt : "+v"(w4) : "v"(a), "v"(b)); asm volatile(op " %0, " fmt : "+v"(w5) : "v"(a), "v"(b)); \
    asm volatile(op " %0, " fmt : "+v"(w6) : "v"(a), "v"(b)); asm volatile(op " %0, " fmt : "+v"(w7) : "v"(a), "v"(b));
BODY(k_mad_u64_u32, R8("v_mad_u64_u32", "vcc, %1, %2, %0"))
BODY(k_lshl_add_u64, R8("v_lshl_add_u64", "%0, 0, %0"))
BODY(k_lshlrev_b64, R8("v_lshlrev_b64", "3, %0"))
BODY(k_add_u32, W8("v_add_u32", "%1, %0"))
BODY(k_add_co_u32, W8("v_add_co_u32", "vcc, %1, %0"))
BODY(k_addc_co_u32, W8("v_addc_co_u32", "vcc, %1, %0, vcc"))
BODY(k_mul_lo_u32, W8("v_mul_lo_u32", "%1, %0"))
BODY(k_mul_hi_u32, W8("v_mul_hi_u32", "%1, %0"))
BODY(k_mul_u32_u24, W8("v_mul_u32_u24", "%1, %0"))
BODY(k_mad_u32_u24, W8("v_mad_u32_u24", "%1, %2, %0"))
BODY(k_mov_b32, W8("v_mov_b32", "%1"))
BODY(k_cndmask, W8("v_cndmask_b32", "%1, %0, vcc"))
BODY(k_cndmask_e64, W8("v_cndmask_b32_e64", "%1, %0, s[10:11]"))
BODY(k_cndmask_imm, W8("v_cndmask_b32_e64", "0, -1, vcc"))
BODY(k_cndmask_dst, asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(w0) : "v"(a), "v"(b)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(w1) : "v"(a), "v"(b));
     asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(w2) : "v"(a), "v"(b)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(w3) : "v"(a), "v"(b));
     asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(w4) : "v"(a), "v"(b)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(w5) : "v"(a), "v"(b));
     asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(w6) : "v"(a), "v"(b)); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(w7) : "v"(a), "v"(b));)
BODY(k_subb_mask, W8("v_subb_co_u32", "vcc, %0, %0, vcc"))
BODY(k_sub_co_e64, W8("v_sub_co_u32_e64", "s[10:11], %1, %0"))
BODY(k_cmp_u32, asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(b), "v"(a) : "vcc");
     asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a), "v"(c) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(c), "v"(a) : "vcc");
     asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(b), "v"(a) : "vcc");
     asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a), "v"(c) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(c), "v"(a) : "vcc");)
BODY(k_cmp_u64, asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(r0), "v"(r1) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(r1), "v"(r2) : "vcc");
     asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(r2), "v"(r3) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(r3), "v"(r4) : "vcc");
     asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(r4), "v"(r5) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(r5), "v"(r6) : "vcc");
     asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(r6), "v"(r7) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(r7), "v"(r0) : "vcc");)
BODY(k_and_or, W8("v_and_or_b32", "%1, %2, %0"))
BODY(k_lshl_add_u32, W8("v_lshl_add_u32", "%1, 3, %0"))
BODY(k_sub_u32, W8("v_sub_u32", "%1, %0"))
BODY(k_and_b32, W8("v_and_b32", "%1, %0"))
BODY(k_lshrrev_b32, W8("v_lshrrev_b32", "3, %0"))
BODY(k_mov_b64, R8("v_mov_b64", "%0"))
BODY(k_pk_add_u16, W8("v_pk_add_u16", "%1, %0"))
BODY(k_xor, W8("v_xor_b32", "%1, %0"))
BODY(k_alignbit, W8("v_alignbit_b32", "%1, %0, 7"))
BODY(k_add3, W8("v_add3_u32", "%1, %2, %0"))

// the same Goldilocks add chain with compiler-chosen selects and with selects forced to the VOP3 (e64) encoding
#define GL_EPS 0xFFFFFFFFULL
__device__ __forceinline__ uint64_t sel64_e64(bool c, uint64_t a, uint64_t b) {  // c ? a : b
    const uint64_t m = __ballot(c);
    uint32_t lo, hi;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"((uint32_t)b), "v"((uint32_t)a), "s"(m));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"((uint32_t)(b >> 32)), "v"((uint32_t)(a >> 32)), "s"(m));
    return ((uint64_t)hi << 32) | lo;
}
template <int E64> __device__ __forceinline__ uint64_t gadd(uint64_t a, uint64_t b) {
    uint64_t s, t;
    const bool c1 = __builtin_add_overflow(a, b, &s);
    const bool c2 = __builtin_add_overflow(s, (uint64_t)GL_EPS, &t);
    if (E64) return sel64_e64(c1 | c2, t, s);
    return (c1 | c2) ? t : s;
}
template <int E64> __global__ __launch_bounds__(256) void k_gladd(uint64_t* out, uint32_t seed) {
    uint64_t x[8];
    for (int j = 0; j < 8; ++j) x[j] = 0x9E3779B97F4A7C15ULL * (threadIdx.x + seed + j) | 1;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = gadd<E64>(x[j], x[(j + 1) & 7]);
    }
    out[blockIdx.x * 256 + threadIdx.x] = x[0] ^ x[1] ^ x[2] ^ x[3] ^ x[4] ^ x[5] ^ x[6] ^ x[7];
}
#undef GL_EPS
#include "../0-kno-vectorx_amd/csrc/gl.cuh"
// 8 independent Goldilocks multiply chains per lane: cycles per modmul = reported value x 8
__global__ __launch_bounds__(256) void k_glmul(uint64_t* out, uint32_t seed) {
    uint64_t x[8];
    for (int j = 0; j < 8; ++j) x[j] = 0x9E3779B97F4A7C15ULL * (threadIdx.x + seed + j) | 1;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = gl_mul_nc(x[j], x[(j + 1) & 7]);
    }
    out[blockIdx.x * 256 + threadIdx.x] = x[0] ^ x[1] ^ x[2] ^ x[3] ^ x[4] ^ x[5] ^ x[6] ^ x[7];
}
// the Poseidon s-box chain: x^7 on 8 independent values (4 modmuls each); cycles per s-box = value x 8
__global__ __launch_bounds__(256) void k_sbox(uint64_t* out, uint32_t seed) {
    uint64_t x[8];
    for (int j = 0; j < 8; ++j) x[j] = 0x9E3779B97F4A7C15ULL * (threadIdx.x + seed + j) | 1;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t v = x[j], x2 = gl_mul_nc(v, v), x3 = gl_mul_nc(x2, v), x4 = gl_mul_nc(x2, x2);
            x[j] = gl_mul_nc(x3, x4);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x[0] ^ x[1] ^ x[2] ^ x[3] ^ x[4] ^ x[5] ^ x[6] ^ x[7];
}
#include "../0-kno-vectorx_amd/csrc/poseidon.cuh"
#define PITER 64
// value reported = cycles per (ITER*64) "instructions"; these two run PITER iterations: scale by ITER*64/PITER outside
__global__ __launch_bounds__(256) void k_mds(uint64_t* out, uint32_t seed) {
    uint64_t s[12];
    for (int j = 0; j < 12; ++j) s[j] = 0x9E3779B97F4A7C15ULL * (threadIdx.x + seed + j) | 1;
#pragma unroll 1
    for (int i = 0; i < PITER * 30; ++i) poseidon_mds<12>(s, (i % 29) * 12);
    uint64_t x = 0;
    for (int j = 0; j < 12; ++j) x ^= s[j];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_perm(uint64_t* out, uint32_t seed) {
    uint64_t s[12];
    for (int j = 0; j < 12; ++j) s[j] = 0x9E3779B97F4A7C15ULL * (threadIdx.x + seed + j) | 1;
#pragma unroll 1
    for (int i = 0; i < PITER; ++i) poseidon_permute(s);
    uint64_t x = 0;
    for (int j = 0; j < 12; ++j) x ^= s[j];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <class T, class TIn>
__global__ __launch_bounds__(256) void k_mds_part(uint64_t* out, uint32_t seed) {
    TIn x[12];
    for (int j = 0; j < 12; ++j) x[j] = (TIn)((0x9E3779B97F4A7C15ULL * (threadIdx.x + seed + j)) & 0xFFF);
#pragma unroll 1
    for (int i = 0; i < PITER * 30; ++i) {
        T y[12];
        poseidon_mds_part<T>(x, y);
        for (int j = 0; j < 12; ++j) x[j] = (TIn)(y[j] & 0xFFF) + (TIn)i;
    }
    uint64_t r = 0;
    for (int j = 0; j < 12; ++j) r ^= (uint64_t)x[j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
typedef void (*kern)(uint64_t*, uint32_t);
int main() {
    uint64_t* d;
    hipMalloc(&d, 8ull * 256 * 8192);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double clk = p.clockRate * 1e3;  // Hz
    const int cus = p.multiProcessorCount;
    struct { const char* n; kern k; } ks[] = {{"v_mad_u64_u32", k_mad_u64_u32}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_lshlrev_b64", k_lshlrev_b64},
        {"v_add_u32", k_add_u32}, {"v_add_co_u32", k_add_co_u32}, {"v_addc_co_u32", k_addc_co_u32}, {"v_mul_lo_u32", k_mul_lo_u32},
        {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mov_b32", k_mov_b32},
        {"PERM: cycles per permutation /(ITER*64/PITER)", k_perm}, {"MDSPART64x30 /(ITER*64/PITER)", k_mds_part<int64_t, uint64_t>}, {"MDSPART32x30 /(ITER*64/PITER)", k_mds_part<int32_t, uint32_t>}, {"MDS30: cycles per 30 MDS layers /(ITER*64/PITER)", k_mds}, {"gl_mul_nc chain (x8 per modmul)", k_glmul}, {"sbox x^7 chain (x8 per sbox)", k_sbox}, {"gl_add chain (8 adds/iter; per add x8) compiler", k_gladd<0>}, {"gl_add chain e64 selects", k_gladd<1>}, {"v_cndmask_b32", k_cndmask}, {"v_cndmask_b32_e64 sgpr", k_cndmask_e64}, {"v_cndmask_b32_e64 0,-1,vcc", k_cndmask_imm}, {"v_cndmask_b32 fresh dst", k_cndmask_dst}, {"v_subb_co_u32 x,x,vcc (mask)", k_subb_mask}, {"v_sub_co_u32_e64 sgpr", k_sub_co_e64}, {"v_cmp_lt_u32", k_cmp_u32}, {"v_cmp_lt_u64", k_cmp_u64}, {"v_and_or_b32", k_and_or}, {"v_lshl_add_u32", k_lshl_add_u32}, {"v_sub_u32", k_sub_u32}, {"v_and_b32", k_and_b32}, {"v_lshrrev_b32", k_lshrrev_b32}, {"v_mov_b64", k_mov_b64}, {"v_pk_add_u16", k_pk_add_u16}, {"v_xor_b32", k_xor}, {"v_alignbit_b32", k_alignbit}, {"v_add3_u32", k_add3}};
    printf("{\"clock_hz\": %.0f, \"cus\": %d, \"rates\": {", clk, cus);
    bool first = true;
    for (auto& e : ks) {
        for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: blocks of 256 threads = 4 waves = one per SIMD of a CU
            const int blocks = cus * wps;
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            e.k<<<blocks, 256>>>(d, 1);
            hipDeviceSynchronize();
            hipEventRecord(a);
            e.k<<<blocks, 256>>>(d, 2);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double cyc = ms * 1e-3 * clk / ((double)ITER * 64 * wps);
            printf("%s\"%s@%dwps\": %.2f", first ? "" : ", ", e.n, wps, cyc);
            first = false;
        }
    }
    printf("}}\n");
    return 0;
}
