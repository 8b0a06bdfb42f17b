// Poseidon permutation / MDS-layer cost on gfx950 for one build of csrc/poseidon.cuh, plus the issue cost of the
// double-precision instructions the exact-fp64 MDS low part is made of.  Built twice by tools/ab_poseidon_fp64.sh
// (-DVX_POSEIDON_FP64=0 / 1); prints cycles per wave-permutation, per MDS layer, and a checksum of fixed permutations that
// must be the same for every build.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../0-kno-vectorx_amd/csrc/poseidon.cuh"
#define PITER 64
#define ITER 4096
__global__ __launch_bounds__(256) void k_mds(uint64_t* out, uint32_t seed) {
    uint64_t s[12];
    for (int j = 0; j < 12; ++j) s[j] = 0x9E3779B97F4A7C15ULL * (threadIdx.x + seed + j) | 1;
#pragma unroll 1
    for (int i = 0; i < PITER * 30; ++i) poseidon_mds<12>(s, (i % 29) * 12);
    uint64_t x = 0;
    for (int j = 0; j < 12; ++j) x ^= s[j];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_mds1(uint64_t* out, uint32_t seed) {
    uint64_t s[12];
    for (int j = 0; j < 12; ++j) s[j] = 0x9E3779B97F4A7C15ULL * (threadIdx.x + seed + j) | 1;
#pragma unroll 1
    for (int i = 0; i < PITER * 30; ++i) poseidon_mds<1>(s, (i % 29) * 12);
    uint64_t x = 0;
    for (int j = 0; j < 12; ++j) x ^= s[j];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_perm(uint64_t* out, uint32_t seed) {
    uint64_t s[12];
    for (int j = 0; j < 12; ++j) s[j] = 0x9E3779B97F4A7C15ULL * (threadIdx.x + seed + j) | 1;
#pragma unroll 1
    for (int i = 0; i < PITER; ++i) poseidon_permute(s);
    uint64_t x = 0;
    for (int j = 0; j < 12; ++j) x ^= s[j];
    out[blockIdx.x * 256 + threadIdx.x] = x;
}
// one permutation of 256 x 64 fixed states; the xor / sum of all outputs is the build-independent checksum
__global__ void k_check(uint64_t* out) {
    uint64_t s[12];
    const uint64_t t = blockIdx.x * 256 + threadIdx.x;
    for (int j = 0; j < 12; ++j) {
        uint64_t v = 0xD1B54A32D192ED03ULL * (t * 12 + j + 1);
        if ((t & 7) == 1) v = ~0ULL - j;            // non-canonical representatives on purpose
        if ((t & 7) == 2) v = 0xFFFFFFFF00000000ULL + j;
        if ((t & 7) == 3) v = j;
        s[j] = v;
    }
    poseidon_permute(s);
    uint64_t x = 0, y = 0;
    for (int j = 0; j < 12; ++j) x ^= s[j] * (2 * j + 1), y += s[j];
    out[2 * t] = x, out[2 * t + 1] = y;
}
#define FBODY(name, ins)                                                                        \
    __global__ __launch_bounds__(256) void name(uint64_t* out, uint32_t seed) {                \
        double a = (double)(threadIdx.x + seed), b = a * 0.5 + 1.0;                             \
        double r0 = a, r1 = b, r2 = a + 2, r3 = a + 3, r4 = 5, r5 = 6, r6 = 7, r7 = 8;          \
        for (int i = 0; i < ITER; ++i) { ins ins ins ins ins ins ins ins }                      \
        out[blockIdx.x * 256 + threadIdx.x] = (uint64_t)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7); \
    }
#define F8(op, fmt) \
    asm volatile(op " %0, " fmt : "+v"(r0) : "v"(a), "v"(b)); asm volatile(op " %0, " fmt : "+v"(r1) : "v"(a), "v"(b)); \
    asm volatile(op " %0, " fmt : "+v"(r2) : "v"(a), "v"(b)); asm volatile(op " %0, " fmt : "+v"(r3) : "v"(a), "v"(b)); \
    asm volatile(op " %0, " fmt : "+v"(r4) : "v"(a), "v"(b)); asm volatile(op " %0, " fmt : "+v"(r5) : "v"(a), "v"(b)); \
    asm volatile(op " %0, " fmt : "+v"(r6) : "v"(a), "v"(b)); asm volatile(op " %0, " fmt : "+v"(r7) : "v"(a), "v"(b));
FBODY(k_add_f64, F8("v_add_f64", "%0, %1"))
FBODY(k_fma_f64, F8("v_fma_f64", "%1, %2, %0"))
FBODY(k_mul_f64, F8("v_mul_f64", "%0, %1"))
typedef void (*kern)(uint64_t*, uint32_t);
int main() {
    uint64_t* d;
    hipMalloc(&d, 8ull * 256 * 8192);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double clk = p.clockRate * 1e3;
    const int cus = p.multiProcessorCount;
    k_check<<<64, 256>>>(d);
    hipDeviceSynchronize();
    static uint64_t h[2 * 64 * 256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    uint64_t cx = 0, cy = 0;
    for (size_t i = 0; i < 64 * 256; ++i) cx ^= h[2 * i] + i, cy += h[2 * i + 1] % 0xFFFFFFFF00000001ULL;
    printf("{\"fp64\": %d, \"split_partial\": %d, \"checksum\": \"%016llx%016llx\"", VX_POSEIDON_FP64, VX_POSEIDON_SPLIT_PARTIAL, (unsigned long long)cx, (unsigned long long)cy);
    struct { const char* n; kern k; double scale; } ks[] = {
        {"perm_cycles_per_wave", k_perm, (double)ITER * 64 / PITER}, {"mds12_cycles_per_layer", k_mds, (double)ITER * 64 / (PITER * 30)},
        {"mds1_cycles_per_layer", k_mds1, (double)ITER * 64 / (PITER * 30)}, {"v_add_f64", k_add_f64, 1}, {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1}};
    for (auto& e : ks)
        for (int wps : {4, 6, 8}) {
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
            printf(", \"%s@%dwps\": %.2f", e.n, wps, ms * 1e-3 * clk / ((double)ITER * 64 * wps) * e.scale);
        }
    printf("}\n");
    return 0;
}
