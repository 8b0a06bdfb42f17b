// Context lifecycle, HBM buffers, K1 batch field arithmetic.
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>

#include "gl.cuh"
#include <dlfcn.h>

#include "vx_internal.h"

// The provers keep 5-6 streams busy per proof and hosts run several proofs at once; the HIP runtime multiplexes all streams of a
// process onto GPU_MAX_HW_QUEUES hardware queues, 4 by default, which serialises them (header_range_256: 7.26 -> 7.86 proofs/s with
// 16; profiles/README.md).  The runtime reads the variable when it initialises (the first HIP call of the process), so it is set
// when the library is loaded, unless the host has chosen a value itself.  A host that touches HIP before loading the library
// exports it on its own.
__attribute__((constructor)) static void vx_runtime_env() { setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite=*/0); }

int32_t vx_fail(vx_ctx* ctx, int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

static int32_t make_powtab(vx_ctx* ctx, uint64_t base, PowTab* out) {
    std::vector<uint64_t> h(3 * 2048);
    uint64_t b = base;
    for (int l = 0; l < 3; ++l) {
        uint64_t acc = 1;
        for (int j = 0; j < 2048; ++j) {
            h[l * 2048 + j] = acc;
            acc = glh::mul(acc, b);
        }
        b = glh::pow(b, 2048);
    }
    VX_HIP(hipMalloc(&out->d, h.size() * 8));
    VX_HIP(hipMemcpyAsync(out->d, h.data(), h.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}

int32_t vx_get_shift_tab(vx_ctx* ctx, uint64_t base, PowTab* out) {
    auto it = ctx->shift_tabs.find(base);
    if (it == ctx->shift_tabs.end()) {
        PowTab t;
        VX_TRY(make_powtab(ctx, base, &t));
        it = ctx->shift_tabs.emplace(base, t).first;
    }
    *out = it->second;
    return VX_OK;
}

int32_t vx_get_tw2(vx_ctx* ctx, int log_s, int inverse, Tw2* out) {
    const int lo_bits = (log_s + 1) / 2, hi_bits = log_s - lo_bits;
    const int key = log_s * 2 + (inverse ? 1 : 0);
    auto it = ctx->tw2.find(key);
    if (it == ctx->tw2.end()) {
        const size_t nlo = (size_t)1 << lo_bits, nhi = (size_t)1 << hi_bits;
        std::vector<uint64_t> h(nlo + nhi);
        uint64_t w = glh::root(log_s);
        if (inverse) w = glh::inv(w);
        uint64_t acc = 1;
        for (size_t j = 0; j < nlo; ++j) {
            h[j] = acc;
            acc = glh::mul(acc, w);
        }
        const uint64_t wh = acc;  // w^(2^lo_bits)
        acc = 1;
        for (size_t j = 0; j < nhi; ++j) {
            h[nlo + j] = acc;
            acc = glh::mul(acc, wh);
        }
        uint64_t* d = nullptr;
        VX_HIP(hipMalloc(&d, h.size() * 8));
        VX_HIP(hipMemcpyAsync(d, h.data(), h.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        VX_HIP(hipStreamSynchronize(ctx->stream));
        it = ctx->tw2.emplace(key, d).first;
    }
    out->lo = it->second;
    out->hi = it->second + ((size_t)1 << lo_bits);
    out->lo_bits = lo_bits;
    return VX_OK;
}

void vx_pool_trim(vx_ctx* ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->pool_free)
        for (void* p : kv.second) (void)hipFree(p);
    ctx->pool_free.clear();
}
void* vx_pool_alloc(vx_ctx* ctx, size_t bytes) {
    const size_t gran = (size_t)2 << 20;
    const size_t sz = ((bytes ? bytes : 1) + gran - 1) / gran * gran;
    auto it = ctx->pool_free.find(sz);
    void* p = nullptr;
    if (it != ctx->pool_free.end() && !it->second.empty()) {
        p = it->second.back();
        it->second.pop_back();
    } else if (hipMalloc(&p, sz) != hipSuccess) {
        (void)hipGetLastError();
        vx_pool_trim(ctx);  // give cached blocks back and retry once
        if (hipMalloc(&p, sz) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    ctx->pool_live[p] = sz;
    return p;
}
void vx_pool_free(vx_ctx* ctx, void* p) {
    if (!p) return;
    auto it = ctx->pool_live.find(p);
    if (it == ctx->pool_live.end()) {  // not ours: plain free
        (void)hipFree(p);
        return;
    }
    ctx->pool_free[it->second].push_back(p);
    ctx->pool_live.erase(it);
}

int32_t vx_scratch(vx_ctx* ctx, size_t n, uint64_t** out) {
    if (ctx->scratch_n < n) {
        if (ctx->scratch) {
            vx_pool_free(ctx, ctx->scratch);
            ctx->scratch = nullptr;
            ctx->scratch_n = 0;
        }
        ctx->scratch = (uint64_t*)vx_pool_alloc(ctx, n * 8);
        if (!ctx->scratch) return vx_fail(ctx, VX_ERR_OOM, "scratch: cannot allocate %zu bytes", n * 8);
        ctx->scratch_n = n;
    }
    *out = ctx->scratch;
    return VX_OK;
}

extern "C" {

const char* vx_backend_name(void) { return "hip-gfx950"; }

// Stream priorities (VX_STREAM_PRIO=1; off by default, see profiles/README.md): a context made by the host gets the device's
// highest priority, the side contexts a proof chains behind it (one per further table) the lowest -- the hash-chain table,
// which is the critical path of a proof, runs on the host's context.
static thread_local int g_side_ctx = 0;
int32_t vx_ctx_create(int device, vx_ctx** out) {
    if (!out) return VX_ERR_ARG;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0 || device < 0 || device >= n_dev) return VX_ERR_DEVICE;
    vx_ctx* ctx = new vx_ctx();
    ctx->device = device;
    ctx->scratch = nullptr;
    ctx->scratch_n = 0;
    ctx->pinned = nullptr;
    ctx->pinned_n = 0;
    {
        // A proof's host threads (one per table, several proofs in flight) mostly wait in hipStreamSynchronize.  The runtime's default
        // spins: 35.6 s of CPU for a 10 s bench run, 9.3 s with blocking waits, at the same 7.95 proofs/s (profiles/README.md) -- and
        // eight ranks share the cores of one node.  VX_SYNC_MODE = s (spin) / y (yield) / b (blocking, the default) overrides.
        const char* sm = getenv("VX_SYNC_MODE");
        const char m = sm ? sm[0] : 'b';
        (void)hipSetDevice(device);
        (void)hipSetDeviceFlags(m == 's' ? hipDeviceScheduleSpin : m == 'y' ? hipDeviceScheduleYield : hipDeviceScheduleBlockingSync);
        (void)hipGetLastError();  // (a host that has fixed the flags already keeps them)
    }
    hipError_t se = hipSetDevice(device);
    if (se == hipSuccess) {
        const char* sp = getenv("VX_STREAM_PRIO");
        int least = 0, greatest = 0;
        if (sp && sp[0] == '1' && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
            se = hipStreamCreateWithPriority(&ctx->stream, hipStreamDefault, g_side_ctx ? least : greatest);
        else se = hipStreamCreate(&ctx->stream);
    }
    if (se != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return VX_ERR_DEVICE;
    }
    int32_t rc = make_powtab(ctx, glh::ROOT_2_32, &ctx->tw_fwd);
    if (rc == VX_OK) rc = make_powtab(ctx, glh::inv(glh::ROOT_2_32), &ctx->tw_inv);
    if (rc == VX_OK) {
        std::vector<uint64_t> f(2048), b(2048);
        uint64_t w = glh::root(12), wi = glh::inv(w), a = 1, c = 1;
        for (int j = 0; j < 2048; ++j) {
            f[j] = a;
            b[j] = c;
            a = glh::mul(a, w);
            c = glh::mul(c, wi);
        }
        if (hipMalloc(&ctx->w12_fwd, 2048 * 8) != hipSuccess || hipMalloc(&ctx->w12_inv, 2048 * 8) != hipSuccess ||
            hipMemcpy(ctx->w12_fwd, f.data(), 2048 * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(ctx->w12_inv, b.data(), 2048 * 8, hipMemcpyHostToDevice) != hipSuccess)
            rc = VX_ERR_DEVICE;
    }
    if (rc == VX_OK) {
        ctx->pinned_n = 1 << 20;
        if (hipHostMalloc(&ctx->pinned, ctx->pinned_n) != hipSuccess) rc = VX_ERR_DEVICE;
    }
    if (rc != VX_OK) {
        delete ctx;
        return rc;
    }
    *out = ctx;
    return VX_OK;
}

int32_t vx_ctx_destroy(vx_ctx* ctx) {
    if (!ctx) return VX_ERR_ARG;
    if (ctx->side) {
        vx_ctx_destroy(ctx->side);
        ctx->side = nullptr;
    }
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->shift_tabs) hipFree(kv.second.d);
    for (auto& kv : ctx->periodic_cache) hipFree(kv.second);
    for (auto& kv : ctx->tw2) hipFree(kv.second);
    hipFree(ctx->tw_fwd.d);
    hipFree(ctx->tw_inv.d);
    hipFree(ctx->w12_fwd);
    hipFree(ctx->w12_inv);
    if (ctx->scratch) vx_pool_free(ctx, ctx->scratch);
    vx_pool_trim(ctx);
    for (auto& kv : ctx->pool_live) hipFree(kv.first);
    if (ctx->pinned) hipHostFree(ctx->pinned);
    hipEventDestroy(ctx->ev0);
    hipEventDestroy(ctx->ev1);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return VX_OK;
}

int32_t vx_sync(vx_ctx* ctx) {
    if (!ctx) return VX_ERR_ARG;
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}
const char* vx_last_error(const vx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int32_t vx_timer_start(vx_ctx* ctx) {
    if (!ctx) return VX_ERR_ARG;
    VX_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return VX_OK;
}
int32_t vx_timer_stop(vx_ctx* ctx, float* ms) {
    if (!ctx || !ms) return VX_ERR_ARG;
    VX_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    VX_HIP(hipEventSynchronize(ctx->ev1));
    VX_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return VX_OK;
}

int32_t vx_alloc(vx_ctx* ctx, size_t n, vx_buf** out) {
    if (!ctx || !out) return VX_ERR_ARG;
    VX_CHECK(n > 0, "vx_alloc: n == 0");
    vx_buf* b = new vx_buf{nullptr, n};
    b->d = (uint64_t*)vx_pool_alloc(ctx, n * 8);
    if (!b->d) {
        delete b;
        return vx_fail(ctx, VX_ERR_OOM, "device allocation of %zu bytes failed", n * 8);
    }
    *out = b;
    return VX_OK;
}
int32_t vx_free(vx_ctx* ctx, vx_buf* buf) {
    if (!ctx || !buf) return VX_ERR_ARG;
    vx_pool_free(ctx, buf->d);
    delete buf;
    return VX_OK;
}
int32_t vx_upload(vx_ctx* ctx, vx_buf* dst, size_t off, const uint64_t* src, size_t n) {
    if (!ctx || !dst || !src) return VX_ERR_ARG;
    VX_CHECK(off + n <= dst->n, "vx_upload: range [%zu,%zu) exceeds buffer of %zu", off, off + n, dst->n);
    VX_HIP(hipMemcpyAsync(dst->d + off, src, n * 8, hipMemcpyHostToDevice, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));  // caller may reuse src immediately
    return VX_OK;
}
int32_t vx_download(vx_ctx* ctx, const vx_buf* src, size_t off, uint64_t* dst, size_t n) {
    if (!ctx || !dst || !src) return VX_ERR_ARG;
    VX_CHECK(off + n <= src->n, "vx_download: range [%zu,%zu) exceeds buffer of %zu", off, off + n, src->n);
    VX_HIP(hipMemcpyAsync(dst, src->d + off, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}
int32_t vx_copy(vx_ctx* ctx, vx_buf* dst, size_t doff, const vx_buf* src, size_t soff, size_t n) {
    if (!ctx || !dst || !src) return VX_ERR_ARG;
    VX_CHECK(doff + n <= dst->n && soff + n <= src->n, "vx_copy: out of range");
    VX_HIP(hipMemcpyAsync(dst->d + doff, src->d + soff, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return VX_OK;
}
void* vx_buf_devptr(const vx_buf* buf) { return buf ? buf->d : nullptr; }
size_t vx_buf_len(const vx_buf* buf) { return buf ? buf->n : 0; }
}  // extern "C"

// ---------------------------------------------------------------- kernels
__global__ void k_fill_random(uint64_t* d, size_t n, uint64_t seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (i + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        d[i] = gl_canon(z);
    }
}
template <int OP>
__global__ void k_batch(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t x = a[i], y = OP == 3 ? 0 : b[i], r;
        if (OP == 0) r = gl_add(x, y);
        else if (OP == 1) r = gl_sub(x, y);
        else if (OP == 2) r = gl_mul(x, y);
        else r = x ? gl_inv(x) : 0;
        o[i] = r;
    }
}
__global__ void k_ext_mul(const uint64_t* a, const uint64_t* b, uint64_t* o, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        gl2 x{a[2 * i], a[2 * i + 1]}, y{b[2 * i], b[2 * i + 1]};
        gl2 r = gl2_mul(x, y);
        o[2 * i] = r.a;
        o[2 * i + 1] = r.b;
    }
}
static inline unsigned grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
template <int OP>
static int32_t batch_op(vx_ctx* ctx, const vx_buf* a, const vx_buf* b, vx_buf* o, size_t n) {
    if (!ctx || !a || !o || (OP != 3 && !b)) return VX_ERR_ARG;
    VX_CHECK(n <= a->n && n <= o->n && (OP == 3 || n <= b->n), "batch op: n exceeds a buffer");
    if (n == 0) return VX_OK;
    hipLaunchKernelGGL(k_batch<OP>, dim3(grid_for(n)), dim3(256), 0, ctx->stream, a->d, OP == 3 ? a->d : b->d, o->d, n);
    VX_HIP(hipGetLastError());
    return VX_OK;
}
extern "C" {
int32_t vx_fill_random(vx_ctx* ctx, vx_buf* dst, size_t off, size_t n, uint64_t seed) {
    if (!ctx || !dst) return VX_ERR_ARG;
    VX_CHECK(off + n <= dst->n, "vx_fill_random: out of range");
    if (n == 0) return VX_OK;
    hipLaunchKernelGGL(k_fill_random, dim3(grid_for(n)), dim3(256), 0, ctx->stream, dst->d + off, n, seed);
    VX_HIP(hipGetLastError());
    return VX_OK;
}
int32_t vx_field_batch_add(vx_ctx* c, const vx_buf* a, const vx_buf* b, vx_buf* o, size_t n) { return batch_op<0>(c, a, b, o, n); }
int32_t vx_field_batch_sub(vx_ctx* c, const vx_buf* a, const vx_buf* b, vx_buf* o, size_t n) { return batch_op<1>(c, a, b, o, n); }
int32_t vx_field_batch_mul(vx_ctx* c, const vx_buf* a, const vx_buf* b, vx_buf* o, size_t n) { return batch_op<2>(c, a, b, o, n); }
int32_t vx_field_batch_inv(vx_ctx* c, const vx_buf* a, vx_buf* o, size_t n) { return batch_op<3>(c, a, nullptr, o, n); }
int32_t vx_ext_batch_mul(vx_ctx* ctx, const vx_buf* a, const vx_buf* b, vx_buf* o, size_t n) {
    if (!ctx || !a || !b || !o) return VX_ERR_ARG;
    VX_CHECK(2 * n <= a->n && 2 * n <= b->n && 2 * n <= o->n, "ext batch mul: n exceeds a buffer");
    if (n == 0) return VX_OK;
    hipLaunchKernelGGL(k_ext_mul, dim3(grid_for(n)), dim3(256), 0, ctx->stream, a->d, b->d, o->d, n);
    VX_HIP(hipGetLastError());
    return VX_OK;
}

// One all-gather of fixed-size proof blobs.  RCCL is bound at call time (dlsym) from whatever copy is
// already loaded in the process -- the one the caller's communicator was made by.
int32_t vx_gather_proofs(vx_ctx* ctx, void* nccl_comm, int world, const uint64_t* mine, size_t n_words, uint64_t* out) {
    if (!ctx || !nccl_comm || !mine || !out || world < 1) return VX_ERR_ARG;
    typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
    static all_gather_fn all_gather = nullptr;
    if (!all_gather) all_gather = (all_gather_fn)dlsym(RTLD_DEFAULT, "ncclAllGather");
    if (!all_gather) {  // loaded with RTLD_LOCAL (a Python extension's dependency): find that copy by its soname
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            void* h = dlopen(name, RTLD_NOLOAD | RTLD_LAZY);
            if (h && (all_gather = (all_gather_fn)dlsym(h, "ncclAllGather"))) break;
        }
    }
    if (!all_gather) return vx_fail(ctx, VX_ERR_DEVICE, "gather: ncclAllGather is not loaded in this process");
    if (n_words == 0) return VX_OK;
    uint64_t* sc;
    VX_TRY(vx_scratch(ctx, n_words * ((size_t)world + 1), &sc));
    VX_HIP(hipMemcpyAsync(sc, mine, n_words * 8, hipMemcpyHostToDevice, ctx->stream));
    const int rc = all_gather(sc, sc + n_words, n_words, /*ncclUint64=*/5, nccl_comm, ctx->stream);
    if (rc != 0) return vx_fail(ctx, VX_ERR_DEVICE, "gather: ncclAllGather failed (%d)", rc);
    VX_HIP(hipMemcpyAsync(out, sc + n_words, n_words * 8 * (size_t)world, hipMemcpyDeviceToHost, ctx->stream));
    VX_HIP(hipStreamSynchronize(ctx->stream));
    return VX_OK;
}
}

vx_ctx* vx_side_ctx(vx_ctx* ctx) {
    if (!ctx->side) {
        g_side_ctx = 1;
        if (vx_ctx_create(ctx->device, &ctx->side) != VX_OK) ctx->side = nullptr;
        g_side_ctx = 0;
    }
    return ctx->side;
}
