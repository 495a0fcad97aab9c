// Load-time re-layout of the reference's checkpoint tensors (NVIDIA mma fragment order,
// reference: nunchaku/lora/flux/packer.py:187-301,362-437; device consumers gemm_base.cuh:265-355,
// lora.cuh:43-59) into the CDNA4 orders used by quantize.hip / gemm_w4a4.hip.
// Pure permutations, HBM-bound, run once per parameter at model load.
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <vector>

#include "svdq_common.h"

namespace svdq {

// ---- launch profiler ------------------------------------------------------------------------
struct ProfRec { hipEvent_t e0, e1; int cls; double work; };
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof_pool;
static size_t g_prof_used = 0;
static bool g_prof_on = false;

int prof_begin(int cls, double work, hipStream_t st) {
    if (!g_prof_on) return -1;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on || g_prof_used >= g_prof_pool.size()) return -1;
    int i = (int)g_prof_used++;
    g_prof_pool[i].cls = cls;
    g_prof_pool[i].work = work;
    hipEventRecord(g_prof_pool[i].e0, st);
    return i;
}
void prof_end(int i, hipStream_t st) {
    if (i >= 0) hipEventRecord(g_prof_pool[i].e1, st);
}

// ---- thread-local error string -----------------------------------------------------------
static thread_local char g_err[512] = {0};

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_check(hipError_t e, const char *what) {
    if (e == hipSuccess) return SVDQ_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return SVDQ_E_HIP;
}

// One thread per OUTPUT dword.  8 consecutive k (aligned to 8) of one output channel are one
// 32-bit word in the reference order too (packer.py:228-233: reg_k = 8 nibbles, low first), so
// the re-layout is a dword permutation.
__global__ void repack_qweight_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int N, int G) {
    size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)N * G * 8; // dwords
    if (d >= total) return;
    // decode T16 dword index: ((((rb*G + g)*8 + rt)*64 + lane)*2 + h)
    int h = d & 1;
    int lane = (d >> 1) & 63;
    int rt = (d >> 7) & 7;
    size_t q = d >> 10;
    int g = q % G;
    int rb = q / G;
    int rl = lane & 15, ks = lane >> 4;
    // reference word: (((nt*KT + kt)*8 + np)*32 + lane_ref)*4 + j
    int n_pack = rl >> 3, n_lane = rl & 7;
    int k_pack = ks >> 1, k_lane = (ks & 1) * 2 + h;
    int lane_ref = n_lane * 4 + k_lane;
    int j = n_pack * 2 + k_pack;
    size_t s = ((((size_t)rb * G + g) * 8 + rt) * 32 + lane_ref) * 4 + j;
    dst[d] = src[s];
}

// packed position inside a 128-channel block of logical channel c (inverse of packer.py:272-278)
__host__ __device__ __forceinline__ int scale_pos128(int c) {
    int a = c >> 4, rem = c & 15;
    int b = rem >> 3, rem8 = rem & 7;
    int cc = rem8 >> 1, dd = rem8 & 1;
    int lane = a * 4 + cc, e = b * 2 + dd;
    return lane * 4 + e;
}

__global__ void repack_wscales_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int G, int N) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)G * N) return;
    int g = i / N, n = i % N;
    int nt = n >> 7;
    dst[i] = src[((size_t)nt * G + g) * 128 + scale_pos128(n & 127)];
}

// 16x16 tiles in mma m16n8k16 fragment order: flat = ((cp*RP + rp)*32 + lane)*8 + h
__device__ __forceinline__ size_t lowrank_src(int c_tile, int r_tile, int RP, int n16, int k16) {
    int nps = n16 >> 3, nl = n16 & 7;
    int kps = k16 >> 3, k8 = k16 & 7;
    int kl = k8 >> 1, rk = k8 & 1;
    int lane = nl * 4 + kl;
    int h = (nps * 2 + kps) * 2 + rk;
    return (((size_t)c_tile * RP + r_tile) * 32 + lane) * 8 + h;
}

__global__ void repack_lowrank_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int C, int R,
                                      int down) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * R) return;
    int RP = R / 16;
    if (!down) {
        // dst natural [n][r]; fragment n-axis = n, k-axis = r
        int n = i / R, r = i % R;
        dst[i] = src[lowrank_src(n >> 4, r >> 4, RP, n & 15, r & 15)];
    } else {
        // dst rank-major [r][k]; fragment n-axis = r, k-axis = k, tiles ordered (k/16, r/16)
        int r = i / C, k = i % C;
        dst[i] = src[lowrank_src(k >> 4, r >> 4, RP, r & 15, k & 15)];
    }
}

// T16 packed codes -> one int8 per element, natural [M_pad, K] (test helper)
__global__ void unpack_act_kernel(const uint8_t *__restrict__ act, int8_t *__restrict__ codes, int M_pad, int K,
                                  int is_unsigned) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M_pad * K) return;
    int row = i / K, k = i % K;
    uint8_t b = act[t16_byte_offset(row, k, K / GROUP)];
    int v = (k & 1) ? (b >> 4) : (b & 15);
    if (!is_unsigned && v >= 8) v -= 16;
    codes[i] = (int8_t)v;
}

} // namespace svdq

using namespace svdq;

static inline unsigned nblk(size_t n, int t) { return (unsigned)((n + t - 1) / t); }

extern "C" {

int svdq_repack_qweight(const void *src, void *dst, int32_t N, int32_t K, void *stream) {
    if (!src || !dst || src == dst) { set_error("svdq_repack_qweight: null or aliasing pointers"); return SVDQ_E_INVALID; }
    if (N <= 0 || K <= 0 || N % 128 || K % 128) {
        set_error("svdq_repack_qweight: N=%d and K=%d must be positive multiples of 128", N, K);
        return SVDQ_E_INVALID;
    }
    size_t total = (size_t)N * (K / 64) * 8;
    hipLaunchKernelGGL(repack_qweight_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t *)src, (uint32_t *)dst, N, K / 64);
    return hip_check(hipGetLastError(), "svdq_repack_qweight launch");
}

int svdq_repack_wscales(const void *src, void *dst, int32_t G, int32_t N, void *stream) {
    if (!src || !dst || src == dst) { set_error("svdq_repack_wscales: null or aliasing pointers"); return SVDQ_E_INVALID; }
    if (G <= 0 || N <= 0 || N % 128) {
        set_error("svdq_repack_wscales: G=%d must be > 0 and N=%d a positive multiple of 128", G, N);
        return SVDQ_E_INVALID;
    }
    hipLaunchKernelGGL(repack_wscales_kernel, dim3(nblk((size_t)G * N, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)src, (uint16_t *)dst, G, N);
    return hip_check(hipGetLastError(), "svdq_repack_wscales launch");
}

int svdq_repack_vec(const void *src, void *dst, int32_t N, void *stream) {
    return svdq_repack_wscales(src, dst, 1, N, stream);
}

int svdq_repack_lowrank(const void *src, void *dst, int32_t C, int32_t R, int32_t down, void *stream) {
    if (!src || !dst || src == dst) { set_error("svdq_repack_lowrank: null or aliasing pointers"); return SVDQ_E_INVALID; }
    if (C <= 0 || R <= 0 || C % 16 || R % 16) {
        set_error("svdq_repack_lowrank: C=%d and R=%d must be positive multiples of 16", C, R);
        return SVDQ_E_INVALID;
    }
    hipLaunchKernelGGL(repack_lowrank_kernel, dim3(nblk((size_t)C * R, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)src, (uint16_t *)dst, C, R, down ? 1 : 0);
    return hip_check(hipGetLastError(), "svdq_repack_lowrank launch");
}

int svdq_unpack_act(const void *act, int8_t *codes, int32_t M_pad, int32_t K, int32_t is_unsigned, void *stream) {
    if (!act || !codes) { set_error("svdq_unpack_act: null pointer"); return SVDQ_E_INVALID; }
    if (M_pad <= 0 || K <= 0 || M_pad % 128 || K % 64) {
        set_error("svdq_unpack_act: M_pad=%d must be a multiple of 128 and K=%d of 64", M_pad, K);
        return SVDQ_E_INVALID;
    }
    hipLaunchKernelGGL(unpack_act_kernel, dim3(nblk((size_t)M_pad * K, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)act, codes, M_pad, K, is_unsigned);
    return hip_check(hipGetLastError(), "svdq_unpack_act launch");
}

int svdq_prof_enable(int32_t max_launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof_pool) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    g_prof_pool.clear();
    g_prof_used = 0;
    g_prof_on = false;
    if (max_launches <= 0) return SVDQ_OK;
    g_prof_pool.resize(max_launches);
    for (auto &r : g_prof_pool) {
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) {
            set_error("svdq_prof_enable: hipEventCreate failed");
            return SVDQ_E_HIP;
        }
    }
    g_prof_on = true;
    return SVDQ_OK;
}

int svdq_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_used = 0;
    return SVDQ_OK;
}

int svdq_prof_read(int32_t kernel_class, int64_t *launches, double *total_ms, double *total_work) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int64_t n = 0;
    double ms = 0, work = 0;
    for (size_t i = 0; i < g_prof_used; i++) {
        ProfRec &r = g_prof_pool[i];
        if (r.cls != kernel_class) continue;
        if (hipEventSynchronize(r.e1) != hipSuccess) { set_error("svdq_prof_read: hipEventSynchronize failed"); return SVDQ_E_HIP; }
        float t = 0;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) { set_error("svdq_prof_read: hipEventElapsedTime failed"); return SVDQ_E_HIP; }
        n++; ms += t; work += r.work;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = work;
    return SVDQ_OK;
}

const char *svdq_last_error(void) { return g_err; }
int svdq_abi_version(void) { return SVDQ_ABI_VERSION; }

} // extern "C"
