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
static uint32_t g_prof_mask = 0xffffffffu; // kernel classes that are bracketed (svdq_prof_select)

int prof_begin(int cls, double work, hipStream_t st) {
    if (!g_prof_on || !((g_prof_mask >> (cls & 0xff)) & 1u)) return -1; // (bits 8.. of cls: sub-class, e.g. the GEMM's epilogue variant)
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on || g_prof_used >= g_prof_pool.size()) return -1;
    int i = (int)g_prof_used++;
    g_prof_pool[i].cls = cls;
    g_prof_pool[i].work = work;
    (void)hipEventRecord(g_prof_pool[i].e0, st);
    return i;
}
void prof_end(int i, hipStream_t st) {
    if (i >= 0) (void)hipEventRecord(g_prof_pool[i].e1, st);
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

// One thread per OUTPUT (lane record, group) = 32 codes = 24 bytes of the F6 image (svdq_common.h).
// In the reference order 8 consecutive k (aligned to 8) of one output channel are one 32-bit word
// (packer.py:228-233: 8 nibbles, low first); the 4 codes e = 0..3 of F6 element j = 16t + 4c + e are
// the nibbles 4h .. 4h+3 of the word that holds k = 64g + 32t + 8c .. +7.
__global__ void repack_qweight_kernel(const uint32_t *__restrict__ src, uint8_t *__restrict__ dst, int N, int KP) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)(N / 32) * KP * 2 * 64;
    if (i >= total) return;
    const int lane = i & 63;
    const int grp = (i >> 6) & 1;
    size_t q = i >> 7;
    const int kp = q % KP;
    const int rt = q / KP;
    const int n = rt * 32 + (lane & 31), h = lane >> 5;
    const int g = kp * 2 + grp, KT = KP * 2;
    // reference word: (((nt*KT + kt)*8 + np)*32 + lane_ref)*4 + jj  (oracle: _qweight_index)
    const int nt = n >> 7, npk = (n & 127) >> 4, n_pack = (n & 15) >> 3, n_lane = n & 7;
    uint32_t d[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int lane_ref = n_lane * 4 + c, jj = n_pack * 2 + t;
            const uint32_t w = src[((((size_t)nt * KT + g) * 8 + npk) * 32 + lane_ref) * 4 + jj];
            const uint32_t four = (w >> (16 * h)) & 0xFFFF;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                int v = (four >> (4 * e)) & 15;
                if (v >= 8) v -= 16;
                const unsigned code = f6_enc_s4(v);
                const int bit = 6 * (16 * t + 4 * c + e);
                d[bit >> 5] |= code << (bit & 31);
                if ((bit & 31) > 26) d[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
            }
        }
    uint8_t *rec = dst + ((size_t)rt * KP + kp) * F6_CHUNK + (size_t)lane * 16;
    if (grp == 0) {
        *reinterpret_cast<uint4 *>(rec) = make_uint4(d[0], d[1], d[2], d[3]);
        *reinterpret_cast<uint2 *>(rec + F6_PLANE) = make_uint2(d[4], d[5]);
    } else {
        *reinterpret_cast<uint2 *>(rec + F6_PLANE + 8) = make_uint2(d[0], d[1]);
        *reinterpret_cast<uint4 *>(rec + 2 * F6_PLANE) = make_uint4(d[2], d[3], d[4], d[5]);
    }
}

// packed position inside a 128-channel block of logical channel c (inverse of packer.py:272-278)
__host__ __device__ __forceinline__ int scale_pos128(int c) {
    int a = c >> 4, rem = c & 15;
    int b = rem >> 3, rem8 = rem & 7;
    int cc = rem8 >> 1, dd = rem8 & 1;
    int lane = a * 4 + cc, e = b * 2 + dd;
    return lane * 4 + e;
}

// reference [g][n]-packed -> natural [g][n] (SIMG == 0; bias / smooth vectors with G == 1) or the
// S image [N/32][G/2][2][32] the GEMM stages per K-step (SIMG == 1)
// The weight-side S image holds 32 x the scale (round 6, ABI 21): the GEMM's product MFMA runs WITHOUT MX block scales (one instruction instead of the
// v_mfma_ld_scale + MFMA pair; P = dot / 64 since both code images hold value / 8), the scale tile is S = 2 ws' as, so ws' = 32 ws makes P S = dot ws as -- the
// same fp32 product bit for bit (a power of two moves between the factors).  Exact for every bf16 scale; an fp16 scale above 2047 overflows (weights beyond
// 14 000: nunchaku_amd/layout.py refuses them at load).
__device__ __forceinline__ uint16_t wscale_times(uint16_t v, int dt, float f) {
    if (dt == SVDQ_BF16) return __builtin_bit_cast(uint16_t, (__bf16)(__builtin_bit_cast(float, (uint32_t)v << 16) * f));
    return __builtin_bit_cast(uint16_t, (_Float16)((float)__builtin_bit_cast(_Float16, v) * f));
}
template <int SIMG>
__global__ void repack_wscales_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int G, int N, int dt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)G * N) return;
    int g = i / N, n = i % N;
    int nt = n >> 7;
    const uint16_t v = src[((size_t)nt * G + g) * 128 + scale_pos128(n & 127)];
    if (SIMG) dst[simg_index(n, g, G / 2)] = wscale_times(v, dt, 32.0f);
    else dst[i] = v;
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

// ---- inverse re-layouts: kernel order -> the reference's checkpoint order (state_dict() of a repacked layer, host offload) ----
// One thread per 32-bit word of the reference qweight (8 consecutive k of one output channel, low nibble first): its
// nibbles 0..3 are elements j = 16t + 4c + e of the h = 0 lane record of (n, group g), nibbles 4..7 those of the h = 1 record.
__global__ void unrepack_qweight_kernel(const uint8_t *__restrict__ img, uint32_t *__restrict__ dst, int N, int KP) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int KT = KP * 2;
    if (i >= (size_t)(N / 128) * KT * 8 * 32 * 4) return;
    const int jj = i & 3, lane_ref = (i >> 2) & 31, npk = (i >> 7) & 7;
    const size_t q = i >> 10;
    const int g = q % KT, nt = q / KT;
    const int n_lane = lane_ref >> 2, c = lane_ref & 3, n_pack = jj >> 1, t = jj & 1;
    const int n = nt * 128 + npk * 16 + n_pack * 8 + n_lane;
    const int kp = g >> 1, grp = g & 1;
    uint32_t w = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint8_t *rec = img + ((size_t)(n >> 5) * KP + kp) * F6_CHUNK + (size_t)((n & 31) | (h << 5)) * 16;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int bit = 192 * grp + 6 * (16 * t + 4 * c + e);
            const int b = bit >> 3;
            unsigned v = rec[(b >> 4) * F6_PLANE + (b & 15)];
            if (b < 47) v |= (unsigned)rec[((b + 1) >> 4) * F6_PLANE + ((b + 1) & 15)] << 8;
            const int code = f6_dec((v >> (bit & 7)) & 63, 0);
            w |= (uint32_t)(code & 15) << (4 * (4 * h + e));
        }
    }
    dst[i] = w;
}

// natural [g][n] (SIMG == 0) or the S image (SIMG == 1) -> the reference's [g][n]-packed order
template <int SIMG>
__global__ void unrepack_wscales_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int G, int N, int dt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)G * N) return;
    int g = i / N, n = i % N;
    int nt = n >> 7;
    dst[((size_t)nt * G + g) * 128 + scale_pos128(n & 127)] = SIMG ? wscale_times(src[simg_index(n, g, G / 2)], dt, 0.03125f) : src[i];
}

__global__ void unrepack_lowrank_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int C, int R, int down) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * R) return;
    int RP = R / 16;
    if (!down) {
        int n = i / R, r = i % R;
        dst[lowrank_src(n >> 4, r >> 4, RP, n & 15, r & 15)] = src[i];
    } else {
        int r = i / C, k = i % C;
        dst[lowrank_src(k >> 4, r >> 4, RP, r & 15, k & 15)] = src[i];
    }
}

// F6 image -> one int8 per element, natural [ROWS, K] (test helper)
__global__ void unpack_act_kernel(const uint8_t *__restrict__ act, int8_t *__restrict__ codes, int M_pad, int K,
                                  int is_unsigned) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M_pad * K) return;
    int row = i / K, k = i % K;
    const size_t base = f6_record_base(row, k, K / 128);
    const int bit = f6_record_bit(k);
    unsigned v = act[f6_byte(base, bit >> 3)] | ((unsigned)act[f6_byte(base, (bit >> 3) + ((bit >> 3) < 47 ? 1 : 0))] << 8);
    codes[i] = (int8_t)f6_dec((v >> (bit & 7)) & 63, is_unsigned);
}

// S image -> natural [G][ROWS] (test helper)
__global__ void unpack_scales_kernel(const uint16_t *__restrict__ simg, uint16_t *__restrict__ nat, int ROWS, int G) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ROWS * G) return;
    int g = i / ROWS, row = i % ROWS;
    nat[i] = simg[simg_index(row, g, G / 2)];
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
    size_t total = (size_t)(N / 32) * (K / 128) * 2 * 64;
    hipLaunchKernelGGL(repack_qweight_kernel, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t *)src, (uint8_t *)dst, N, K / 128);
    return hip_check(hipGetLastError(), "svdq_repack_qweight launch");
}

int svdq_repack_wscales(const void *src, void *dst, int32_t G, int32_t N, int32_t dtype, void *stream) {
    if (!src || !dst || src == dst) { set_error("svdq_repack_wscales: null or aliasing pointers"); return SVDQ_E_INVALID; }
    if (dtype != SVDQ_BF16 && dtype != SVDQ_FP16) { set_error("svdq_repack_wscales: dtype must be SVDQ_BF16 or SVDQ_FP16"); return SVDQ_E_INVALID; }
    if (G <= 0 || G % 2 || N <= 0 || N % 128) {
        set_error("svdq_repack_wscales: G=%d must be a positive even number and N=%d a positive multiple of 128", G, N);
        return SVDQ_E_INVALID;
    }
    hipLaunchKernelGGL(repack_wscales_kernel<1>, dim3(nblk((size_t)G * N, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)src, (uint16_t *)dst, G, N, dtype);
    return hip_check(hipGetLastError(), "svdq_repack_wscales launch");
}

int svdq_repack_vec(const void *src, void *dst, int32_t N, void *stream) {
    if (!src || !dst || src == dst) { set_error("svdq_repack_vec: null or aliasing pointers"); return SVDQ_E_INVALID; }
    if (N <= 0 || N % 128) { set_error("svdq_repack_vec: N=%d must be a positive multiple of 128", N); return SVDQ_E_INVALID; }
    hipLaunchKernelGGL(repack_wscales_kernel<0>, dim3(nblk((size_t)N, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)src, (uint16_t *)dst, 1, N, 0);
    return hip_check(hipGetLastError(), "svdq_repack_vec launch");
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

int svdq_unrepack_qweight(const void *src, void *dst, int32_t N, int32_t K, void *stream) {
    if (!src || !dst || src == dst) { set_error("svdq_unrepack_qweight: null or aliasing pointers"); return SVDQ_E_INVALID; }
    if (N <= 0 || K <= 0 || N % 128 || K % 128) {
        set_error("svdq_unrepack_qweight: N=%d and K=%d must be positive multiples of 128", N, K);
        return SVDQ_E_INVALID;
    }
    const size_t words = (size_t)N * K / 8;
    hipLaunchKernelGGL(unrepack_qweight_kernel, dim3(nblk(words, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)src, (uint32_t *)dst, N, K / 128);
    return hip_check(hipGetLastError(), "svdq_unrepack_qweight launch");
}

int svdq_unrepack_wscales(const void *src, void *dst, int32_t G, int32_t N, int32_t dtype, void *stream) {
    if (!src || !dst || src == dst) { set_error("svdq_unrepack_wscales: null or aliasing pointers"); return SVDQ_E_INVALID; }
    if (dtype != SVDQ_BF16 && dtype != SVDQ_FP16) { set_error("svdq_unrepack_wscales: dtype must be SVDQ_BF16 or SVDQ_FP16"); return SVDQ_E_INVALID; }
    if (G <= 0 || G % 2 || N <= 0 || N % 128) {
        set_error("svdq_unrepack_wscales: G=%d must be a positive even number and N=%d a positive multiple of 128", G, N);
        return SVDQ_E_INVALID;
    }
    hipLaunchKernelGGL(unrepack_wscales_kernel<1>, dim3(nblk((size_t)G * N, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)src, (uint16_t *)dst, G, N, dtype);
    return hip_check(hipGetLastError(), "svdq_unrepack_wscales launch");
}

int svdq_unrepack_vec(const void *src, void *dst, int32_t N, void *stream) {
    if (!src || !dst || src == dst) { set_error("svdq_unrepack_vec: null or aliasing pointers"); return SVDQ_E_INVALID; }
    if (N <= 0 || N % 128) { set_error("svdq_unrepack_vec: N=%d must be a positive multiple of 128", N); return SVDQ_E_INVALID; }
    hipLaunchKernelGGL(unrepack_wscales_kernel<0>, dim3(nblk((size_t)N, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)src, (uint16_t *)dst, 1, N, 0);
    return hip_check(hipGetLastError(), "svdq_unrepack_vec launch");
}

int svdq_unrepack_lowrank(const void *src, void *dst, int32_t C, int32_t R, int32_t down, void *stream) {
    if (!src || !dst || src == dst) { set_error("svdq_unrepack_lowrank: null or aliasing pointers"); return SVDQ_E_INVALID; }
    if (C <= 0 || R <= 0 || C % 16 || R % 16) {
        set_error("svdq_unrepack_lowrank: C=%d and R=%d must be positive multiples of 16", C, R);
        return SVDQ_E_INVALID;
    }
    hipLaunchKernelGGL(unrepack_lowrank_kernel, dim3(nblk((size_t)C * R, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)src, (uint16_t *)dst, C, R, down ? 1 : 0);
    return hip_check(hipGetLastError(), "svdq_unrepack_lowrank launch");
}

int svdq_unpack_act(const void *act, int8_t *codes, int32_t M_pad, int32_t K, int32_t is_unsigned, void *stream) {
    if (!act || !codes) { set_error("svdq_unpack_act: null pointer"); return SVDQ_E_INVALID; }
    if (M_pad <= 0 || K <= 0 || M_pad % 32 || K % 128) {
        set_error("svdq_unpack_act: M_pad=%d must be a multiple of 32 and K=%d of 128", M_pad, K);
        return SVDQ_E_INVALID;
    }
    hipLaunchKernelGGL(unpack_act_kernel, dim3(nblk((size_t)M_pad * K, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)act, codes, M_pad, K, is_unsigned);
    return hip_check(hipGetLastError(), "svdq_unpack_act launch");
}

int svdq_unpack_scales(const void *simg, void *natural, int32_t ROWS, int32_t G, void *stream) {
    if (!simg || !natural || simg == natural) { set_error("svdq_unpack_scales: null or aliasing pointers"); return SVDQ_E_INVALID; }
    if (ROWS <= 0 || G <= 0 || ROWS % 32 || G % 2) {
        set_error("svdq_unpack_scales: ROWS=%d must be a multiple of 32 and G=%d even", ROWS, G);
        return SVDQ_E_INVALID;
    }
    hipLaunchKernelGGL(unpack_scales_kernel, dim3(nblk((size_t)ROWS * G, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)simg, (uint16_t *)natural, ROWS, G);
    return hip_check(hipGetLastError(), "svdq_unpack_scales launch");
}

int svdq_prof_enable(int32_t max_launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof_pool) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
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

int svdq_prof_select(uint32_t class_mask) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_mask = class_mask;
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
        if (kernel_class < 256 ? (r.cls & 0xff) != kernel_class : r.cls != kernel_class) continue; // a class with all its sub-classes, or one sub-class
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
