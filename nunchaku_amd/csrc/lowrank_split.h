// Split low-rank down projection: the kernels and the host launcher shared by svdq_gemm_w4a4 (GELU_QUANT: the next layer's branch) and svdq_attention (the
// fused quantiser of the output projection).  Included by gemm_w4a4.hip and attention.hip; part of the hashed kernel sources (bench.kernel_sources_sha16).
#pragma once
#include "svdq_common.h"

namespace svdq {

// ---- split low-rank down (next-layer rank 48 .. 160 of a GELU_QUANT launch, rank 48 .. 160 of the attention epilogue's quantiser; DESIGN.md 5 "Round 5") --------------------------------------------------------------
// The next layer's low-rank down projection D'[m][r] = sum_n g[m][n] * ld[n][r] (lora.cuh:243-353, launch_impl.cuh:226-262) is a GEMM over the WHOLE output row of
// the launch; inside a 128-column tile every workgroup holds a 1/96 partial of it, which beyond 32 ranks neither fits an LDS carry on 256 x 128 tiles nor is cheap as
// per-tile fp32 atomics (rank 128: as much as the rank-32 launch itself; the solo-carry kernel buys the LDS with one wave per SIMD).  Split: the epilogue stores the
// 16-bit GELU output it already holds as MFMA A-operand fragments (one coalesced 16-byte store per lane and 16 columns -- the bytes of a default epilogue's store),
// and this kernel streams that image once: 64 rows x all ranks per wave, K split over the four waves of a workgroup (summed through LDS in a fixed order) and over
// `ks` workgroups (fp32 atomics, ks x M_pad x R2 of them where the tiles issued N / 128 x as many).
//   act16:  [M_pad / 32 row tiles][N / 16 units][64 lanes][8]   lane (row & 31, h), slot j <- column 16 u + 8 (j >> 2) + 4 h + (j & 3)   (the C layout's own order)
//   ldp:    [N / 16 units][NB rank blocks][64 lanes][8]         lane (rank & 31, h), slot j <- the same column of rank 32 b + (lane & 31); ranks >= R2: zeros
template <int DT>
__global__ __launch_bounds__(256) void pack_lora_down_kernel(const typename Half<DT>::T *__restrict__ ld /* rank-major [R2][N] */, typename Half<DT>::V8 *__restrict__ out,
                                                             int N, int R2, int nb) {
    using T = typename Half<DT>::T;
    using V8 = typename Half<DT>::V8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = blockIdx.x * 4 + wave; // (unit, block)
    if (idx >= (N / 16) * nb) return;
    const int unit = idx / nb, b = idx % nb, rank = b * 32 + (lane & 31), h = lane >> 5;
    V8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = (T)0.f;
    if (rank < R2) {
        const T *src = ld + (size_t)rank * N + unit * 16 + h * 4;
        const u16x4 w0 = *reinterpret_cast<const u16x4 *>(src), w1 = *reinterpret_cast<const u16x4 *>(src + 8);
#pragma unroll
        for (int j = 0; j < 4; j++) { o[j] = hfrom<T>(w0[j]); o[4 + j] = hfrom<T>(w1[j]); }
    }
    out[(size_t)idx * 64 + lane] = o;
}

template <int DT, int NB>
__global__ __launch_bounds__(256, 2) void lowrank_down_split_kernel(const typename Half<DT>::V8 *__restrict__ a16, const typename Half<DT>::V8 *__restrict__ ldp,
                                                                     const typename Half<DT>::V8 *__restrict__ ldp2, float *__restrict__ out, int split_row,
                                                                     int units_n, int R2, int ks) {
    using V8 = typename Half<DT>::V8;
    // UN units per step and wave, the fragments double-buffered (below).  Two workgroups per CU: 2 NB x 16 accumulators + the fragment sets in <= 256 registers per
    // lane; eight waves' loads in flight carry the bandwidth, the matrix pipe is ~ 20 % busy.
    constexpr int UN = 2;
    __shared__ v4f red[2 * NB * 4 * 64]; // one wave's accumulators: [(mi, block)][4 register quads][64 lanes]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rg = blockIdx.x / ks, sl = blockIdx.x % ks; // 64-row group, K slice
    const int per = units_n / ks;                         // (host: a multiple of 4 * UN)
    // uniform byte pointers + ONE 32-bit lane offset: every load takes the saddr form
    const char *pb = (const char *)(rg * 64 >= split_row ? ldp2 : ldp);
    const char *pa0 = (const char *)a16 + (size_t)(rg * 2) * units_n * 1024, *pa1 = pa0 + (size_t)units_n * 1024;
    const unsigned lo = (unsigned)lane * 16u;
    v16f d[2][NB];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) d[mi][b][i] = 0.f;
    // Fragment sets: activations a[k][row tile], weights w[k][rank block].  NB <= 4 (DEEP): BOTH are requested one step ahead -- a wave always has a whole step
    // (UN x (2 + NB) KiB) in flight, the wait in front of a step's MFMAs finds its operands landed.  NB = 5 has no registers for a second weight set (2 x 16 x 10
    // accumulators + 2 x 56 > 256): its weights are requested at the top of their step (the L2 round trip shows per step; rank 144 / 160 only).
    constexpr bool DEEP = NB <= 4;
    struct Frags { V8 a[UN][2], w[UN][NB]; };
    auto load_a = [&](Frags &f, int u) {
        const char *qa0 = pa0 + (size_t)u * 1024, *qa1 = pa1 + (size_t)u * 1024;
#pragma unroll
        for (int k = 0; k < UN; k++) {
            f.a[k][0] = *reinterpret_cast<const V8 *>(qa0 + k * 1024 + lo);
            f.a[k][1] = *reinterpret_cast<const V8 *>(qa1 + k * 1024 + lo);
        }
    };
    auto load_w = [&](Frags &f, int u) {
        const char *qb = pb + (size_t)u * (NB * 1024);
#pragma unroll
        for (int k = 0; k < UN; k++)
#pragma unroll
            for (int b = 0; b < NB; b++) f.w[k][b] = *reinterpret_cast<const V8 *>(qb + (k * NB + b) * 1024 + lo);
    };
    // one step: every load is issued before the first MFMA (the scheduling barriers keep the compiler from sinking loads between the MFMAs to save registers:
    // it would leave two loads in flight per wave)
    auto step = [&](Frags &f, Frags &next, int u, int un) {
        if constexpr (DEEP) load_w(next, un);
        else load_w(f, u);
        load_a(next, un);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < UN; k++)
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int mi = 0; mi < 2; mi++) d[mi][b] = Half<DT>::mfma32(f.a[k][mi], f.w[k][b], d[mi][b]);
        __builtin_amdgcn_sched_barrier(0);
    };
    const int steps = per / (4 * UN); // per wave, uniform, even (host: a slice is a multiple of 16 units)
    int u = sl * per + wave * UN;
    Frags x, y;
    load_a(x, u);
    if constexpr (DEEP) load_w(x, u);
    for (int i = 0; i < steps; i += 2) {
        const int u1 = u + 4 * UN, u2 = i + 2 < steps ? u1 + 4 * UN : u1; // (the last step requests its own fragments again: no branch around the loads)
        step(x, y, u, u1);
        step(y, x, u1, u2);
        u = u2;
    }
    // the four waves' partial sums, added in a fixed order through LDS (wave 3 writes, 2, 1, 0 add), then a quarter of the atomics per wave
    v4f *mine = red + lane;
    if (wave == 3) {
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int g = 0; g < 4; g++) mine[((mi * NB + b) * 4 + g) * 64] = v4f{d[mi][b][4 * g], d[mi][b][4 * g + 1], d[mi][b][4 * g + 2], d[mi][b][4 * g + 3]};
    }
    __syncthreads();
#pragma unroll 1
    for (int turn = 2; turn >= 0; turn--) {
        if (wave == turn) {
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
#pragma unroll
                for (int b = 0; b < NB; b++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        v4f *slot = mine + ((mi * NB + b) * 4 + g) * 64;
                        const v4f o = *slot;
                        *slot = v4f{d[mi][b][4 * g] + o[0], d[mi][b][4 * g + 1] + o[1], d[mi][b][4 * g + 2] + o[2], d[mi][b][4 * g + 3] + o[3]};
                    }
        }
        __syncthreads();
    }
    const int lr = lane & 31, h = lane >> 5;
    for (int pr = wave; pr < 2 * NB; pr += 4) { // pair (mi, block)
        const int mi = pr / NB, b = pr % NB;
        if (b * 32 + lr < R2) {
            float *dst = out + (size_t)(rg * 64 + mi * 32 + h * 4) * R2 + b * 32 + lr;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const v4f v = mine[(pr * 4 + g) * 64];
#pragma unroll
                for (int e = 0; e < 4; e++) unsafeAtomicAdd(dst + (size_t)(e + 8 * g) * R2, v[e]); // register 4 g + e = row 8 g + 4 h + e of the row tile
            }
        }
    }
}

// bytes the packed down projection(s) of a launch take ([N / 16 units][nb blocks][64 lanes][8] 16-bit per weight set)
static inline long long lowrank_split_pack_bytes(int N, int R2, bool two_sets) { return (two_sets ? 2LL : 1LL) * (N / 16) * ((R2 + 31) / 32) * 1024; }
static inline bool lowrank_split_shape_ok(int N, int R2) { return R2 > 32 && (R2 + 31) / 32 <= 5 && N % 256 == 0; }

// pack the down projection(s) into `ldp` (lowrank_split_pack_bytes), then -- `producer()` enqueues the kernel that writes the act16 image -- contract:
// out[M_pad][R2] += act16 . ld.  ld / ld2: rank-major [R2][N] 16-bit; rows >= split_row use ld2 (0x7fffffff: one set).  cus: compute units the grid is sized for.
template <int DT, typename Producer>
static void launch_lowrank_down_split(const void *act16, const void *ld, const void *ld2, int split_row, int M_pad, int N, int R2, float *out, void *ldp_scratch,
                                      int cus, hipStream_t st, Producer producer, const void *ldp_pre = nullptr, const void *ldp2_pre = nullptr) {
    using V8 = typename Half<DT>::V8;
    using T = typename Half<DT>::T;
    const int nb = (R2 + 31) / 32, units_n = N / 16, rgs = M_pad / 64;
    const V8 *ldp = (const V8 *)ldp_scratch, *ldp2 = ldp + (size_t)units_n * nb * 64;
    const dim3 pg((units_n * nb + 3) / 4), pb(256);
    // (ABI 21: a caller that keeps the images -- svdq_pack_lora_down, once per parameter -- passes them and the per-launch pack is skipped)
    if (ldp_pre) ldp = (const V8 *)ldp_pre;
    else hipLaunchKernelGGL((pack_lora_down_kernel<DT>), pg, pb, 0, st, (const T *)ld, (V8 *)ldp_scratch, N, R2, nb);
    if (ld2 && split_row < M_pad) {
        if (ldp2_pre) ldp2 = (const V8 *)ldp2_pre;
        else hipLaunchKernelGGL((pack_lora_down_kernel<DT>), pg, pb, 0, st, (const T *)ld2, (V8 *)ldp_scratch + (size_t)units_n * nb * 64, N, R2, nb);
    }
    producer();
    // K split over workgroups: the largest divisor of N / 256 (a slice is then a multiple of the 16 units the four waves take in two steps) that
    // keeps the grid within two workgroups per CU -- one round, every slice streaming at once
    int ks = 1;
    for (int c = 1; c <= N / 256; c++)
        if ((N / 256) % c == 0 && (long long)rgs * c <= 2LL * cus) ks = c;
    const dim3 sg(rgs * ks), sb(256);
    const V8 *a16 = (const V8 *)act16;
    switch (nb) {
    case 2: hipLaunchKernelGGL((lowrank_down_split_kernel<DT, 2>), sg, sb, 0, st, a16, ldp, ldp2, out, split_row, units_n, R2, ks); break;
    case 3: hipLaunchKernelGGL((lowrank_down_split_kernel<DT, 3>), sg, sb, 0, st, a16, ldp, ldp2, out, split_row, units_n, R2, ks); break;
    case 4: hipLaunchKernelGGL((lowrank_down_split_kernel<DT, 4>), sg, sb, 0, st, a16, ldp, ldp2, out, split_row, units_n, R2, ks); break;
    default: hipLaunchKernelGGL((lowrank_down_split_kernel<DT, 5>), sg, sb, 0, st, a16, ldp, ldp2, out, split_row, units_n, R2, ks); break;
    }
}

// svdq_pack_lora_down (ABI 21): the image above for one weight set
template <int DT>
static void launch_pack_lora_down(const void *ld, void *out, int N, int R2, hipStream_t st) {
    const int nb = (R2 + 31) / 32, units_n = N / 16;
    hipLaunchKernelGGL((pack_lora_down_kernel<DT>), dim3((units_n * nb + 3) / 4), dim3(256), 0, st, (const typename Half<DT>::T *)ld, (typename Half<DT>::V8 *)out, N, R2, nb);
}

} // namespace svdq
