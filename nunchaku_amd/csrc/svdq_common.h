// Shared device/host helpers of the gfx950 SVDQuant kernels.
//
// Packed int4 "T16" tile order (both the activations produced by our quantiser and the weights
// after svdq_repack_qweight use it; DESIGN.md "Data layout in HBM"):
//
//   matrix [ROWS, K] of 4-bit codes, ROWS % 128 == 0, K % 64 == 0, G = K / 64
//   byte(row, k) = ((((row/128)*G + k/64)*8 + (row%128)/16)*64 + lane)*8 + (k%16)/2,
//   lane = ((k%64)/16)*16 + row%16,   low nibble = even k.
//
// i.e. one (16 rows x 64 k) MFMA operand tile is 512 contiguous bytes in wave-lane order: lane l
// owns 8 bytes = the 16 codes of row (l&15), k-slot (l>>4).  A (128 rows x 64 k) block is one
// contiguous 4 KiB chunk, so a workgroup's operand tile for one quantisation group is staged
// with fully coalesced 16-byte loads and read back from LDS with conflict-free ds_read_b64.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svdq_amd.h"

namespace svdq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

constexpr int GROUP = 64;   // int4 quantisation group (reference: gemm_base.cuh:89-95)
constexpr int ROWBLK = 128; // rows per contiguous T16 block

// ---- 16-bit model dtype traits -------------------------------------------------------------
template <int DT> struct Half;
template <> struct Half<SVDQ_BF16> {
    using T = __bf16;
    using V8 = bf16x8;
    static __device__ __forceinline__ v4f mfma(V8 a, V8 b, v4f c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Half<SVDQ_FP16> {
    using T = _Float16;
    using V8 = f16x8;
    static __device__ __forceinline__ v4f mfma(V8 a, V8 b, v4f c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

template <typename T> __device__ __forceinline__ float h2f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T f2h(float v) { return (T)v; } // RNE

template <typename T> __device__ __forceinline__ unsigned short hbits(T v) {
    return __builtin_bit_cast(unsigned short, v);
}
template <typename T> __device__ __forceinline__ T hfrom(unsigned short b) { return __builtin_bit_cast(T, b); }

// round a float to the 16-bit type and come back (the reference keeps its tile in 16-bit between
// epilogue stages; this reproduces those rounding points)
template <typename T> __device__ __forceinline__ float round16(float v) { return (float)(T)v; }

__host__ __device__ __forceinline__ size_t t16_byte_offset(int row, int k, int G) {
    int lane = ((k & 63) >> 4) * 16 + (row & 15);
    return ((((size_t)(row >> 7) * G + (k >> 6)) * 8 + ((row & 127) >> 4)) * 64 + lane) * 8 + ((k & 15) >> 1);
}

// ---- host-side error plumbing --------------------------------------------------------------
void set_error(const char *fmt, ...);
int hip_check(hipError_t e, const char *what);
// launch profiler hooks (repack.hip): returns a record index or -1 when profiling is off
int prof_begin(int cls, double work, hipStream_t st);
void prof_end(int rec, hipStream_t st);

} // namespace svdq
