// gemm128_probe (round 6, VERDICT r5 #2 i): the W4A4 GEMM with a 128 x 64 wave tile and ONE wave per SIMD, on hardware, against the product kernel.
//
// A stand-alone kernel around tools/gen_gemm_loop3.py's generated main loop: workgroup = 4 waves (2 along M x 2 along N) on a 256 x 128 output tile -- the
// product's tile, LDS stage image, operand images (F6 / S), XCD-aware persistent schedule (whole tiles, whole rounds) and cross-tile prefetch protocol --
// with the plain epilogue (no bias, no low-rank branch: the 16-bit store only), so that its output can be compared BIT FOR BIT with svdq_gemm_w4a4 of the
// product library on the same operands (R = 0, bias = NULL, geometry 1) and timed beside it on the same box.
//   build: python tools/gen_gemm_loop3.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Inunchaku_amd/csrc -Itools/ablate/gen -o tools/ablate/gemm128_probe tools/ablate/gemm128_probe.hip -ldl
//   run:   tools/ablate/gemm128_probe --shape 4608 3072 3072 [--fp16] [--grid G] [--lib path/to/libsvdq_amd.so]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "svdq_common.h"

#ifndef SVDQ_LOOP3_BF16
#define SVDQ_LOOP3_BF16 "gemm_loop3_bf16.inc"
#endif
#ifndef SVDQ_LOOP3_FP16
#define SVDQ_LOOP3_FP16 "gemm_loop3_fp16.inc"
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using namespace svdq;

#ifndef PROBE_NSTAGE
#define PROBE_NSTAGE 3
#endif
constexpr int BM = 256, BN = 128, NSTAGE = PROBE_NSTAGE; // 4: the generator's option "b2" (ring of four, a barrier every other K-step)
constexpr int A_BYTES = (BM / 32) * F6_CHUNK, W_BYTES = (BN / 32) * F6_CHUNK, AS_BYTES = 1024, WS_BYTES = 1024;
constexpr int STAGE_BYTES = A_BYTES + W_BYTES + AS_BYTES + WS_BYTES; // 38912

struct P128 {
    const uint8_t *act, *wgt;
    const void *ascales, *wscales;
    void *out;
    int M, M_pad, N, K, ldo;
    long long *clk; // per workgroup {shader cycles, 100 MHz ticks} or NULL
};

typedef __attribute__((address_space(3))) void lds_void;

template <int DT>
__global__ __launch_bounds__(256, 1) void gemm128_kernel(const P128 p) {
    using T = typename Half<DT>::T;
    __shared__ __attribute__((aligned(16))) uint8_t lds[NSTAGE * STAGE_BYTES];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, h = lane >> 5;
    const int KP = p.K / 128, TM = p.M_pad / BM, TN = p.N / BN, NT = TM * TN;
    const int G = gridDim.x;
    const int pos = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const long long clk_t0 = __builtin_readcyclecounter(), clk_r0 = wall_clock64();
    auto tile_coords = [&](int t, int &bm, int &bn) {
        const int strip = t / (8 * TM);
        const int w = min(8, TN - 8 * strip);
        const int r = t - strip * 8 * TM;
        bm = r / w;
        bn = 8 * strip + r % w;
    };
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_base = (unsigned)(size_t)(lds_void *)lds;
    // DMA roles: every wave the three planes of A chunks w and w + 4 and of W chunk w; wave 0 the activation scale image (8 x 128 B), wave 1 the weight
    // scale image (4 x 128 B, twice), waves 2, 3 their first W plane once more (10 per wave and K-step)
    const unsigned dA = lds_base + wv * F6_CHUNK;
    const unsigned dX1 = lds_base + A_BYTES + wv * F6_CHUNK;
    unsigned dX2 = dX1, iX2v = F6_CHUNK, offX2 = lane * 16;
    const unsigned offA = lane * 16, offA2 = lane * 16 + 4u * KP * F6_CHUNK, offX1 = lane * 16;
    if (wv == 0) { offX2 = ((lane >> 3) & 7) * KP * 128 + (lane & 7) * 16; dX2 = lds_base + A_BYTES + W_BYTES; iX2v = 128; }
    else if (wv == 1) { offX2 = ((lane >> 3) & 3) * KP * 128 + (lane & 7) * 16; dX2 = lds_base + A_BYTES + W_BYTES + AS_BYTES; iX2v = 128; }
    const unsigned in_la = lds_base + (wm * 4) * F6_CHUNK + lane * 16;
    const unsigned in_lw = lds_base + A_BYTES + (wn * 2) * F6_CHUNK + lane * 16;
    const unsigned in_lsa = lds_base + A_BYTES + W_BYTES + (wm * 4) * 128 + lr * 2;
    const unsigned in_lsw = lds_base + A_BYTES + W_BYTES + AS_BYTES + (wn * 2) * 128 + lr * 2;
    auto stream_ptrs = [&](int bm, int bn, unsigned long long &a, unsigned long long &x1, unsigned long long &x2) {
        a = (unsigned long long)(p.act + ((size_t)(bm * (BM / 32) + wv) * KP) * F6_CHUNK);
        x1 = (unsigned long long)(p.wgt + ((size_t)(bn * 4 + wv) * KP) * F6_CHUNK);
        if (wv == 0) x2 = (unsigned long long)((const uint8_t *)p.ascales + ((size_t)(bm * (BM / 32)) * KP) * 128);
        else if (wv == 1) x2 = (unsigned long long)((const uint8_t *)p.wscales + ((size_t)(bn * 4) * KP) * 128);
        else x2 = x1;
    };

    v16f acc[2][4]; // [n tile][m tile]
    unsigned ring = 0, npre = 0, landed = 0;
    unsigned long long pA = 0, pX1 = 0, pX2 = 0;
    int tile = pos, bm = 0, bn = 0;
    bool have = tile < NT;
    if (have) { tile_coords(tile, bm, bn); stream_ptrs(bm, bn, pA, pX1, pX2); }
    while (have) {
        const int m0 = bm * BM, n0 = bn * BN;
        const int ntile = tile + G;
        const bool have_next = ntile < NT;
        int nbm = 0, nbn = 0;
        unsigned ncnt = 0;
        unsigned long long nA = 0, nX1 = 0, nX2 = 0;
        if (have_next) { tile_coords(ntile, nbm, nbn); stream_ptrs(nbm, nbn, nA, nX1, nX2); ncnt = KP; }
        {
            const unsigned kp_s = KP;
            auto srd = [](unsigned long long ptr) {
                v4i r;
                r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ptr);
                r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(ptr >> 32));
                r[2] = -1;
                r[3] = 0x00020000;
                return r;
            };
            v4i rA = srd(pA), rX1 = srd(pX1), rX2 = srd(pX2);
#define CLOB_V                                                                                                          \
    "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146",  \
    "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165",  \
    "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184",  \
    "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v196", "v197", "v198", "v199", "v204", "v205"
#define CLOB_A                                                                                                          \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23",     \
    "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46",  \
    "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69",  \
    "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92",  \
    "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113",  \
    "a114", "a115", "a116", "a117", "a118", "a119"
#define LOOP3_OPERANDS                                                                                                  \
    : "={v[0:15]}"(acc[0][0]), "={v[16:31]}"(acc[0][1]), "={v[32:47]}"(acc[0][2]), "={v[48:63]}"(acc[0][3]),                                   \
      "={v[64:79]}"(acc[1][0]), "={v[80:95]}"(acc[1][1]), "={v[96:111]}"(acc[1][2]), "={v[112:127]}"(acc[1][3]),                               \
      "+{s[72:75]}"(rA), "+{s[76:79]}"(rX1), "+{s[80:83]}"(rX2)                                                                                  \
    : "{v192}"(in_la), "{v193}"(in_lw), "{v194}"(in_lsa), "{v195}"(in_lsw), "{v200}"(offA), "{v201}"(offA2), "{v202}"(offX1), "{v203}"(offX2),   \
      "{s46}"(kp_s), "{s47}"(dA), "{s48}"(dX1), "{s49}"(dX2), "{s50}"(iX2v), "{s58}"(ring), "{s59}"(npre), "{s60}"(ncnt),                        \
      "{s[62:63]}"(nA), "{s[64:65]}"(nX1), "{s[66:67]}"(nX2), "{s68}"(landed)                                                                     \
    : "memory", "scc", "m0", "s51", "s53", "s55", "s56", "s57", "s61", "s84", "s85", "s86", "s87", CLOB_V, CLOB_A
            if constexpr (DT == SVDQ_BF16) {
                asm volatile(
#include SVDQ_LOOP3_BF16
                    LOOP3_OPERANDS);
            } else {
                asm volatile(
#include SVDQ_LOOP3_FP16
                    LOOP3_OPERANDS);
            }
            ring = (ring + (kp_s % NSTAGE) * STAGE_BYTES) % (NSTAGE * STAGE_BYTES);
            npre = min(3u, ncnt); // every body of the loop leaves three K-steps of the operand stream ahead of the one it computed
            landed = npre;   // the epilogue below waits vmcnt(0) before its first store: the next tile's prefetch has landed
            pA = nA; pX1 = nX1; pX2 = nX2;
        }
        // ---- epilogue: the single rounding to 16 bits and the store (gemm_w4a4.hip "EpilogueDefault"); lane owns rows mw0 + 32 mi + lr, columns nw0 + 32 ni + 8 c + 4 h + e
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int nw0 = n0 + wn * 64, mw0 = m0 + wm * 128;
#pragma unroll
        for (int mi = 0; mi < 4; mi++) {
            const int m_abs = mw0 + mi * 32 + lr;
            T *orow = (T *)p.out + (size_t)m_abs * p.ldo + nw0 + h * 8;
            const bool ok = m_abs < p.M;
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    unsigned x[2], y[2];
#pragma unroll
                    for (int d = 0; d < 2; d++) {
                        float v0 = acc[ni][mi][(2 * j) * 4 + 2 * d], v1 = acc[ni][mi][(2 * j) * 4 + 2 * d + 1];
                        float w0 = acc[ni][mi][(2 * j + 1) * 4 + 2 * d], w1 = acc[ni][mi][(2 * j + 1) * 4 + 2 * d + 1];
                        if constexpr (DT == SVDQ_FP16) {
                            v0 = fminf(fmaxf(v0, -65504.f), 65504.f); v1 = fminf(fmaxf(v1, -65504.f), 65504.f);
                            w0 = fminf(fmaxf(w0, -65504.f), 65504.f); w1 = fminf(fmaxf(w1, -65504.f), 65504.f);
                        }
                        x[d] = (unsigned)hbits(f2h<T>(v0)) | ((unsigned)hbits(f2h<T>(v1)) << 16);
                        y[d] = (unsigned)hbits(f2h<T>(w0)) | ((unsigned)hbits(f2h<T>(w1)) << 16);
                        auto sw = __builtin_amdgcn_permlane32_swap(x[d], y[d], false, false);
                        x[d] = sw[0];
                        y[d] = sw[1];
                    }
                    if (ok) {
                        v4i o = {(int)x[0], (int)x[1], (int)y[0], (int)y[1]};
                        *reinterpret_cast<v4i *>(orow + ni * 32 + j * 16) = o;
                    }
                }
        }
        have = have_next;
        tile = ntile;
        bm = nbm;
        bn = nbn;
    }
    if (p.clk && tid == 0) {
        p.clk[2 * blockIdx.x] = __builtin_readcyclecounter() - clk_t0;
        p.clk[2 * blockIdx.x + 1] = wall_clock64() - clk_r0;
    }
}

// ---- host: operands as tools/ablate/gemm_probe.hip generates them ---------------------------------------------------------------------------------------------
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }
static inline float rndf() { return (rnd() >> 8) * (1.0f / 16777216.0f); }
static inline uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline uint16_t fp16(float f) { _Float16 hh = (_Float16)f; uint16_t u; memcpy(&u, &hh, 2); return u; }
static void *dev_codes(size_t bytes, bool is_unsigned) {
    std::vector<uint8_t> hb(bytes);
    for (size_t i = 0; i + 2 < bytes; i += 3) {
        uint32_t w = 0;
        for (int j = 0; j < 4; j++) {
            float g = (rndf() + rndf() + rndf() + rndf() - 2.0f) * 5.2f;
            int q = (int)lrintf(g);
            if (is_unsigned) { q = abs(q) * 2; if (q > 15) q = 15; }
            else { if (q > 7) q = 7; if (q < -7) q = -7; }
            uint32_t c = q < 0 ? (32u | (uint32_t)(-q)) : (uint32_t)q;
            w |= c << (6 * j);
        }
        hb[i] = w & 255; hb[i + 1] = (w >> 8) & 255; hb[i + 2] = (w >> 16) & 255;
    }
    void *d; CK(hipMalloc(&d, bytes)); CK(hipMemcpy(d, hb.data(), bytes, hipMemcpyHostToDevice));
    return d;
}
static void *dev_half(size_t n, int dtype, float lo, float hi) {
    std::vector<uint16_t> hb(n);
    for (size_t i = 0; i < n; i++) { float v = lo + (hi - lo) * rndf(); hb[i] = dtype == SVDQ_BF16 ? bf16(v) : fp16(v); }
    void *d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, hb.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}

typedef int (*gemm_fn)(const svdq_gemm_args *, void *);
typedef int64_t (*wsb_fn)(void);
typedef const char *(*err_fn)(void);

int main(int argc, char **argv) {
    std::string lib = "nunchaku_amd/csrc/libsvdq_amd.so";
    int M = 4608, K = 3072, N = 3072, iters = 50, warm = 1000, dtype = SVDQ_BF16, grid = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--lib") lib = argv[++i];
        else if (a == "--shape") { M = atoi(argv[++i]); K = atoi(argv[++i]); N = atoi(argv[++i]); }
        else if (a == "--iters") iters = atoi(argv[++i]);
        else if (a == "--warm") warm = atoi(argv[++i]);
        else if (a == "--fp16") dtype = SVDQ_FP16;
        else if (a == "--grid") grid = atoi(argv[++i]);
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (K % 128 || N % 128) { fprintf(stderr, "K and N must be multiples of 128\n"); return 2; }
    const int M_pad = (M + 255) / 256 * 256, Gk = K / 64;
    const bool act_unsigned = K == 12288;
    void *act = dev_codes((size_t)M_pad * K * 3 / 4, act_unsigned), *wgt = dev_codes((size_t)N * K * 3 / 4, false);
    void *asc = dev_half((size_t)Gk * M_pad, dtype, 0.05f, 0.4f), *wsc = dev_half((size_t)Gk * N, dtype, 0.002f, 0.01f);
    void *out_ref, *out; CK(hipMalloc(&out_ref, (size_t)M_pad * N * 2)); CK(hipMalloc(&out, (size_t)M_pad * N * 2));
    CK(hipMemset(out_ref, 0, (size_t)M_pad * N * 2)); CK(hipMemset(out, 0xff, (size_t)M_pad * N * 2));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double ops = 2.0 * M_pad * (double)N * K;

    // ---- the product kernel on the same operands (R = 0, no bias, 256 x 128 tiles)
    double us_ref = 0;
    void *hl = dlopen(lib.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!hl) { fprintf(stderr, "dlopen %s: %s\n", lib.c_str(), dlerror()); return 1; }
    gemm_fn gemm = (gemm_fn)dlsym(hl, "svdq_gemm_w4a4");
    wsb_fn wsb = (wsb_fn)dlsym(hl, "svdq_gemm_workspace_bytes");
    err_fn last_error = (err_fn)dlsym(hl, "svdq_last_error");
    svdq_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.act = act; a.wgt = wgt; a.ascales = asc; a.wscales = wsc; a.out = out_ref;
    a.M = M; a.M_pad = M_pad; a.N = N; a.K = K; a.R = 0; a.ldo = N; a.dtype = dtype; a.fuse = SVDQ_FUSE_NONE; a.act_unsigned = act_unsigned; a.geometry = 1;
    a.workspace_bytes = wsb(); CK(hipMalloc(&a.workspace, a.workspace_bytes)); CK(hipMemset(a.workspace, 0, a.workspace_bytes));
    if (gemm(&a, st)) { fprintf(stderr, "product gemm: %s\n", last_error()); return 1; }
    CK(hipStreamSynchronize(st));
    for (int i = 0; i < warm; i++) gemm(&a, st);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; i++) gemm(&a, st);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); us_ref = ms * 1e3 / iters; }

    // ---- the probe kernel
    const int NT = (M_pad / BM) * (N / BN);
    int cus = 256; { int dev = 0; hipGetDevice(&dev); hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); }
    if (grid <= 0) { // whole rounds on the fewest workgroups (the product's rule for launches without a stream-K split), a multiple of 8
        const int rounds = (NT + cus - 1) / cus;
        grid = (NT + rounds - 1) / rounds;
        grid = std::min(cus, (grid + 7) / 8 * 8);
    }
    long long *clk; CK(hipMalloc((void **)&clk, 2 * 512 * sizeof(long long))); CK(hipMemset(clk, 0, 2 * 512 * sizeof(long long)));
    P128 p{(const uint8_t *)act, (const uint8_t *)wgt, asc, wsc, out, M, M_pad, N, K, N, nullptr};
    auto launch = [&](const P128 &pp) {
        if (dtype == SVDQ_BF16) hipLaunchKernelGGL(gemm128_kernel<SVDQ_BF16>, dim3(grid), dim3(256), 0, st, pp);
        else hipLaunchKernelGGL(gemm128_kernel<SVDQ_FP16>, dim3(grid), dim3(256), 0, st, pp);
    };
    launch(p);
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    std::vector<uint16_t> h0((size_t)M * N), h1((size_t)M * N);
    CK(hipMemcpy(h0.data(), out_ref, h0.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), out, h1.size() * 2, hipMemcpyDeviceToHost));
    size_t bad = 0, first = (size_t)-1;
    for (size_t i = 0; i < h0.size(); i++) if (h0[i] != h1[i]) { if (!bad) first = i; bad++; }
    for (int i = 0; i < warm; i++) launch(p);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; i++) launch(p);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    P128 pc = p; pc.clk = clk;
    launch(pc);
    CK(hipStreamSynchronize(st));
    std::vector<long long> hc(2 * 512);
    CK(hipMemcpy(hc.data(), clk, hc.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double sc = 0, stt = 0; int n = 0;
    for (int i = 0; i < 512; i++) if (hc[2 * i + 1] > 0) { sc += hc[2 * i]; stt += hc[2 * i + 1]; n++; }
    const double tiles_per_wg = (double)NT / grid, tg_per_simd_tile = 8.0 * (K / 64); // tile-groups (32 x 32 x 64) per SIMD and tile: 4 x 2 MFMA tiles x K / 64
    printf("{\"probe\":\"128x64 wave tile, one wave per SIMD\",\"M\":%d,\"K\":%d,\"N\":%d,\"dtype\":%d,\"grid\":%d,\"tiles\":%d,\"mismatches\":%zu,\"first_mismatch\":%lld,"
           "\"us\":%.2f,\"TOPS\":%.1f,\"product_us\":%.2f,\"product_TOPS\":%.1f,\"ratio\":%.4f,\"wg_cycles\":%.0f,\"eff_GHz\":%.3f,\"cycles_per_tile_group_incl_epilogue\":%.1f}\n",
           M, K, N, dtype, grid, NT, bad, (long long)(bad ? (long long)first : -1), us, ops / us * 1e-6, us_ref, ops / us_ref * 1e-6, us / us_ref,
           n ? sc / n : 0.0, n && stt > 0 ? sc / (stt * 10.0) : 0.0, n ? (sc / n) / (tiles_per_wg * tg_per_simd_tile) : 0.0);
    return bad ? 3 : 0;
}
