// gemm_probe: time svdq_gemm_w4a4 straight through the C ABI, no Python (a fresh GPU box pays 1-2 minutes for its first
// `import torch`; this binary starts in milliseconds).  Loads any build of the library with dlopen, fills the operand
// images with synthetic codes / scales, and prints one JSON line per (shape, variant):
//   us per launch (hipEvent over `iters` back-to-back launches), TOP/s = (2 M N K + 2 M N R) / t,
//   a checksum of the output (compare geometries bit for bit), and -- with the tools-built probe library -- the effective shader clock of a launch
//   (per-workgroup s_memtime cycles / s_memrealtime 100 MHz ticks, svdq_ablate_set_clk).
// build: tools/ablate/build.py          run: tools/ablate/gemm_probe --lib <so> --shape 4096 12288 3072 --geoms 1,2,3
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "svdq_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }
static inline float rndf() { return (rnd() >> 8) * (1.0f / 16777216.0f); }
static inline uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline uint16_t fp16(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }

// FP6 image filled with signed 4-bit codes of a rounded Gaussian (|q| <= 7): the code distribution of a quantised
// activation / weight (layout does not matter for timing; the toggling statistics do, the chip is power-limited)
static void *dev_codes(size_t bytes, bool zero, bool is_unsigned) {
    std::vector<uint8_t> h(bytes);
    if (!zero) {
        for (size_t i = 0; i + 2 < bytes; i += 3) {
            uint32_t w = 0;
            for (int j = 0; j < 4; j++) {
                float g = (rndf() + rndf() + rndf() + rndf() - 2.0f) * 5.2f; // ~N(0, 3)
                int q = (int)lrintf(g);
                if (is_unsigned) { q = abs(q) * 2; if (q > 15) q = 15; }
                else { if (q > 7) q = 7; if (q < -7) q = -7; }
                uint32_t c = q < 0 ? (32u | (uint32_t)(-q)) : (uint32_t)q;
                w |= c << (6 * j);
            }
            h[i] = w & 255; h[i + 1] = (w >> 8) & 255; h[i + 2] = (w >> 16) & 255;
        }
    }
    void *d; CK(hipMalloc(&d, bytes)); CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
    return d;
}
static void *dev_half(size_t n, int dtype, float lo, float hi, bool zero) {
    std::vector<uint16_t> h(n);
    for (size_t i = 0; i < n; i++) { float v = zero ? 0.f : lo + (hi - lo) * rndf(); h[i] = dtype == SVDQ_BF16 ? bf16(v) : fp16(v); }
    void *d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}
static float *dev_f32(size_t n, float lo, float hi, bool zero) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; i++) h[i] = zero ? 0.f : lo + (hi - lo) * rndf();
    float *d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

typedef int (*gemm_fn)(const svdq_gemm_args *, void *);
typedef int64_t (*wsb_fn)(void);
typedef int64_t (*wsbf_fn)(const svdq_gemm_args *);
typedef void (*clk_fn)(long long *);
typedef const char *(*err_fn)(void);

int main(int argc, char **argv) {
    std::string lib = "nunchaku_amd/csrc/libsvdq_amd.so";
    int R2opt = 32;
    int M = 4096, K = 12288, N = 3072, R = 32, fuse = 0, iters = 20, warm = 1500, dtype = SVDQ_BF16, reserved = 0, use_ws = 1, split = 0;
    bool zero = false, do_trace = false, q32 = false, q32_runs = false, no_bias = false;
    double sustain = 0;
    std::vector<int> variants = {1}; // svdq_gemm_args.geometry values
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--lib") lib = argv[++i];
        else if (a == "--shape") { M = atoi(argv[++i]); K = atoi(argv[++i]); N = atoi(argv[++i]); }
        else if (a == "--R") R = atoi(argv[++i]);
        else if (a == "--R2") R2opt = atoi(argv[++i]);
        else if (a == "--fuse") fuse = atoi(argv[++i]);
        else if (a == "--iters") iters = atoi(argv[++i]);
        else if (a == "--warm") warm = atoi(argv[++i]);
        else if (a == "--fp16") dtype = SVDQ_FP16;
        else if (a == "--reserved") reserved = atoi(argv[++i]);
        else if (a == "--no-ws") use_ws = 0;
        else if (a == "--zero") zero = true;
        else if (a == "--trace") do_trace = true;
        else if (a == "--split") split = atoi(argv[++i]);
        else if (a == "--sustain") sustain = atof(argv[++i]);
        else if (a == "--q32") q32 = true;
        else if (a == "--q32runs") { q32 = true; q32_runs = true; }
        else if (a == "--no-bias") no_bias = true;
        else if (a == "--geoms") { variants.clear(); char *s = argv[++i]; for (char *t = strtok(s, ","); t; t = strtok(nullptr, ",")) variants.push_back(atoi(t)); }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    void *h = dlopen(lib.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", lib.c_str(), dlerror()); return 1; }
    gemm_fn gemm = (gemm_fn)dlsym(h, "svdq_gemm_w4a4");
    wsb_fn wsb = (wsb_fn)dlsym(h, "svdq_gemm_workspace_bytes");
    clk_fn set_clk = (clk_fn)dlsym(h, "svdq_ablate_set_clk");
    clk_fn set_trace = (clk_fn)dlsym(h, "svdq_ablate_set_trace");
    err_fn last_error = (err_fn)dlsym(h, "svdq_last_error");
    if (!gemm || !wsb || !last_error) { fprintf(stderr, "missing symbols in %s\n", lib.c_str()); return 1; }

    const int M_pad = (M + 255) / 256 * 256, G = K / 64;
    svdq_gemm_args a;
    memset(&a, 0, sizeof(a));
    const bool act_unsigned = fuse == SVDQ_FUSE_NONE && K == 12288; // the fc2 shape reads GELU_QUANT codes
    a.act = dev_codes((size_t)M_pad * K * 3 / 4, zero, act_unsigned);
    a.wgt = dev_codes((size_t)N * K * 3 / 4, zero, false);
    a.ascales = dev_half((size_t)G * M_pad, dtype, 0.05f, 0.4f, zero);
    a.wscales = dev_half((size_t)G * N, dtype, 0.002f, 0.01f, zero);
    a.bias = no_bias ? nullptr : dev_half(N, dtype, -0.1f, 0.1f, zero);
    if (R > 0) { a.lora_act_in = dev_f32((size_t)M_pad * R * (q32 ? 2 : 1), -1.f, 1.f, zero || q32); a.lora_act_format = q32_runs ? SVDQ_LORA_ACT_Q32_RUNS : q32 ? SVDQ_LORA_ACT_Q32 : SVDQ_LORA_ACT_F32; a.lora_up = dev_half((size_t)N * R, dtype, -0.05f, 0.05f, zero); }
    a.M = M; a.M_pad = M_pad; a.N = N; a.K = K; a.R = R; a.ldo = N; a.dtype = dtype; a.fuse = fuse; a.reserved = reserved;
    a.act_unsigned = act_unsigned;
    void *out; CK(hipMalloc(&out, (size_t)M_pad * N * 2)); a.out = out;
    if (fuse == SVDQ_FUSE_GELU_QUANT) {
        a.out = nullptr;
        CK(hipMalloc(&a.qout, (size_t)M_pad * N * 3 / 4));
        CK(hipMalloc(&a.oscales, (size_t)(N / 64) * M_pad * 2));
        a.next_smooth = dev_half(N, dtype, 0.5f, 2.0f, false);
        a.R2 = R2opt;
        a.next_lora_down = dev_half((size_t)N * a.R2, dtype, -0.05f, 0.05f, zero);
        CK(hipMalloc((void **)&a.lora_act_out, (size_t)M_pad * a.R2 * 8)); CK(hipMemset(a.lora_act_out, 0, (size_t)M_pad * a.R2 * 8));
    } else if (fuse == SVDQ_FUSE_RMSNORM_ROPE) {
        a.norm_q = dev_half(128, dtype, 0.5f, 1.5f, false); a.norm_k = dev_half(128, dtype, 0.5f, 1.5f, false);
        a.rotary_emb = dev_f32((size_t)M_pad * 128, -1.f, 1.f, false);
        CK(hipMalloc(&a.out_vt, (size_t)(N / 3) * M_pad * 2)); a.ldvt = M_pad;
    }
    if (split > 0) { // grouped launch: second weight set of the same shape
        a.wgt2 = dev_codes((size_t)N * K * 3 / 4, zero, false); a.wscales2 = dev_half((size_t)G * N, dtype, 0.002f, 0.01f, zero);
        a.bias2 = dev_half(N, dtype, -0.1f, 0.1f, zero); if (R > 0) a.lora_up2 = dev_half((size_t)N * R, dtype, -0.05f, 0.05f, zero);
        a.next_smooth2 = a.next_smooth; a.next_lora_down2 = a.next_lora_down; a.norm_q2 = a.norm_q; a.norm_k2 = a.norm_k;
        a.split_rows = split;
    }
    if (use_ws) {
        // ABI 20: room for the launch's 16-bit output image when its next-layer low-rank down projection can run split (geometry 0 from rank 96, geometry 7)
        wsbf_fn wsbf = (wsbf_fn)dlsym(h, "svdq_gemm_workspace_bytes_for");
        a.workspace_bytes = wsb();
        if (wsbf) { a.geometry = 7; a.workspace_bytes = std::max<int64_t>(a.workspace_bytes, wsbf(&a)); a.geometry = 0; }
        CK(hipMalloc(&a.workspace, a.workspace_bytes)); CK(hipMemset(a.workspace, 0, a.workspace_bytes));
    }
    long long *clk = nullptr;
    if (set_clk) { CK(hipMalloc((void **)&clk, 4 * 512 * sizeof(long long))); }

    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double ops = 2.0 * M_pad * (double)N * K + 2.0 * M_pad * (double)N * R;
    for (int v : variants) {
        a.variant = 0;
        a.geometry = v;
        if (set_clk) set_clk(nullptr);
        for (int i = 0; i < 3; i++)
            if (gemm(&a, st)) { fprintf(stderr, "variant %d: %s\n", v, last_error()); goto next; }
        CK(hipStreamSynchronize(st));
        {
            // a second or so of back-to-back launches first: DVFS settles on the sustained clock
            for (int i = 0; i < warm; i++) gemm(&a, st);
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; i++) gemm(&a, st);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            double eff_ghz = 0, cyc = 0;
            if (set_clk) {
                CK(hipMemset(clk, 0, 4 * 512 * sizeof(long long)));
                set_clk(clk);
                gemm(&a, st);
                CK(hipStreamSynchronize(st));
                set_clk(nullptr);
                std::vector<long long> hc(4 * 512);
                CK(hipMemcpy(hc.data(), clk, hc.size() * sizeof(long long), hipMemcpyDeviceToHost));
                double sc = 0, st_ = 0; int n = 0;
                long long t_min = 0x7fffffffffffffffLL, t_max = 0, life_max = 0, life_min = 0x7fffffffffffffffLL;
                std::vector<unsigned> cu_key;
                for (int i = 0; i < 512; i++) if (hc[4 * i + 1] > 0) {
                    sc += hc[4 * i]; st_ += hc[4 * i + 1]; n++;
                    t_min = std::min(t_min, hc[4 * i + 2]); t_max = std::max(t_max, hc[4 * i + 2] + hc[4 * i + 1]);
                    life_max = std::max(life_max, hc[4 * i + 1]); life_min = std::min(life_min, hc[4 * i + 1]);
                    const unsigned hw = (unsigned)hc[4 * i + 3], xcc = (unsigned)(hc[4 * i + 3] >> 32);
                    cu_key.push_back((xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)); // xcc, se, sh, cu
                }
                if (n) { cyc = sc / n; eff_ghz = sc / (st_ * 10.0); } // ticks are 10 ns
                // placement: workgroups per physical CU, and for CUs with two tenants how much of their lifetimes overlap
                std::vector<unsigned> uniq = cu_key; std::sort(uniq.begin(), uniq.end()); uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
                int cus2 = 0, cus1 = 0, cus3 = 0; double ov = 0, life2 = 0, life1 = 0; int n2 = 0, n1 = 0;
                for (unsigned k : uniq) {
                    std::vector<int> w; int idx = 0;
                    for (int i = 0; i < 512; i++) if (hc[4 * i + 1] > 0) { if (cu_key[idx] == k) w.push_back(i); idx++; }
                    if (w.size() == 1) { cus1++; life1 += hc[4 * w[0] + 1]; n1++; }
                    else if (w.size() == 2) {
                        cus2++;
                        const long long s0 = hc[4 * w[0] + 2], e0_ = s0 + hc[4 * w[0] + 1], s1 = hc[4 * w[1] + 2], e1_ = s1 + hc[4 * w[1] + 1];
                        const long long o = std::min(e0_, e1_) - std::max(s0, s1);
                        ov += (double)std::max(0LL, o) / std::max(1LL, std::min(e0_ - s0, e1_ - s1));
                        life2 += hc[4 * w[0] + 1] + hc[4 * w[1] + 1]; n2 += 2;
                    } else cus3++;
                }
                printf("{\"placement\":{\"wgs\":%d,\"cus_used\":%zu,\"cus_with_1\":%d,\"cus_with_2\":%d,\"cus_with_3plus\":%d,\"pair_overlap\":%.3f,"
                       "\"life_us_alone\":%.2f,\"life_us_paired\":%.2f,\"life_us_min\":%.2f,\"life_us_max\":%.2f,\"span_us\":%.2f}}\n",
                       n, uniq.size(), cus1, cus2, cus3, cus2 ? ov / cus2 : 0.0, n1 ? life1 / n1 * 0.01 : 0.0, n2 ? life2 / n2 * 0.01 : 0.0,
                       life_min * 0.01, life_max * 0.01, (t_max - t_min) * 0.01);
                if (getenv("SVDQ_PROBE_LIFETIMES")) { // every workgroup's lifetime and END time (us since the first start), by blockIdx
                    printf("{\"lifetimes_us\":[");
                    for (int i = 0, k = 0; i < 512; i++) if (hc[4 * i + 1] > 0) printf("%s[%d,%.1f,%.1f]", k++ ? "," : "", i, hc[4 * i + 1] * 0.01, (hc[4 * i + 2] + hc[4 * i + 1] - t_min) * 0.01);
                    printf("]}\n");
                }
            }
            if (set_trace && do_trace) { // per-segment phase stamps of workgroup 0
                long long *tr; CK(hipMalloc((void **)&tr, 192 * sizeof(long long))); CK(hipMemset(tr, 0, 192 * sizeof(long long)));
                set_trace(tr);
                gemm(&a, st);
                CK(hipStreamSynchronize(st));
                set_trace(nullptr);
                long long ht[192];
                CK(hipMemcpy(ht, tr, sizeof(ht), hipMemcpyDeviceToHost));
                printf("{\"trace_variant\":%d,\"segments\":[", v);
                for (int i = 0; i < 32 && ht[6 * i + 5] > 0; i++)
                    printf("%s[%lld,%lld,%lld,%lld,%lld,%lld]", i ? "," : "", ht[6 * i], ht[6 * i + 1], ht[6 * i + 2], ht[6 * i + 3], ht[6 * i + 4], ht[6 * i + 5]);
                printf("]}\n");
                CK(hipFree(tr));
            }
            unsigned long long sum = 1469598103934665603ull; // FNV-1a over the 16-bit output (or the code image)
            {
                const void *src = fuse == SVDQ_FUSE_GELU_QUANT ? a.qout : a.out;
                const size_t bytes = fuse == SVDQ_FUSE_GELU_QUANT ? (size_t)M_pad * N * 3 / 4 : (size_t)M * N * 2;
                std::vector<uint8_t> hb(bytes);
                CK(hipMemcpy(hb.data(), src, bytes, hipMemcpyDeviceToHost));
                const uint64_t *w = (const uint64_t *)hb.data();
                for (size_t i = 0; i < bytes / 8; i++) { sum ^= w[i]; sum *= 1099511628211ull; }
            }
            printf("{\"sum\":\"%016llx\",", sum);
            printf("\"M\":%d,\"K\":%d,\"N\":%d,\"R\":%d,\"fuse\":%d,\"geometry\":%d,\"reserved\":%d,\"zero\":%d,\"ws\":%d,\"split\":%d,\"us\":%.2f,\"TOPS\":%.1f,"
                   "\"wg_cycles\":%.0f,\"eff_GHz\":%.3f}\n", M, K, N, R, fuse, v, reserved, (int)zero, use_ws, split, us, ops / us * 1e-6, cyc, eff_ghz);
            fflush(stdout);
            if (sustain > 0) { // keep the GPU on this kernel for `sustain` seconds (rocm-smi power / clock sampling)
                const int n = (int)(sustain * 1e6 / us);
                for (int i = 0; i < n; i++) gemm(&a, st);
                CK(hipStreamSynchronize(st));
            }
        }
    next:;
    }
    return 0;
}
