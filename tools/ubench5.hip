// probe: v_cvt_scalef32_2xpk16_fp6_f32 element order, scale semantics and rounding (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v6i __attribute__((ext_vector_type(6)));
__global__ void k(const float *in, unsigned *out, float scale) {
    v16f a, b;
    for (int i = 0; i < 16; i++) { a[i] = in[threadIdx.x * 32 + i]; b[i] = in[threadIdx.x * 32 + 16 + i]; }
    v6i r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    for (int i = 0; i < 6; i++) out[threadIdx.x * 6 + i] = (unsigned)r[i];
}
static float dec(unsigned c) { // e2m3
    int s = c >> 5, e = (c >> 3) & 3, m = c & 7;
    float v = e == 0 ? m / 8.0f : (1 + m / 8.0f) * (float)(1 << (e - 1));
    return s ? -v : v;
}
int main() {
    float h[64 * 32]; unsigned o[64 * 6];
    for (int l = 0; l < 64; l++) for (int j = 0; j < 32; j++) {
        float v;
        if (l == 0) v = (float)(j - 16) * 0.5f;          // -8 .. 7.5 in halves (ties)
        else if (l == 1) v = (float)j * 0.5f;            // 0 .. 15.5 (unsigned range, ties)
        else if (l == 2) v = (float)(j - 16) * 0.5f + 0.01f;
        else if (l == 3) v = (float)j + 0.49f;
        else v = sinf(l * 32 + j) * 7.0f;
        h[l * 32 + j] = v;
    }
    float *di; unsigned *dout;
    hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    for (float scale : {8.0f, 1.0f, 0.125f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, scale);
        hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
        printf("scale %g\n", scale);
        for (int l = 0; l < 5; l++) {
            printf(" lane %d:", l);
            for (int j = 0; j < 32; j++) {
                int bit = 6 * j; unsigned long long w = o[l * 6 + (bit >> 5)] | ((unsigned long long)((bit >> 5) + 1 < 6 ? o[l * 6 + (bit >> 5) + 1] : 0) << 32);
                unsigned c = (unsigned)((w >> (bit & 31)) & 63);
                printf(" %g>%g", h[l * 32 + j], dec(c) * 8);
            }
            printf("\n");
        }
    }
    return 0;
}
