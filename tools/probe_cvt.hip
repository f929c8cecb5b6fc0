// probe: how does v_cvt_pk_u8_f32 round?  (decides whether the epilogue needs its v_rndne_f32)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float *x, unsigned *y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0u, 0u);
}
int main() {
    const int n = 1 << 16;
    float *hx = new float[n];
    unsigned *hy = new unsigned[n];
    for (int i = 0; i < n; ++i) hx[i] = -8.0f + (float)i / 128.0f;   // -8 .. 504 in steps of 1/128
    float *dx; unsigned *dy;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
    hipMemcpy(hy, dy, n * 4, hipMemcpyDeviceToHost);
    long bad_rne = 0, bad_trunc = 0, bad_half_up = 0;
    for (int i = 0; i < n; ++i) {
        float v = hx[i];
        float r = rintf(v); r = r < 0 ? 0 : (r > 255 ? 255 : r);
        float t = truncf(v); t = t < 0 ? 0 : (t > 255 ? 255 : t);
        float h = floorf(v + 0.5f); h = h < 0 ? 0 : (h > 255 ? 255 : h);
        if (hy[i] != (unsigned)r) ++bad_rne;
        if (hy[i] != (unsigned)t) ++bad_trunc;
        if (hy[i] != (unsigned)h) ++bad_half_up;
    }
    printf("v_cvt_pk_u8_f32 vs RNE+sat: %ld mismatches; vs trunc+sat: %ld; vs half-up+sat: %ld\n", bad_rne, bad_trunc, bad_half_up);
    printf("samples: 0.5->%u 1.5->%u 2.5->%u 2.49->%u 2.51->%u 254.5->%u 255.5->%u -0.6->%u\n",
           hy[(int)((0.5f + 8) * 128)], hy[(int)((1.5f + 8) * 128)], hy[(int)((2.5f + 8) * 128)], hy[(int)((2.4921875f + 8) * 128)],
           hy[(int)((2.5078125f + 8) * 128)], hy[(int)((254.5f + 8) * 128)], hy[(int)((255.5f + 8) * 128)], hy[(int)((-0.6015625f + 8) * 128)]);
    return 0;
}
