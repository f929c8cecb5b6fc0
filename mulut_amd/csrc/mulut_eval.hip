// mulut_eval.hip -- device-side evaluation of a super-resolved frame against its ground truth:
// Y-channel PSNR and SSIM exactly as the reference's test script computes them
//   y = _rgb2ycbcr(img)[:, :, 0]            common/utils.py:42-60   (BT.601 studio swing, float64)
//   PSNR(y_gt, y_out, shave=scale)          common/utils.py:63-72   (float32 difference, border shaved)
//   cal_ssim(y_gt, y_out)                   common/utils.py:75-101  (11x11 Gaussian sigma 1.5, 'valid', float64)
// used at sr/4_test_lut.py:313-315.  Not on the inference hot path: it saves the D2H copy of two HR frames
// per image when a whole benchmark set is scored.  Two kernels, each leaving one float64 partial sum per
// workgroup; the host adds the partials in index order (deterministic).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/mulut.h"

namespace {

__device__ __forceinline__ double luma(const uint8_t *p) {
    // first row of T (common/utils.py:46) and O[0] = 16
    return (double)p[0] * 0.256788235294118 + (double)p[1] * 0.504129411764706 + (double)p[2] * 0.097905882352941 + 16.0;
}

constexpr int kSseThreads = 256;

// sum over the shaved interior of (float32(y_out) - float32(y_gt))^2, squared in float32 as the reference does
__global__ void __launch_bounds__(kSseThreads) sse_y_kernel(const uint8_t *gt, const uint8_t *out, int H, int W, int shave,
                                                           double *partial) {
    const int h = H - 2 * shave, w = W - 2 * shave;
    const long long n = (long long)h * w;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * kSseThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kSseThreads) {
        const int y = (int)(i / w) + shave, x = (int)(i % w) + shave;
        const size_t o = ((size_t)y * W + x) * 3;
        const float d = (float)luma(out + o) - (float)luma(gt + o);
        acc += (double)(d * d);
    }
    __shared__ double s[kSseThreads];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int k = kSseThreads / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

constexpr int kWin = 11, kT = 32;          // window, output tile edge
constexpr int kP = kT + kWin - 1;          // 42: input patch edge

struct Gauss {
    double k[kWin];
};

// one workgroup = one 32x32 tile of the 'valid' SSIM map; both luma patches staged in LDS as float64
__global__ void __launch_bounds__(256) ssim_y_kernel(const uint8_t *gt, const uint8_t *out, int H, int W, Gauss g, double *partial) {
    __shared__ double sa[kP * kP], sb[kP * kP];
    __shared__ double red[256];
    const int oh = H - (kWin - 1), ow = W - (kWin - 1);
    const int tiles_x = (ow + kT - 1) / kT;
    const int ty0 = (int)(blockIdx.x / tiles_x) * kT, tx0 = (int)(blockIdx.x % tiles_x) * kT;
    for (int i = threadIdx.x; i < kP * kP; i += 256) {
        int y = ty0 + i / kP, x = tx0 + i % kP;
        y = y < H ? y : H - 1;             // outside the image only for tile positions that are masked below
        x = x < W ? x : W - 1;
        const size_t o = ((size_t)y * W + x) * 3;
        sa[i] = luma(gt + o);
        sb[i] = luma(out + o);
    }
    __syncthreads();
    const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
    const int lx = threadIdx.x % kT, ly0 = (int)(threadIdx.x / kT) * 4;
    double sum = 0.0;
    for (int q = 0; q < 4; ++q) {
        const int ly = ly0 + q;
        if (ty0 + ly >= oh || tx0 + lx >= ow) continue;
        double m1 = 0, m2 = 0, s11 = 0, s22 = 0, s12 = 0;
        for (int i = 0; i < kWin; ++i) {
            const double *ra = sa + (ly + i) * kP + lx, *rb = sb + (ly + i) * kP + lx;
            for (int j = 0; j < kWin; ++j) {
                const double wgt = g.k[i] * g.k[j], a = ra[j], b = rb[j];
                m1 += wgt * a;
                m2 += wgt * b;
                s11 += wgt * (a * a);
                s22 += wgt * (b * b);
                s12 += wgt * (a * b);
            }
        }
        const double v1 = s11 - m1 * m1, v2 = s22 - m2 * m2, cov = s12 - m1 * m2;
        sum += ((2 * m1 * m2 + C1) * (2 * cov + C2)) / ((m1 * m1 + m2 * m2 + C1) * (v1 + v2 + C2));
    }
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

constexpr int kSseBlocks = 1024;

long long ssim_tiles(int H, int W) {
    const long long oh = H - (kWin - 1), ow = W - (kWin - 1);
    if (oh <= 0 || ow <= 0) return 0;
    return ((oh + kT - 1) / kT) * ((ow + kT - 1) / kT);
}

}  // namespace

extern "C" {

long long mulut_eval_ws_doubles(int H, int W) { return (long long)kSseBlocks + ssim_tiles(H, W); }

int mulut_eval_y(int device, const void *gt_hwc, const void *out_hwc, int H, int W, int shave, double *ws, long long ws_doubles,
                 double *psnr, double *ssim, void *stream) {
    if (!gt_hwc || !out_hwc || !ws || !psnr || !ssim || H <= 0 || W <= 0 || shave < 0) return MULUT_EINVAL;
    if (H - 2 * shave <= 0 || W - 2 * shave <= 0 || H < kWin || W < kWin) return MULUT_ESHAPE;
    const long long tiles = ssim_tiles(H, W);
    if (ws_doubles < kSseBlocks + tiles) return MULUT_EWORKSPACE;
    if (hipSetDevice(device) != hipSuccess) return MULUT_ENODEVICE;
    hipStream_t st = (hipStream_t)stream;
    Gauss g;   // cv2.getGaussianKernel(11, 1.5): exp(-(i-5)^2 / (2 sigma^2)), normalised (common/utils.py:78)
    double tot = 0;
    for (int i = 0; i < kWin; ++i) { g.k[i] = std::exp(-((i - 5.0) * (i - 5.0)) / (2.0 * 1.5 * 1.5)); tot += g.k[i]; }
    for (int i = 0; i < kWin; ++i) g.k[i] /= tot;
    hipLaunchKernelGGL(sse_y_kernel, dim3(kSseBlocks), dim3(kSseThreads), 0, st, (const uint8_t *)gt_hwc, (const uint8_t *)out_hwc, H, W,
                       shave, ws);
    hipLaunchKernelGGL(ssim_y_kernel, dim3((unsigned)tiles), dim3(256), 0, st, (const uint8_t *)gt_hwc, (const uint8_t *)out_hwc, H, W, g,
                       ws + kSseBlocks);
    if (hipGetLastError() != hipSuccess) return MULUT_EHIP;
    std::vector<double> h((size_t)(kSseBlocks + tiles));
    if (hipMemcpyAsync(h.data(), ws, h.size() * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return MULUT_EHIP;
    if (hipStreamSynchronize(st) != hipSuccess) return MULUT_EHIP;
    double sse = 0, s = 0;
    for (int i = 0; i < kSseBlocks; ++i) sse += h[i];
    for (long long i = 0; i < tiles; ++i) s += h[kSseBlocks + i];
    const double n = (double)(H - 2 * shave) * (double)(W - 2 * shave);
    const double rmse = std::sqrt((double)(float)(sse / n));   // np.mean of a float32 array is a float32
    *psnr = 20.0 * std::log10(255.0 / rmse);
    *ssim = s / ((double)(H - (kWin - 1)) * (double)(W - (kWin - 1)));
    return MULUT_OK;
}

}  // extern "C"
