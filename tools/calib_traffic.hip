// calib_traffic.hip -- known-byte-count kernels to calibrate the rocprofv3 memory-side counters
// (WRITE_SIZE / TCC_EA0_WRREQ*, FETCH_SIZE / TCC_EA0_RDREQ*) for the access widths our kernels use
// (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a known byte count").
//   hipcc --offload-arch=gfx950 -O3 -o build/calib_traffic tools/calib_traffic.hip
//   rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d out -- build/calib_traffic
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u3 __attribute__((ext_vector_type(3)));

__global__ void store_b32(uint32_t *p, size_t n) {            // 4 B per lane, contiguous
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}
__global__ void store_b96(uint32_t *p, size_t n) {            // 12 B per lane, contiguous (our RGB rows)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { u3 v = {(uint32_t)i, 1u, 2u}; *(u3 *)(p + 3 * i) = v; }
}
__global__ void store_b128(uint4 *p, size_t n) {              // 16 B per lane, contiguous
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void store_b8(uint8_t *p, size_t n) {              // 1 B per lane, contiguous (K1 output)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint8_t)i;
}
__global__ void load_b8_rows(const uint8_t *p, uint32_t *out, int W, int rows) {   // 68-B row segments, as a tile load
    // block b reads `rows` segments of 68 B starting at column 64*b-2 (clamped) of consecutive rows
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < rows * 68; i += blockDim.x) {
        int r = i / 68, c = i % 68;
        int x = 64 * (blockIdx.x % (W / 64)) - 2 + c;
        x = x < 0 ? 0 : (x >= W ? W - 1 : x);
        int y = (blockIdx.x / (W / 64)) * (rows - 4) + r;
        acc += p[(size_t)y * W + x];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    const size_t bytes = 768ull << 20;   // > 256 MiB Infinity Cache
    void *buf; hipMalloc(&buf, bytes + 4096);
    uint32_t *out; hipMalloc((void **)&out, 64 << 20);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    size_t n;
    n = bytes / 4;  hipLaunchKernelGGL(store_b32, dim3((n + 255) / 256), dim3(256), 0, 0, (uint32_t *)buf, n);
    n = bytes / 12; hipLaunchKernelGGL(store_b96, dim3((n + 255) / 256), dim3(256), 0, 0, (uint32_t *)buf, n);
    n = bytes / 16; hipLaunchKernelGGL(store_b128, dim3((n + 255) / 256), dim3(256), 0, 0, (uint4 *)buf, n);
    n = bytes;      hipLaunchKernelGGL(store_b8, dim3((n + 255) / 256), dim3(256), 0, 0, (uint8_t *)buf, n);
    // tile-style byte loads: image W=1920, 20-row x 68-byte segments per block, 16 useful rows per block
    const int W = 1920, H = 1080 * 24;   // 24 planes of 1080 rows = 49.8 MB (one K2 launch of 8 frames x 3 channels)
    const int blocks = (W / 64) * (H / 16);
    hipLaunchKernelGGL(load_b8_rows, dim3(blocks), dim3(256), 0, 0, (const uint8_t *)buf, out, W, 20);
    hipDeviceSynchronize();
    printf("bytes written per store kernel: %zu ; tile-load useful bytes %zu (fetched segments %zu B)\n", bytes,
           (size_t)W * H, (size_t)blocks * 20 * 68);
    return 0;
}
