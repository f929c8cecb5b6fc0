// cabi_bench.cpp -- a plain C++ consumer of the C ABI (include/mulut.h): no Python, no torch.
// Loads the six shipped .npy tables (or seeded random ones), builds D-natural-like frames on the host, runs the
// 2-stage sdy x4 cascade and prints per-stage device milliseconds.  Start-up takes a second, which makes it the
// driver of choice under rocprofv3 (tools/prof_quick.sh).
//   build: hipcc -O2 -o build/tools/cabi_bench tools/cabi_bench.cpp -Iinclude -Lmulut_amd/lib -lmulut_hip -Wl,-rpath,'$ORIGIN/../../mulut_amd/lib'
//   run  : build/tools/cabi_bench [--frames 8] [--h 1080] [--w 1920] [--iters 5] [--tuning key=val,...] [--dist natural|noise] [--luts DIR]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mulut.h"

#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, mulut_strerror(rc_)); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static float gauss() { float s = 0; for (int i = 0; i < 6; ++i) s += (float)(rnd() & 0xFFFF) / 65536.0f; return (s - 3.0f) * 1.41421356f; }

// int8 payload of an NPY v1 file (dtype |i1, C order); empty on any mismatch
static std::vector<int8_t> load_npy(const std::string &path, size_t want) {
    std::vector<int8_t> out;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return out;
    unsigned char hdr[10];
    if (fread(hdr, 1, 10, f) == 10 && !memcmp(hdr, "\x93NUMPY", 6)) {
        const size_t hl = hdr[8] | (hdr[9] << 8);
        fseek(f, (long)(10 + hl), SEEK_SET);
        out.resize(want);
        if (fread(out.data(), 1, want, f) != want) out.clear();
    }
    fclose(f);
    return out;
}

int main(int argc, char **argv) {
    int frames = 8, H = 1080, W = 1920, iters = 5;
    std::string tuning, dist = "natural", lutdir;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--frames")) frames = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--h")) H = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--w")) W = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--iters")) iters = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--tuning")) tuning = argv[i + 1];
        else if (!strcmp(argv[i], "--dist")) dist = argv[i + 1];
        else if (!strcmp(argv[i], "--luts")) lutdir = argv[i + 1];
    }
    mulut_ctx *ctx = nullptr;
    CK(mulut_create(0, &ctx));
    CK(mulut_configure(ctx, 2, "sdy", 4, 4));
    const size_t rows = 83521;
    for (int s = 1; s <= 2; ++s)
        for (const char *m = "sdy"; *m; ++m) {
            const int vn = s == 2 ? 16 : 1;
            std::vector<int8_t> t;
            if (!lutdir.empty()) {
                char name[256];
                snprintf(name, sizeof name, "%s/LUT_ft_x4_4bit_int8_s%d_%c.npy", lutdir.c_str(), s, *m);
                t = load_npy(name, rows * vn);
            }
            if (t.empty()) {
                t.resize(rows * vn);
                for (auto &v : t) v = (int8_t)((int)(rnd() & 0xFF) - 128);
            }
            CK(mulut_set_lut(ctx, s, *m, t.data(), (int64_t)rows, vn));
        }
    size_t start = 0;
    while (start < tuning.size()) {
        size_t end = tuning.find(',', start);
        if (end == std::string::npos) end = tuning.size();
        const std::string kv = tuning.substr(start, end - start);
        const size_t eq = kv.find('=');
        if (eq != std::string::npos) CK(mulut_set_tuning(ctx, kv.substr(0, eq).c_str(), atoi(kv.c_str() + eq + 1)));
        start = end + 1;
    }
    // one frame on the host, replicated with a shift: smooth field (6 sinusoids per channel) + sigma-2 noise, or noise
    std::vector<uint8_t> img((size_t)frames * H * W * 3);
    {
        std::vector<uint8_t> one((size_t)H * W * 3);
        for (int c = 0; c < 3; ++c) {
            float fy[6], fx[6], am[6], ph[6];
            for (int k = 0; k < 6; ++k) {
                fy[k] = (0.5f + 5.5f * (rnd() & 0xFFFF) / 65536.0f) * 6.2831853f / (float)(H > W ? H : W);
                fx[k] = (0.5f + 5.5f * (rnd() & 0xFFFF) / 65536.0f) * 6.2831853f / (float)(H > W ? H : W);
                am[k] = 0.3f + 0.7f * (rnd() & 0xFFFF) / 65536.0f;
                ph[k] = 6.2831853f * (rnd() & 0xFFFF) / 65536.0f;
            }
            float amax = 0;
            for (int k = 0; k < 6; ++k) amax += am[k];
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float v;
                    if (dist == "noise") v = (float)(rnd() & 0xFF);
                    else {
                        float a = 0;
                        for (int k = 0; k < 6; ++k) a += am[k] * sinf(fy[k] * y + fx[k] * x + ph[k]);
                        v = (a / amax * 0.5f + 0.5f) * 255.0f + 2.0f * gauss();
                    }
                    one[((size_t)y * W + x) * 3 + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : lrintf(v));
                }
        }
        for (int n = 0; n < frames; ++n)
            for (size_t i = 0; i < one.size(); ++i) img[(size_t)n * one.size() + i] = one[(i + (size_t)n * 3 * 977) % one.size()];
    }
    uint8_t *din = nullptr, *dout = nullptr;
    HK(hipMalloc((void **)&din, img.size()));
    HK(hipMalloc((void **)&dout, img.size() * 16));
    HK(hipMemcpy(din, img.data(), img.size(), hipMemcpyHostToDevice));
    CK(mulut_reserve(ctx, frames, H, W, 3));
    CK(mulut_set_stage_timing(ctx, 1));
    printf("kernels: %s | %s\n", mulut_kernel_name(ctx, 0), mulut_kernel_name(ctx, 1));
    for (int it = 0; it < iters; ++it) {
        CK(mulut_pipeline(ctx, din, dout, frames, H, W, 3, MULUT_LAYOUT_HWC, nullptr));
        float ms[8];
        const int n = mulut_last_stage_ms(ctx, ms, 8);
        printf("iter %d:", it);
        for (int s = 0; s < n; ++s) printf("  stage %d %.4f ms (%.1f us/frame)", s + 1, ms[s], ms[s] * 1e3f / frames);
        printf("\n");
    }
    HK(hipDeviceSynchronize());
    // checksum so that a run can be compared across builds
    std::vector<uint8_t> out(img.size() * 16);
    HK(hipMemcpy(out.data(), dout, out.size(), hipMemcpyDeviceToHost));
    unsigned long long sum = 1469598103934665603ull;
    for (size_t i = 0; i < out.size(); i += 97) sum = (sum ^ out[i]) * 1099511628211ull;
    printf("checksum %016llx\n", sum);
    if (getenv("MULUT_CLKPROBE")) printf("clock probe: %llu ticks in block 0 of the last final-stage launch\n", *(unsigned long long *)out.data());
    hipFree(din); hipFree(dout);
    mulut_destroy(ctx);
    return 0;
}
