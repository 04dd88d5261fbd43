// gsr_harness — a C++ caller written the way the reference's GSGaussians is
// (apps/gsrast/GSGaussians.cpp): grow-only chunk functionals (resizeFunctional, :27-42), one-time
// SoA upload (:109-137), per-frame matrix prep (:155-176), then gscuda::forward(...) with the
// reference's exact argument list through include/gscuda_shim.hpp, then the device sync + error
// poll of CHECK_CUDA_ERROR (CudaBuffer.hpp:8-12). It stands in for the app, which needs
// GLFW/GL/imgui/lmdb and cannot exist on a headless MI355X box.
//
//   gsr_harness <scene.bin> <out.bin> [frames]
// scene.bin: int32 N, W, H; float32 view[16], proj[16], camPos[3], tanFOVx, tanFOVy, bg[3];
//            then means3D[4N], scales[4N], rotations[4N], opacities[N], shs[48N].
// out.bin  : float32 outColor[3*W*H] (planar), then uint32 numRendered-independent checksum-free raw.
//   gsr_harness --bench K <camera.bin> <N> <seed> [warmup]
// the timed loop of a C++ caller (what bench.py's Python step costs beside it): camera.bin = int32 W, H + the 40 camera
// floats above; the scene is the garden-like stand-in of gsrast_amd/scenes.py (splitmix64 stream `seed`, N splats) generated
// here on the host; `warmup` frames, then K frames, each with the reference caller's device sync; prints bench_ms = mean.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../include/gscuda_shim.hpp"

#define CHECK_HIP_ERROR(expr)                                                          \
    do {                                                                               \
        expr;                                                                          \
        hipError_t e_ = hipDeviceSynchronize();                                        \
        if (e_ == hipSuccess) e_ = hipPeekAtLastError();                               \
        if (e_ != hipSuccess) fprintf(stderr, "HIP error: %s\n", hipGetErrorString(e_)); \
    } while (0)

static std::function<char*(size_t)> resizeFunctional(void** ptr, size_t& S) {
    return [ptr, &S](size_t N) {
        if (N > S) {
            if (*ptr) (void)hipFree(*ptr);
            (void)hipMalloc(ptr, 2 * N);
            S = 2 * N;
        }
        return reinterpret_cast<char*>(*ptr);
    };
}

template <typename T>
static T* upload(const std::vector<T>& v) {
    T* d = nullptr;
    (void)hipMalloc(reinterpret_cast<void**>(&d), v.size() * sizeof(T));
    (void)hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

template <typename T>
static bool read_vec(FILE* f, std::vector<T>& v, size_t n) {
    v.resize(n);
    return fread(v.data(), sizeof(T), n, f) == n;
}

// gsrast_amd/scenes.py restated: uniform u_k = (splitmix64(seed, k) >> 40) * 2^-24, normals by Box-Muller on two blocks of
// uniforms, in the stream order of garden_like_scene (positions, scales, quaternions, opacities, DC).
struct Stream {
    uint64_t seed, pos = 0;
    double uniform() {
        uint64_t z = seed + (++pos) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        return (double)(z >> 40) * (1.0 / 16777216.0);
    }
    void normal(std::vector<double>& out, size_t n) {
        std::vector<double> u1(n);
        for (auto& v : u1) v = uniform();
        out.resize(n);
        for (size_t i = 0; i < n; ++i) out[i] = std::sqrt(-2.0 * std::log(1.0 - u1[i])) * std::cos(2.0 * M_PI * uniform());
    }
};
static void garden_like(int N, uint64_t seed, std::vector<float>& means, std::vector<float>& scales, std::vector<float>& rots,
                        std::vector<float>& opac, std::vector<float>& shs) {
    Stream s{seed};
    std::vector<double> g;
    const size_t n = (size_t)N;
    means.assign(4 * n, 1.0f); scales.assign(4 * n, (float)M_E); rots.resize(4 * n); opac.resize(n); shs.assign(48 * n, 0.0f);
    const double spread[3] = {4.0, 1.5, 4.0};
    s.normal(g, 3 * n);
    for (size_t i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) means[4 * i + c] = (float)(g[3 * i + c] * spread[c]);
    s.normal(g, 3 * n);
    for (size_t i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) scales[4 * i + c] = (float)std::exp(-4.5 + 1.2 * g[3 * i + c]);
    s.normal(g, 4 * n);
    for (size_t i = 0; i < n; ++i) {
        const double* q = &g[4 * i];
        const double len = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int c = 0; c < 4; ++c) rots[4 * i + c] = (float)(q[c] / len);
    }
    s.normal(g, n);
    for (size_t i = 0; i < n; ++i) opac[i] = (float)(1.0 / (1.0 + std::exp(-3.0 * g[i])));
    s.normal(g, 3 * n);
    for (size_t i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) shs[48 * i + c] = (float)g[3 * i + c];
}

int main(int argc, char** argv) {
    const bool bench = argc >= 6 && std::string(argv[1]) == "--bench";
    if (!bench && argc < 3) { fprintf(stderr, "usage: %s scene.bin out.bin [frames] | --bench K camera.bin N seed [warmup]\n", argv[0]); return 2; }
    int frames = bench ? atoi(argv[2]) : (argc > 3 ? atoi(argv[3]) : 1);
    const int warmup = bench ? (argc > 6 ? atoi(argv[6]) : 5) : 0;
    FILE* f = fopen(bench ? argv[3] : argv[1], "rb");
    if (!f) { perror("scene"); return 2; }
    int32_t hdr[3];
    int N, W, H;
    std::vector<float> cam, means, scales, rots, opac, shs;
    if (bench) {
        if (fread(hdr, 4, 2, f) != 2 || !read_vec(f, cam, 16 + 16 + 3 + 2 + 3)) { fprintf(stderr, "short camera file\n"); return 2; }
        W = hdr[0]; H = hdr[1]; N = atoi(argv[4]);
        garden_like(N, (uint64_t)atoll(argv[5]), means, scales, rots, opac, shs);
        frames += warmup;
    } else {
        if (fread(hdr, 4, 3, f) != 3) return 2;
        N = hdr[0]; W = hdr[1]; H = hdr[2];
        if (!read_vec(f, cam, 16 + 16 + 3 + 2 + 3) || !read_vec(f, means, 4 * (size_t)N) || !read_vec(f, scales, 4 * (size_t)N) ||
            !read_vec(f, rots, 4 * (size_t)N) || !read_vec(f, opac, (size_t)N) || !read_vec(f, shs, 48 * (size_t)N)) {
            fprintf(stderr, "short scene file\n");
            return 2;
        }
    }
    fclose(f);

    float *dMeans = upload(means), *dScales = upload(scales), *dRots = upload(rots), *dOpac = upload(opac), *dShs = upload(shs);
    std::vector<float> view(cam.begin(), cam.begin() + 16), proj(cam.begin() + 16, cam.begin() + 32);
    std::vector<float> camPos(cam.begin() + 32, cam.begin() + 35), bg(cam.begin() + 37, cam.begin() + 40);
    const float tanFOVx = cam[35], tanFOVy = cam[36];
    float *dView = upload(view), *dProj = upload(proj), *dCamPos = upload(camPos), *dBg = upload(bg);
    int* dRects = nullptr;
    (void)hipMalloc(reinterpret_cast<void**>(&dRects), sizeof(int) * 2 * (size_t)N);
    float* dOut = nullptr;
    (void)hipMalloc(reinterpret_cast<void**>(&dOut), sizeof(float) * 3 * (size_t)W * H);
    (void)hipMemset(dOut, 0, sizeof(float) * 3 * (size_t)W * H);

    void *geomPtr = nullptr, *binningPtr = nullptr, *imgPtr = nullptr;
    size_t allocatedGeom = 0, allocatedBinning = 0, allocatedImg = 0;
    auto geomFunc = resizeFunctional(&geomPtr, allocatedGeom);
    auto binningFunc = resizeFunctional(&binningPtr, allocatedBinning);
    auto imgFunc = resizeFunctional(&imgPtr, allocatedImg);

    double best_ms = 1e30;
    auto bench_t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < frames; ++i) {
        if (bench && i == warmup) bench_t0 = std::chrono::steady_clock::now();
        const auto t0 = std::chrono::steady_clock::now();
        CHECK_HIP_ERROR(gscuda::forward(geomFunc, binningFunc, imgFunc, N, 3, 16, dBg, W, H, dMeans, dShs, nullptr, dOpac,
                                        dScales, 1.0f, dRots, nullptr, dView, dProj, dCamPos, tanFOVx, tanFOVy, false, dOut,
                                        nullptr, dRects, nullptr, nullptr));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms < best_ms) best_ms = ms;
        if (gscuda::lastError() != GSR_OK) {
            fprintf(stderr, "gscuda::forward failed: %s (%s)\n", gsr_error_string(gscuda::lastError()), gsr_last_hip_error());
            return 1;
        }
        // (after the sync inside CHECK_HIP_ERROR: the device-side half of the sticky-error poll)
        if (gscuda::pollAsyncError() != GSR_OK) {
            fprintf(stderr, "gscuda::forward: a device-side wait gave up, frame %d is invalid\n", i);
            return 1;
        }
    }
    const double bench_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - bench_t0).count() / std::max(1, frames - warmup);
    // What the Inspector does: re-derive GeometryState pointers from the caller-owned chunk.
    char* chunk = reinterpret_cast<char*>(geomPtr);
    gscuda::gs::GeometryState st = gscuda::gs::GeometryState::fromChunk(chunk, N);
    uint32_t lastOffset = 0;
    (void)hipMemcpy(&lastOffset, st.pointOffsets + (N - 1), 4, hipMemcpyDeviceToHost);

    if (bench) {
        printf("gsr_harness: N=%d %dx%d numRendered=%u warmup=%d frames=%d bench_ms=%.4f best_ms=%.4f\n", N, W, H, lastOffset, warmup,
               frames - warmup, bench_ms, best_ms);
        return 0;
    }
    std::vector<float> out(3 * (size_t)W * H);
    (void)hipMemcpy(out.data(), dOut, out.size() * sizeof(float), hipMemcpyDeviceToHost);
    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror("out"); return 2; }
    fwrite(out.data(), sizeof(float), out.size(), o);
    fclose(o);
    printf("gsr_harness: N=%d %dx%d numRendered=%u frames=%d best_ms=%.3f required(Geometry)=%zu\n", N, W, H, lastOffset, frames,
           best_ms, gscuda::required<gscuda::gs::GeometryState>(N));
    return 0;
}
