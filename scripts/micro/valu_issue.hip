// What a vector instruction costs a gfx950 SIMD, by instruction class and by the number of waves resident on the SIMD.
// Every wave runs the same unrolled stream of one class (eight independent register chains, or ONE dependent chain).
// W waves per SIMD = ONE workgroup of 256 x W lanes per CU (a workgroup lives on one CU, its waves dealt round the four
// SIMDs; 100 KB of dynamic LDS keep a second workgroup off the CU), 256 workgroups; W = 8: two workgroups of 1024 lanes
// per CU (70 KB each), 512 workgroups. Reported: SIMD cycles per wave-instruction
//   = elapsed shader cycles of a wave (s_memtime, median over the waves) / (instructions per wave x W)
// — the figure an issue roofline needs — and, to check that the waves really ran together, the same from the kernel's
// wall time (in brackets; it includes the launch and the tail).
// hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_issue.hip -o scripts/micro/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kUnroll = 64, kIters = 400;

enum Class { FMA32, PKFMA32, FMA64, MUL64, ADD64, EXP32, CVT64, CVT32, CMP_SAND, CNDMASK, READLANE, LDSREAD, MIX_BLEND, CND_RAW, FMA_NOP, CMP_ONLY, CMP_SAND_RAW, ROUND_TRIP, CND_SGPR, CMP_CND, V_MOV, V_AND, NUM_CLASSES };
const char* kNames[NUM_CLASSES] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_exp_f32", "v_cvt_f64_f32",
                                   "v_cvt_f32_f64", "v_cmp_lt_f32 + s_and_b64", "v_cndmask_b32", "v_readlane_b32", "ds_read_b32",
                                   "blend mix (11 f64, 3 pk, 12 f32, 6 cmp+s_and)", "v_cndmask_b32, one asm block (no s_nop)",
                                   "v_fma_f32 + s_nop 0", "v_cmp_lt_f32 -> vcc only", "v_cmp -> vcc, s_and_b64 (one block, no s_nop)",
                                   "v_cmp -> s pair -> s_andn2 -> v_cndmask on it",
                                   "v_cndmask_b32 on a scalar pair (VOP3)", "v_cmp -> vcc, v_cndmask on it (a pair)", "v_mov_b32", "v_and_b32"};

template <int C, bool DEP>
__global__ __launch_bounds__(1024) void issue_kernel(unsigned long long* out, float seed) {
    extern __shared__ float lds[];
    lds[threadIdx.x & 255] = seed;
    __syncthreads();
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {seed, seed}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0;
    unsigned long long m = 0, m2 = 0;
    int addr = (threadIdx.x & 63) * 4;
    const float k = 0.999f; const double kd = 0.999;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll / 8; ++u) {
#define EIGHT(OP) if (DEP) { OP(0, 0) OP(0, 0) OP(0, 0) OP(0, 0) OP(0, 0) OP(0, 0) OP(0, 0) OP(0, 0) } else { OP(0, 0) OP(1, 1) OP(2, 2) OP(3, 3) OP(4, 4) OP(5, 5) OP(6, 6) OP(7, 7) }
            if (C == FMA32) {
#define OPF(i, j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a##i) : "v"(k));
                EIGHT(OPF)
            } else if (C == PKFMA32) {
#define OPP(i, j) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p##i) : "v"(p7));
                if (DEP) { OPP(0, 0) OPP(0, 0) OPP(0, 0) OPP(0, 0) OPP(0, 0) OPP(0, 0) OPP(0, 0) OPP(0, 0) }
                else { OPP(0, 0) OPP(1, 1) OPP(2, 2) OPP(3, 3) OPP(4, 4) OPP(5, 5) OPP(6, 6) OPP(0, 0) }
            } else if (C == FMA64) {
#define OPD(i, j) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d##i) : "v"(kd));
                EIGHT(OPD)
            } else if (C == MUL64) {
#define OPM(i, j) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##i) : "v"(kd));
                EIGHT(OPM)
            } else if (C == ADD64) {
#define OPA(i, j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##i) : "v"(kd));
                EIGHT(OPA)
            } else if (C == EXP32) {
#define OPE(i, j) asm volatile("v_exp_f32 %0, %0" : "+v"(a##i));
                EIGHT(OPE)
            } else if (C == CVT64) {
#define OPC(i, j) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d##i) : "v"(a##j));
                EIGHT(OPC)
            } else if (C == CVT32) {
#define OPG(i, j) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a##i) : "v"(d##j));
                EIGHT(OPG)
            } else if (C == CMP_SAND) {
                // the blend's decision idiom: a compare into a scalar pair, merged into a running lane mask
#define OPS(i, j) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n s_and_b64 %0, %0, vcc" : "+s"(m) : "v"(a##i), "v"(k) : "vcc", "scc");
                EIGHT(OPS)
            } else if (C == CNDMASK) {
#define OPN(i, j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(k) : "vcc");
                EIGHT(OPN)
            } else if (C == READLANE) {
                int s;
#define OPR(i, j) asm volatile("v_readlane_b32 %0, %1, 3\n v_mov_b32 %1, %0" : "=s"(s), "+v"(a##i));
                EIGHT(OPR)
            } else if (C == LDSREAD) {
#define OPL(i, j) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(a##i) : "v"(addr));
                if (DEP) { asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_and_b32 %1, 0xfc, %0" : "=v"(a0), "+v"(addr)); }
                else { asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8\n ds_read_b32 %2, %8\n ds_read_b32 %3, %8\n ds_read_b32 %4, %8\n ds_read_b32 %5, %8\n ds_read_b32 %6, %8\n ds_read_b32 %7, %8\n s_waitcnt lgkmcnt(0)"
                                    : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr)); }
            } else if (C == CND_RAW) {
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
            } else if (C == FMA_NOP) {
                asm volatile("v_fma_f32 %0, %0, %8, %0\n s_nop 0\n v_fma_f32 %1, %1, %8, %1\n s_nop 0\n v_fma_f32 %2, %2, %8, %2\n s_nop 0\n v_fma_f32 %3, %3, %8, %3\n s_nop 0\n"
                             "v_fma_f32 %4, %4, %8, %4\n s_nop 0\n v_fma_f32 %5, %5, %8, %5\n s_nop 0\n v_fma_f32 %6, %6, %8, %6\n s_nop 0\n v_fma_f32 %7, %7, %8, %7\n s_nop 0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
            } else if (C == CMP_ONLY) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                             "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");
            } else if (C == CMP_SAND_RAW) {
                asm volatile("v_cmp_lt_f32 vcc, %1, %9\n s_and_b64 %0, %0, vcc\n v_cmp_lt_f32 vcc, %2, %9\n s_and_b64 %0, %0, vcc\n v_cmp_lt_f32 vcc, %3, %9\n s_and_b64 %0, %0, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %9\n s_and_b64 %0, %0, vcc\n v_cmp_lt_f32 vcc, %5, %9\n s_and_b64 %0, %0, vcc\n v_cmp_lt_f32 vcc, %6, %9\n s_and_b64 %0, %0, vcc\n"
                             "v_cmp_lt_f32 vcc, %7, %9\n s_and_b64 %0, %0, vcc\n v_cmp_lt_f32 vcc, %8, %9\n s_and_b64 %0, %0, vcc"
                             : "+s"(m), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc", "scc");
            } else if (C == ROUND_TRIP) {
                // a decision as the blend takes it: compare -> lane mask in a scalar pair -> merged with another mask -> back as a select
#define RT(i) "v_cmp_lt_f32 vcc, %" #i ", %9\n s_andn2_b64 vcc, vcc, %0\n s_nop 0\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
                asm volatile(RT(1) RT(2) RT(3) RT(4) RT(5) RT(6) RT(7) RT(8)
                             : "+s"(m), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc", "scc");
            } else if (C == CND_SGPR) {
                asm volatile("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                             "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "s"(m2));
            } else if (C == CMP_CND) {
#define CC(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
                asm volatile(CC(0) CC(1) CC(2) CC(3) CC(4) CC(5) CC(6) CC(7)
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");
            } else if (C == V_MOV) {
                asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (C == V_AND) {
                asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
            } else if (C == MIX_BLEND) {
                // one strip evaluation of the blend, as instruction classes: 11 double, 3 packed, 12 single, 6 compare + scalar merge
                asm volatile(
                    "v_cvt_f64_f32 %0, %8\n v_mul_f64 %0, %0, %12\n v_add_f64 %1, %0, %12\n v_add_f64 %2, %1, %12\n v_add_f64 %0, %0, %2\n"
                    "v_fma_f64 %1, %0, %12, %0\n v_mul_f64 %2, %0, %0\n v_fma_f64 %0, %0, %12, %1\n v_fma_f64 %0, %1, %2, %0\n v_mul_f64 %0, %0, %3\n v_cvt_f32_f64 %8, %0\n"
                    "v_pk_fma_f32 %4, %4, %5, %4\n v_pk_fma_f32 %5, %4, %5, %5\n v_pk_mul_f32 %4, %4, %5\n"
                    "v_mul_f32 %9, %8, %11\n v_min_f32 %9, %9, %11\n v_sub_f32 %10, %11, %9\n v_mul_f32 %10, %10, %9\n v_mul_f32 %9, %9, %10\n"
                    "v_fma_f32 %9, %9, %11, %10\n v_fma_f32 %10, %9, %11, %10\n v_fma_f32 %9, %9, %11, %10\n v_and_b32 %9, 0x3f7fffff, %9\n v_lshl_add_u32 %10, %9, 3, %10\n"
                    "v_and_b32 %10, 0x3f7fffff, %10\n v_max_f32 %8, %8, %11\n"
                    "v_cmp_lt_f32 vcc, %8, %11\n s_and_b64 %6, %6, vcc\n v_cmp_lt_f32 vcc, %9, %11\n s_and_b64 %6, %6, vcc\n v_cmp_gt_f32 vcc, %10, %11\n s_andn2_b64 %6, %6, vcc\n"
                    "v_cmp_lt_f32 vcc, %8, %11\n s_and_b64 %7, %7, vcc\n v_cmp_lt_f32 vcc, %9, %11\n s_and_b64 %7, %7, vcc\n v_cmp_gt_f32 vcc, %10, %11\n s_or_b64 %7, %7, vcc\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(p0), "+v"(p1), "+s"(m), "+s"(m2), "+v"(a0), "+v"(a1), "+v"(a2)
                    : "v"(k), "v"(kd) : "vcc", "scc");
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + (float)m + (float)m2;
    if (acc == 12345.678f) out[1023] = 1;
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 64) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int C, bool DEP>
int run(unsigned long long* dev, int waves_per_simd, double* cyc_per_instr, double* wall_cyc_per_instr) {
    const int per_wg = waves_per_simd == 8 ? 4 : waves_per_simd, wgs = waves_per_simd == 8 ? 512 : 256;
    const int lds = waves_per_simd == 8 ? 70 * 1024 : 100 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(issue_kernel<C, DEP>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipMemset(dev, 0, 1024 * 8));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((issue_kernel<C, DEP>), dim3(wgs), dim3(256 * per_wg), lds, 0, dev, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
    }
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(1024);
    CHECK(hipMemcpy(h.data(), dev, 1024 * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> got;
    for (int b = 0; b < 64; ++b) for (int w = 0; w < 4 * per_wg; ++w) got.push_back(h[b * 16 + w]);
    std::sort(got.begin(), got.end());
    const double per_wave = (C == MIX_BLEND ? 38.0 : 1.0) * kUnroll * (C == MIX_BLEND ? 0.125 : 1.0) * (double)kIters;
    *cyc_per_instr = (double)got[got.size() / 2] / (per_wave * waves_per_simd);
    *wall_cyc_per_instr = ms * 1e-3 * 2.19e9 / (per_wave * waves_per_simd);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return 0;
}

template <int C>
int row(unsigned long long* dev) {
    const int ws[5] = {1, 2, 3, 4, 8};
    printf("%-48s", kNames[C]);
    for (int w : ws) { double c = 0, wc = 0; if (run<C, false>(dev, w, &c, &wc)) return 1; printf("  %5.2f (%5.2f)", c, wc); }
    if (C != MIX_BLEND) { double c = 0, wc = 0; if (run<C, true>(dev, 1, &c, &wc)) return 1; printf("   | dependent, 1 wave: %5.2f", c); }
    printf("\n");
    return 0;
}

int main() {
    unsigned long long* dev;
    CHECK(hipMalloc(reinterpret_cast<void**>(&dev), 1024 * 8));
    printf("SIMD cycles (s_memtime units: see the clock line) per wave-instruction; columns: waves resident per SIMD = 1, 2, 3, 4, 8 (independent chains); in brackets: from the wall time at 2.19 GHz\n");
    printf("(pairs — v_cmp + s_and, v_fma + s_nop, v_readlane + v_mov — the round-trip triple and ds_read + wait count as ONE; the mix row: per instruction of its 38)\n");
    if (row<FMA32>(dev) || row<PKFMA32>(dev) || row<FMA64>(dev) || row<MUL64>(dev) || row<ADD64>(dev) || row<EXP32>(dev) || row<CVT64>(dev) ||
        row<CVT32>(dev) || row<CMP_SAND>(dev) || row<CNDMASK>(dev) || row<READLANE>(dev) || row<LDSREAD>(dev) || row<MIX_BLEND>(dev) ||
        row<CND_RAW>(dev) || row<FMA_NOP>(dev) || row<CMP_ONLY>(dev) || row<CMP_SAND_RAW>(dev) || row<ROUND_TRIP>(dev) ||
        row<CND_SGPR>(dev) || row<CMP_CND>(dev) || row<V_MOV>(dev) || row<V_AND>(dev)) return 1;
    // what a tick of the cycle counter is: a known wall time against it
    {
        double c = 0, wc = 0;
        if (run<FMA32, false>(dev, 1, &c, &wc)) return 1;
        printf("clock: v_fma_f32, 1 wave per SIMD: %.2f ticks per instruction in the wave, %.2f from the wall time at an assumed 2.19 GHz -> the tick is %.3f GHz\n",
               c, wc, 2.19 * c / wc);
    }
    return 0;
}
