// What an event costs the stream it is recorded on: kernel A, [event], kernel B on one stream; a second stream waits for the
// event and runs kernel C. The gap A's end -> B's start (device clock, wall_clock64: 100 MHz) with
//   0: no event at all            1: hipEventRecord behind A            2: A launched with hipExtLaunchKernelGGL(stopEvent)
// hipcc --offload-arch=gfx950 -O3 scripts/micro/event_gap.hip -o scripts/micro/event_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void work(unsigned long long* stamps, int slot, int spin) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long t = t0;
    while (t - t0 < (unsigned long long)spin) t = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot] = t0;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot + 1] = wall_clock64();
}

int main() {
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev;
    CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    unsigned long long *stamps;
    CHECK(hipHostMalloc(reinterpret_cast<void**>(&stamps), 64 * sizeof(unsigned long long), hipHostMallocMapped));
    const int spin = 3000;      // 30 us
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<double> gap_ab, gap_ac;
        for (int it = 0; it < 60; ++it) {
            if (mode == 2) {
                hipExtLaunchKernelGGL(work, dim3(256), dim3(256), 0, s1, nullptr, ev, 0, stamps, 0, spin);
            } else {
                hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s1, stamps, 0, spin);
                if (mode == 1) CHECK(hipEventRecord(ev, s1));
            }
            if (mode != 0) {
                CHECK(hipStreamWaitEvent(s2, ev, 0));
                hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s2, stamps, 2, spin);
            }
            hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s1, stamps, 1, spin);
            CHECK(hipStreamSynchronize(s1));
            CHECK(hipStreamSynchronize(s2));
            if (it >= 10) {
                gap_ab.push_back((double)(stamps[2] - stamps[1]) / 100.0);
                if (mode != 0) gap_ac.push_back((double)(stamps[4] - stamps[1]) / 100.0);
            }
        }
        std::sort(gap_ab.begin(), gap_ab.end());
        std::sort(gap_ac.begin(), gap_ac.end());
        printf("mode %d (%s): A end -> B start median %.2f us (min %.2f, max %.2f)", mode,
               mode == 0 ? "no event" : (mode == 1 ? "hipEventRecord behind A" : "hipExtLaunchKernelGGL stopEvent"),
               gap_ab[gap_ab.size() / 2], gap_ab.front(), gap_ab.back());
        if (mode != 0) printf(";  A end -> C start (other stream) median %.2f us (min %.2f)", gap_ac[gap_ac.size() / 2], gap_ac.front());
        printf("\n");
    }
    // the host's wait on a stop event: A's end stamp must be there when hipEventSynchronize returns (and hipEventQuery agrees)
    int late = 0, not_ready = 0;
    for (int it = 0; it < 200; ++it) {
        stamps[1] = 0;
        hipExtLaunchKernelGGL(work, dim3(256), dim3(256), 0, s1, nullptr, ev, 0, stamps, 0, spin);
        hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, s1, stamps, 1, spin);
        if (hipEventQuery(ev) == hipSuccess) ++not_ready; else (void)hipGetLastError();      // (30 us of work: must not be ready yet)
        CHECK(hipEventSynchronize(ev));
        if (*reinterpret_cast<volatile unsigned long long*>(&stamps[1]) == 0) ++late;
        if (hipEventQuery(ev) != hipSuccess) ++late;
        CHECK(hipStreamSynchronize(s1));
    }
    printf("host wait on a stop event, 200 rounds: returned before the kernel's end %d times; ready at launch %d times\n", late, not_ready);
    return 0;
}
