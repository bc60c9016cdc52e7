// How many kernels of different streams run side by side: 8 one-workgroup spin kernels on 8 streams; the wall time is one kernel's
// when every stream has a hardware queue of its own, two kernels' with the runtime's default of 4 queues.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hwq_check tools/calib/hwq_check.hip && /tmp/hwq_check [setenv]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__global__ void spin(long long cycles, unsigned* out) {
    const long long t0 = wall_clock64();
    unsigned x = 0;
    while (wall_clock64() - t0 < cycles) x++;
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "setenv")) setenv("GPU_MAX_HW_QUEUES", "8", 0);
    const int N = 8;
    hipStream_t s[N]; unsigned* d;
    hipMalloc(&d, 4096);
    for (int i = 0; i < N; i++) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    for (int rep = 0; rep < 2; rep++) {
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], 5000000ll, d + i);   // 50 ms at 100 MHz
        hipDeviceSynchronize();
        printf("%s: 8 spin kernels on 8 streams: %.1f ms\n", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "default",
               std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
    }
    return 0;
}
