// What does the shader-clock counter (s_memtime, what tools/phase_cycles.py stamps with) count per microsecond while a SHORT, lightly
// occupied kernel runs?  hipcc --offload-arch=gfx950 -O3 tools/ubench/clock_probe.hip -o /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long *out, int n) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    float x = threadIdx.x * 1e-3f;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x = fmaf(x, 1.0001f, 1e-7f);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = (unsigned long long)(x == 123.0f); }
}
int main() {
    unsigned long long *d, h[3];
    hipMalloc(&d, 64);
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, device clock rate attribute %d kHz\n", rate, clk);
    const int shapes[][2] = {{1, 64}, {206, 512}, {1024, 256}, {4096, 256}};
    for (auto &sh : shapes)
        for (int n : {200, 2000, 20000}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe, dim3(sh[0]), dim3(sh[1]), 0, 0, d, n);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(probe, dim3(sh[0]), dim3(sh[1]), 0, 0, d, n);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            const double us_wall = (double)h[1] / (rate ? rate : 100000) * 1e3;
            printf("%5d x %4d threads, %6d x 16 dependent fma: event %8.2f us; counter %9llu ticks over %8.2f us of wall clock = %7.1f ticks/us; %5.2f ticks per fma\n",
                   sh[0], sh[1], n, ms * 1e3, h[0], us_wall, h[0] / us_wall, (double)h[0] / (16.0 * n));
        }
    return 0;
}
