// How long does a launch of the count pass's SHAPE take when the kernel does nothing?  (round 5: the count pass takes ~26 us whatever
// the ray count (3.5 k - 8 k), the lanes per ray (16 / 32) or the workgroup width — is part of that the dispatch of ~200 workgroups
// that each own a CU's LDS?)      hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(int *out) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0 && blockIdx.x == 0 && out) smem[0] = 1;
}
// every thread does `n` dependent FMAs: a fixed per-wave critical path of ~n * 4-8 cycles
__global__ void chain_kernel(float *out, int n) {
    extern __shared__ char smem[];
    float x = threadIdx.x * 1e-3f;
    for (int i = 0; i < n; ++i) x = fmaf(x, 1.0001f, 1e-7f);
    if (x == 123.456f) out[0] = x;
}
static float time_launch(void (*launch)(hipStream_t), int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch(0);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        launch(0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}
static int g_blocks, g_threads, g_lds, g_n;
static float *g_out;
static void l_empty(hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(g_blocks), dim3(g_threads), g_lds, s, (int *)nullptr); }
static void l_chain(hipStream_t s) { hipLaunchKernelGGL(chain_kernel, dim3(g_blocks), dim3(g_threads), g_lds, s, g_out, g_n); }
int main() {
    hipMalloc(&g_out, 64);
    hipFuncSetAttribute((const void *)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int shapes[][3] = {{1, 64, 0}, {206, 512, 144 * 1024}, {206, 512, 0}, {103, 1024, 144 * 1024}, {206, 1024, 144 * 1024}, {412, 256, 56 * 1024},
                             {1641, 64, 0}, {256, 512, 144 * 1024}, {26, 512, 144 * 1024}};
    for (auto &sh : shapes) {
        g_blocks = sh[0]; g_threads = sh[1]; g_lds = sh[2];
        const float t_e = time_launch(l_empty, 50);
        g_n = 2000;
        const float t_c = time_launch(l_chain, 50);
        g_n = 8000;
        const float t_c4 = time_launch(l_chain, 50);
        printf("%5d workgroups x %4d threads, %3d KB LDS: empty %6.2f us   2000 dependent fma %6.2f us   8000: %6.2f us\n", sh[0], sh[1], sh[2] / 1024, t_e, t_c, t_c4);
    }
    return 0;
}
