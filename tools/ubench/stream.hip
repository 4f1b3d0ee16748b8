// micro-benchmark: what does the memory system give for the access pattern of rendering_fwd
// (i64 keys + 3 f32 arrays + [N,3] rgb in, 3 f32 out) with 4-byte vs 16-byte lanes, no arithmetic?
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" {
__global__ __launch_bounds__(256) void k_dword(const int64_t* keys, const float* a, const float* b, const float* c, const float* rgb,
                                               float* o0, float* o1, float* o2, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = a[i] + b[i] + c[i] + rgb[3*i] + rgb[3*i+1] + rgb[3*i+2] + (float)keys[i];
        o0[i] = x; o1[i] = x * 2; o2[i] = x * 3;
    }
}
__global__ __launch_bounds__(256) void k_vec4(const int64_t* keys, const float* a, const float* b, const float* c, const float* rgb,
                                              float* o0, float* o1, float* o2, int64_t n) {
    const int64_t n4 = n / 4;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (int64_t)gridDim.x * 256) {
        const float4 va = ((const float4*)a)[j], vb = ((const float4*)b)[j], vc = ((const float4*)c)[j];
        const float4 r0 = ((const float4*)rgb)[3*j], r1 = ((const float4*)rgb)[3*j+1], r2 = ((const float4*)rgb)[3*j+2];
        const longlong2 k0 = ((const longlong2*)keys)[2*j], k1 = ((const longlong2*)keys)[2*j+1];
        float4 x;
        x.x = va.x + vb.x + vc.x + r0.x + r0.y + r0.z + (float)k0.x;
        x.y = va.y + vb.y + vc.y + r0.w + r1.x + r1.y + (float)k0.y;
        x.z = va.z + vb.z + vc.z + r1.z + r1.w + r2.x + (float)k1.x;
        x.w = va.w + vb.w + vc.w + r2.y + r2.z + r2.w + (float)k1.y;
        ((float4*)o0)[j] = x;
        ((float4*)o1)[j] = make_float4(x.x*2, x.y*2, x.z*2, x.w*2);
        ((float4*)o2)[j] = make_float4(x.x*3, x.y*3, x.z*3, x.w*3);
    }
}
void run(int which, const void* keys, const void* a, const void* b, const void* c, const void* rgb, void* o0, void* o1, void* o2,
         int64_t n, int blocks, void* stream) {
    if (which == 0) hipLaunchKernelGGL(k_dword, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const int64_t*)keys, (const float*)a, (const float*)b, (const float*)c, (const float*)rgb, (float*)o0, (float*)o1, (float*)o2, n);
    else hipLaunchKernelGGL(k_vec4, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const int64_t*)keys, (const float*)a, (const float*)b, (const float*)c, (const float*)rgb, (float*)o0, (float*)o1, (float*)o2, n);
}
}
