// What does THIS box's memory system give a tuned streaming kernel?  16-byte lanes, grid-stride loops with U independent
// accesses in flight per lane, plain and non-temporal forms; copy (read + write), read-only (sum), write-only (fill).
// tools/roofline_sweep.py prints the best of each next to the kernels of the path and to the guide's 6.29 TB/s vf4 copy.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float vf4 __attribute__((ext_vector_type(4)));     // the nontemporal builtins want a native vector type

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const vf4 *__restrict__ src, vf4 *__restrict__ dst, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += stride * U) {
        vf4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (j + u * stride < n4) v[u] = NT ? __builtin_nontemporal_load(src + j + u * stride) : src[j + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) if (j + u * stride < n4) { if (NT) __builtin_nontemporal_store(v[u], dst + j + u * stride); else dst[j + u * stride] = v[u]; }
    }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const vf4 *__restrict__ src, float *__restrict__ out, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    float acc = 0.f;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += stride * U) {
        vf4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (j + u * stride < n4) ? (NT ? __builtin_nontemporal_load(src + j + u * stride) : src[j + u * stride]) : (vf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;          // never true: keeps the loads alive without a store per lane
}
template <bool NT>
__global__ __launch_bounds__(256) void k_write(vf4 *__restrict__ dst, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const vf4 v = {1.f, 2.f, 3.f, 4.f};
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += stride) { if (NT) __builtin_nontemporal_store(v, dst + j); else dst[j] = v; }
}
extern "C" void run(int kind, int unroll, int nt, const void *src, void *dst, int64_t n4, int blocks, void *stream) {
    hipStream_t s = (hipStream_t)stream;
#define L(K, ...) hipLaunchKernelGGL((K), dim3(blocks), dim3(256), 0, s, __VA_ARGS__)
    if (kind == 0) {
        if (unroll == 1) { if (nt) L((k_copy<1, true>), (const vf4 *)src, (vf4 *)dst, n4); else L((k_copy<1, false>), (const vf4 *)src, (vf4 *)dst, n4); }
        else if (unroll == 2) { if (nt) L((k_copy<2, true>), (const vf4 *)src, (vf4 *)dst, n4); else L((k_copy<2, false>), (const vf4 *)src, (vf4 *)dst, n4); }
        else { if (nt) L((k_copy<4, true>), (const vf4 *)src, (vf4 *)dst, n4); else L((k_copy<4, false>), (const vf4 *)src, (vf4 *)dst, n4); }
    } else if (kind == 1) {
        if (unroll == 1) { if (nt) L((k_read<1, true>), (const vf4 *)src, (float *)dst, n4); else L((k_read<1, false>), (const vf4 *)src, (float *)dst, n4); }
        else if (unroll == 2) { if (nt) L((k_read<2, true>), (const vf4 *)src, (float *)dst, n4); else L((k_read<2, false>), (const vf4 *)src, (float *)dst, n4); }
        else { if (nt) L((k_read<4, true>), (const vf4 *)src, (float *)dst, n4); else L((k_read<4, false>), (const vf4 *)src, (float *)dst, n4); }
    } else {
        if (nt) L((k_write<true>), (vf4 *)dst, n4); else L((k_write<false>), (vf4 *)dst, n4);
    }
#undef L
}
