// Host-execution shim for the reference's CUDA sources (TEST INFRASTRUCTURE, never shipped).
//
// Force-included (g++ -include) in front of /root/reference/nerfacc/cuda/csrc/grid.cu so that the
// reference's own kernel and host wrapper compile as plain C++ against libtorch CPU tensors and run
// one "thread" after another.  Nothing in here restates the reference's algorithm: it only supplies
// what nvcc/cuda_runtime.h would (vector types, execution-space keywords, launch geometry).
#pragma once
#include <torch/extension.h>

#include <cmath>
#include <cstdint>

// --- execution-space keywords -------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__

// --- vector types of cuda_runtime.h (only the members helper_math uses) --------------------------
#define REF_VEC2(T, N) struct N { T x, y; }; static inline N make_##N(T x, T y) { return N{x, y}; }
#define REF_VEC3(T, N) struct N { T x, y, z; }; static inline N make_##N(T x, T y, T z) { return N{x, y, z}; }
#define REF_VEC4(T, N) struct N { T x, y, z, w; }; static inline N make_##N(T x, T y, T z, T w) { return N{x, y, z, w}; }
REF_VEC2(float, float2) REF_VEC3(float, float3) REF_VEC4(float, float4)
REF_VEC2(int, int2) REF_VEC3(int, int3) REF_VEC4(int, int4)
REF_VEC2(unsigned, uint2) REF_VEC3(unsigned, uint3) REF_VEC4(unsigned, uint4)

// CUDA's math overloads for float (cuda_runtime.h / math_functions.hpp): min/max(float, float) are
// fminf/fmaxf.  Without them `min(tdist.x, ...)` (grid.cu:185) would pick helper_math's int overload.
static inline float min(float a, float b) { return std::fmin(a, b); }
static inline float max(float a, float b) { return std::fmax(a, b); }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// launch geometry seen by the "kernel" while it runs on the host
static thread_local dim3 gridDim, blockDim;
static thread_local uint3 blockIdx, threadIdx;

// kernel<<<grid, block, shmem, stream>>>(args...) is rewritten by the recipe (sed, in the pipe feeding
// the compiler; no copy of the source is written anywhere) into REF_LAUNCH(kernel, grid, block, shmem,
// stream)(args...): every (block, thread) pair runs to completion in order.  Valid for kernels without
// __syncthreads / shared memory, which is the case for everything in grid.cu.
template <class K> struct RefLauncher {
    K k;
    dim3 g, b;
    template <class... A> void operator()(A... a) const {
        gridDim = g;
        blockDim = b;
        for (unsigned bx = 0; bx < g.x; ++bx)
            for (unsigned tx = 0; tx < b.x; ++tx) {
                blockIdx = uint3{bx, 0, 0};
                threadIdx = uint3{tx, 0, 0};
                k(a...);
            }
    }
};
template <class K, class S0, class S1> static inline RefLauncher<K> ref_launch(K k, dim3 g, dim3 b, S0, S1) {
    return RefLauncher<K>{k, g, b};
}
#define REF_LAUNCH(kernel, ...) ref_launch(kernel, __VA_ARGS__)

// --- argument checks: the reference's CHECK_INPUT (include/utils_cuda.cuh:12-17) insists on CUDA tensors.  The header is
// pulled in HERE (its `#pragma once` then keeps later includes out) so that CHECK_CUDA can be re-defined for host
// tensors; contiguity is still checked by the reference's own macro.
#include "include/utils_cuda.cuh"
#undef CHECK_CUDA
#define CHECK_CUDA(x)

// --- random numbers of pdf.cu (only reached with stratified = true, which the fixtures never use): enough of
// at::PhiloxCudaState / CUDAGeneratorImpl / cuRAND for the file to compile; the generator look-up
// `at::get_generator_or_default<at::CUDAGeneratorImpl>(` is rewritten to `ref_generator(` by the recipe's sed
#include <mutex>
#include <tuple>
namespace at {
struct PhiloxCudaState { uint64_t seed = 0, offset = 0; };
struct RefGenerator {
    std::mutex mutex_;
    PhiloxCudaState philox_cuda_state(uint64_t) { return PhiloxCudaState{}; }
};
namespace cuda { namespace detail { static inline int getDefaultCUDAGenerator() { return 0; } }
namespace philox { static inline std::tuple<uint64_t, uint64_t> unpack(PhiloxCudaState s) { return {s.seed, s.offset}; } } }
}  // namespace at
template <class A, class B> static inline at::RefGenerator *ref_generator(A, B) { static at::RefGenerator g; return &g; }
struct curandStatePhilox4_32_10_t {};
static inline void curand_init(uint64_t, uint64_t, uint64_t, curandStatePhilox4_32_10_t *) {}
static inline float curand_uniform(curandStatePhilox4_32_10_t *) { return 0.5f; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }   // CUDA's integer overloads used by pdf.cu
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }

// --- the two pieces of at::cuda the host wrappers touch ---------------------------------------------
namespace at { namespace cuda {
struct CUDAStream {};
static inline CUDAStream getCurrentCUDAStream() { return CUDAStream{}; }
struct OptionalCUDAGuard {
    template <class T> explicit OptionalCUDAGuard(T&&) {}
};
}}  // namespace at::cuda

// --- float -> int conversion rule (build `gpu` only, -DREF_GPU_F2I) ----------------------------------------------------
// The reference converts with C casts `int(a.x)` (utils_math.cuh:177-179, reached from utils_grid.cuh:72-82, 104).  On a GPU
// that cast saturates and maps NaN to 0 (cvt.rzi.s32.f32 / v_cvt_i32_f32); x86's cvttss2si returns INT_MIN for both.  The
// two differ only for rays lying IN a bounding plane of a level (0 * inf = NaN in the slab test).  A function-like macro
// named `int` re-routes exactly the `int(expr)` casts of the sources parsed after this point; declarations (`int x`) are
// not function-like uses and stay as they are.  Every header the sources need is already included above.
#ifdef REF_GPU_F2I
#include <climits>
template <class T> static inline int ref_f2i(T v) { return static_cast<int>(v); }
template <> inline int ref_f2i<float>(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT_MAX;
    if (v <= -2147483648.0f) return INT_MIN;
    return static_cast<int>(v);
}
#define int(x) ref_f2i(x)
#endif
