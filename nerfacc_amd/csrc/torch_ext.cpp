// torch_ext.cpp — the PyTorch-ROCm C++ extension in front of libnerfacc_hip.so (module `nerfacc_amd._hip`).
//
// Same role as the reference's pybind module `nerfacc.csrc` (nerfacc/cuda/csrc/nerfacc.cpp:126-163): the 21
// names of that module with the same argument order, plus the fused entry points of this implementation.
// The reference's functions take torch::Tensor, allocate their outputs from the caching allocator, launch on
// the current stream under a device guard and block the host only where a size is needed (data_spec.hpp:91);
// so do these.  All kernels live behind the C ABI of include/nerfacc_hip.h — this file is host plumbing only:
// argument checks (CHECK_INPUT, utils_cuda.cuh:12-17), allocation, the readback of the few integers that size
// an output, the cache of the bit-packed occupancy grid.  No CPU path: a host tensor is an error.
#include <torch/extension.h>

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <limits>
#include <map>
#include <mutex>
#include <optional>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/nerfacc_hip.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

// ---------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------
inline void check_rc(int rc) {
    TORCH_CHECK(rc == NFA_OK, "nerfacc_amd: ", nfa_last_error());
}

// the library's option table (nfa_set_option): 0 = launch the emit pass after the read-back of the totals
inline bool speculative_emit_allowed() {
    int64_t v = 1;
    int32_t is_set = 0;
    nfa_get_option("speculative_emit", &v, &is_set);
    return !is_set || v != 0;
}

inline void check_input(const Tensor &t, const char *name, std::optional<at::ScalarType> dtype = std::nullopt) {
    TORCH_CHECK(t.defined(), name, " is undefined");
    TORCH_CHECK(t.is_cuda(), name, " must be a CUDA/HIP tensor (nerfacc_amd has no CPU kernels)");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
    if (dtype) TORCH_CHECK(t.scalar_type() == *dtype, name, " must have dtype ", *dtype, ", got ", t.scalar_type());
}

// 1-D (or [n, inner]) tensor of the given dtype with exactly n leading elements, on the reference tensor's device
inline void check_len(const Tensor &t, const char *name, at::ScalarType dtype, int64_t n, const Tensor &like, int64_t inner = 1) {
    check_input(t, name, dtype);
    TORCH_CHECK(t.numel() == n * inner, name, " must have ", n * inner, " elements, got ", t.numel());
    TORCH_CHECK(t.device() == like.device(), name, " is on ", t.device(), ", expected ", like.device());
}
inline void check_len(const OptTensor &t, const char *name, at::ScalarType dtype, int64_t n, const Tensor &like, int64_t inner = 1) {
    if (t) check_len(*t, name, dtype, n, like, inner);
}

template <class T> inline T *ptr(const Tensor &t) { return t.defined() ? reinterpret_cast<T *>(t.data_ptr()) : nullptr; }
template <class T> inline T *ptr(const OptTensor &t) { return (t && t->defined()) ? reinterpret_cast<T *>(t->data_ptr()) : nullptr; }

inline hipStream_t stream_of(const Tensor &t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

using Guard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;

inline at::TensorOptions opts(const Tensor &like, at::ScalarType dt) { return like.options().dtype(dt); }

[[noreturn]] void raise_not_implemented(const char *msg) {
    PyErr_SetString(PyExc_NotImplementedError, msg);
    throw py::error_already_set();
}

// ---------------------------------------------------------------------------------------------------
// optional HIP-event timing of named single-kernel calls (bench.py's live roofline measurement).  Events are
// recorded on the stream the kernel is launched on; nothing is synchronised until timing_summary().
// ---------------------------------------------------------------------------------------------------
struct Timing {
    std::mutex mu;
    bool on = false;
    std::vector<std::string> names;
    std::map<std::string, std::vector<std::pair<hipEvent_t, hipEvent_t>>> records;
} g_timing;

struct Timed {
    const char *name;
    hipStream_t s;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    Timed(const char *n, hipStream_t st) : name(n), s(st) {
        if (!g_timing.on) return;
        std::lock_guard<std::mutex> l(g_timing.mu);
        bool want = false;
        for (auto &x : g_timing.names) want |= (x == n);
        if (!want) return;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, s);
    }
    ~Timed() {
        if (!e0) return;
        hipEventRecord(e1, s);
        std::lock_guard<std::mutex> l(g_timing.mu);
        g_timing.records[name].emplace_back(e0, e1);
    }
};

// names = None stops recording and KEEPS what was recorded (timing_summary reads it); a list starts afresh
void set_timing(std::optional<std::vector<std::string>> names) {
    std::lock_guard<std::mutex> l(g_timing.mu);
    if (names.has_value()) {
        for (auto &kv : g_timing.records)
            for (auto &p : kv.second) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
        g_timing.records.clear();
    }
    g_timing.on = names.has_value();
    g_timing.names = names.value_or(std::vector<std::string>{});
}

py::dict timing_summary() {
    hipDeviceSynchronize();
    std::lock_guard<std::mutex> l(g_timing.mu);
    py::dict out;
    for (auto &kv : g_timing.records) {
        double total = 0.0;
        for (auto &p : kv.second) { float ms = 0.f; hipEventElapsedTime(&ms, p.first, p.second); total += ms; }
        const size_t n = kv.second.size();
        out[py::str(kv.first)] = py::make_tuple(n, n ? total / n : 0.0);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------
// host readback of the few integers a call needs (sample totals, kept counts): the kernels store them straight
// into pinned host memory, so the readback is a stream wait plus a CPU load.  One slot per (device, stream,
// host thread): the calls stay re-entrant across threads and streams like the reference's stateless functions.
// ---------------------------------------------------------------------------------------------------
// false once a slot had to be allocated without hipHostMallocCoherent: a polling host may then never see a running kernel's
// store, so wait_stamp synchronises the stream straight away instead of spinning for its 2 ms first
std::atomic<bool> g_host_ints_coherent{true};

int64_t *host_ints(int device, hipStream_t s, int which = 0) {
    // which = 1: the slot of a count pass launched AHEAD of its call (sample_occgrid's next-chunk prefetch): the calls in between
    // (visibility_compact of the current chunk) keep using slot 0
    thread_local std::map<std::tuple<int, hipStream_t, int>, int64_t *> slots;
    auto key = std::make_tuple(device, s, which);
    auto it = slots.find(key);
    if (it != slots.end()) return it->second;
    int64_t *p = nullptr;
    // coherent (fine-grained) pinned memory: a kernel's store is visible to a polling host while the kernel runs (wait_stamp)
    if (hipHostMalloc(reinterpret_cast<void **>(&p), 8 * sizeof(int64_t), hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        TORCH_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p), 8 * sizeof(int64_t), hipHostMallocDefault) == hipSuccess,
                    "nerfacc_amd: hipHostMalloc failed");
        g_host_ints_coherent = false;
    }
    for (int i = 0; i < 8; ++i) p[i] = 0;
    if (slots.size() > 64) slots.clear();      // (slots of dead streams are leaked: 64 B each)
    slots[key] = p;
    return p;
}

// The read-backs of this file wait for ONE word: the kernel that produces the integers stores the caller's stamp behind them
// (system-scope fence), and the host polls that word instead of asking the runtime to synchronise the stream — 1-2 us instead of
// the 8-9 us a hipStreamSynchronize takes to come back (tools/sync_latency.py), and the host is released when the integers exist,
// not when the kernel that wrote them (and goes on to write its outputs) has finished.  Bounded: after ~2 ms of polling the
// stream is synchronised the ordinary way (a launch that failed, a size range whose kernels do not stamp).
inline int64_t next_stamp() {
    thread_local int64_t seq = 0;
    return ++seq;
}
inline void wait_stamp(const int64_t *slot, int64_t stamp, hipStream_t s) {
    py::gil_scoped_release nogil;
    const volatile int64_t *p = slot;
    const auto t0 = std::chrono::steady_clock::now();
    for (int64_t spins = 0; g_host_ints_coherent.load(std::memory_order_relaxed); ++spins) {
        if (*p == stamp) { std::atomic_thread_fence(std::memory_order_acquire); return; }
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    TORCH_CHECK(hipStreamSynchronize(s) == hipSuccess, "nerfacc_amd: hipStreamSynchronize failed");
}

inline void wait_stream(hipStream_t s) {
    py::gil_scoped_release nogil;               // other Python threads may run while this one waits for its stream
    TORCH_CHECK(hipStreamSynchronize(s) == hipSuccess, "nerfacc_amd: hipStreamSynchronize failed");
}

// ---------------------------------------------------------------------------------------------------
// occupancy bricks: packed once per distinct state of a `binaries` tensor (identity + in-place version)
// ---------------------------------------------------------------------------------------------------
struct BrickEntry {
    c10::weak_intrusive_ptr<c10::TensorImpl> ref;
    c10::TensorImpl *impl;
    uint32_t version;
    Tensor bricks;
    int64_t nonempty;
    int64_t level_counts[8];
    hipStream_t stream;
    hipEvent_t event;
};
std::mutex g_brick_mu;
std::vector<BrickEntry> g_bricks;     // most recently used first
constexpr size_t kBrickSlots = 4;

// caller holds g_brick_mu: makes `bricks` (packed on stream s) the front entry for this state of `binaries`
void insert_brick_entry(const Tensor &binaries, const Tensor &bricks, hipStream_t s) {
    c10::TensorImpl *impl = binaries.unsafeGetTensorImpl();
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    hipEventRecord(ev, s);
    // drop slots whose tensor is gone or is this tensor in an older state, then the oldest
    std::vector<BrickEntry> keep;
    for (auto &c : g_bricks) {
        if (c.ref.expired() || c.impl == impl) { hipEventDestroy(c.event); continue; }
        keep.push_back(std::move(c));
    }
    g_bricks = std::move(keep);
    BrickEntry e{c10::weak_intrusive_ptr<c10::TensorImpl>(binaries.getIntrusivePtr()), impl, binaries._version(), bricks, -1,
                 {0, 0, 0, 0, 0, 0, 0, 0}, s, ev};
    g_bricks.insert(g_bricks.begin(), std::move(e));
    while (g_bricks.size() > kBrickSlots) { hipEventDestroy(g_bricks.back().event); g_bricks.pop_back(); }
}

// returns (bricks, number of non-empty bricks); reads the count back once per grid state
std::pair<Tensor, int64_t> brick_entry(const Tensor &binaries, bool need_count) {
    check_input(binaries, "binaries", at::kBool);
    TORCH_CHECK(binaries.dim() == 4, "binaries must have shape [n_grids, resx, resy, resz]");
    Guard g(device_of(binaries));                       // events and the copy below belong to the tensor's device
    std::lock_guard<std::mutex> l(g_brick_mu);
    hipStream_t s = stream_of(binaries);
    c10::TensorImpl *impl = binaries.unsafeGetTensorImpl();
    const uint32_t ver = binaries._version();
    size_t hit = g_bricks.size();
    for (size_t k = 0; k < g_bricks.size(); ++k) {
        auto &c = g_bricks[k];
        if (c.impl == impl && c.version == ver && !c.ref.expired()) { hit = k; break; }
    }
    if (hit == g_bricks.size()) {
        const int G = (int)binaries.size(0), rx = (int)binaries.size(1), ry = (int)binaries.size(2), rz = (int)binaries.size(3);
        const int64_t words = nfa_packed_grid_words(G, rx, ry, rz);
        Tensor bricks = at::empty({words}, opts(binaries, at::kLong));
        check_rc(nfa_pack_binaries(ptr<uint8_t>(binaries), G, rx, ry, rz, ptr<uint64_t>(bricks), s));
        insert_brick_entry(binaries, bricks, s);
        hit = 0;
    } else if (hit != 0) {
        BrickEntry e = std::move(g_bricks[hit]);
        g_bricks.erase(g_bricks.begin() + hit);
        g_bricks.insert(g_bricks.begin(), std::move(e));
        hit = 0;
    }
    auto &c = g_bricks[0];
    if (c.stream != s) hipStreamWaitEvent(s, c.event, 0);      // packed on another stream: order this one after the pack
    if (need_count && c.nonempty < 0) {
        const int64_t nb = binaries.size(0) * ((binaries.size(1) + 3) / 4) * ((binaries.size(2) + 3) / 4) * ((binaries.size(3) + 3) / 4);
        // header of the packed grid: [0] non-empty bricks, [1..8] occupied voxels per level
        int64_t hdr[9];
        TORCH_CHECK(hipMemcpyAsync(hdr, ptr<int64_t>(c.bricks) + nb, sizeof(hdr), hipMemcpyDeviceToHost, s) == hipSuccess,
                    "nerfacc_amd: readback of the packed-grid header failed");
        // (the GIL stays held: releasing it while holding g_brick_mu would let another Python thread block on the mutex
        // with the GIL, and this thread could then never take the GIL back.  Once per grid state, ~20 us.)
        TORCH_CHECK(hipStreamSynchronize(s) == hipSuccess, "nerfacc_amd: hipStreamSynchronize failed");
        c.nonempty = hdr[0];
        for (int g = 0; g < 8; ++g) c.level_counts[g] = hdr[1 + g];
    }
    return {c.bricks, c.nonempty};
}

Tensor packed_bricks(const Tensor &binaries) { return brick_entry(binaries, true).first; }

// occupied voxels per level of `binaries` (what nonzero(binaries[level]) would count), from the packed grid's header: one
// read-back per grid state, shared with the traversal's
std::vector<int64_t> grid_occupied_counts(const Tensor &binaries) {
    brick_entry(binaries, true);
    std::lock_guard<std::mutex> l(g_brick_mu);
    c10::TensorImpl *impl = binaries.unsafeGetTensorImpl();
    for (auto &c : g_bricks)
        if (c.impl == impl && c.version == binaries._version() && !c.ref.expired())
            return std::vector<int64_t>(c.level_counts, c.level_counts + std::min<int64_t>(binaries.size(0), 8));
    TORCH_CHECK(false, "nerfacc_amd: packed grid vanished from the cache");
}

void *sync_block(const Tensor &like, hipStream_t s);

// the occupied cells of level `lvl`, ascending (torch.nonzero(binaries[lvl].flatten())[:, 0], occ_grid.py:356) without rocprim's
// partition and without a read-back: the count comes from the packed grid's header
Tensor grid_occupied_cells(const Tensor &binaries, int64_t lvl) {
    check_input(binaries, "binaries", at::kBool);
    TORCH_CHECK(binaries.dim() == 4 && lvl >= 0 && lvl < binaries.size(0), "binaries must be [levels, rx, ry, rz] and lvl one of its levels");
    const std::vector<int64_t> counts = grid_occupied_counts(binaries);
    TORCH_CHECK(lvl < (int64_t)counts.size(), "grid_occupied_cells: level counts are kept for the first 8 levels");
    const int64_t n_cells = binaries.size(1) * binaries.size(2) * binaries.size(3), cnt = counts[lvl];
    Tensor out = at::empty({cnt}, opts(binaries, at::kLong));
    if (cnt == 0) return out;
    Guard g(device_of(binaries));
    hipStream_t s = stream_of(binaries);
    check_rc(nfa_grid_occupied_cells((const uint8_t *)binaries.data_ptr() + lvl * n_cells, n_cells, ptr<int64_t>(out), cnt, sync_block(binaries, s), s));
    return out;
}

// ---------------------------------------------------------------------------------------------------
// RaySegmentsSpec (data_spec.hpp:6-14; nerfacc.cpp:128-137): seven optional tensors
// ---------------------------------------------------------------------------------------------------
struct RaySegmentsSpec {
    OptTensor vals, is_left, is_right, is_valid, chunk_starts, chunk_cnts, ray_indices;

    void check() const {      // data_spec.hpp:15-51
        TORCH_CHECK(vals.has_value(), "RaySegmentsSpec.vals is not set");
        check_input(*vals, "vals", at::kFloat);
        if (vals->dim() > 1) return;
        TORCH_CHECK(chunk_starts && chunk_cnts, "flattened RaySegmentsSpec needs chunk_starts and chunk_cnts");
        check_input(*chunk_starts, "chunk_starts", at::kLong);
        check_input(*chunk_cnts, "chunk_cnts", at::kLong);
        TORCH_CHECK(chunk_starts->dim() == 1 && chunk_cnts->dim() == 1, "chunk_starts / chunk_cnts must be 1-D");
        TORCH_CHECK(chunk_starts->numel() == chunk_cnts->numel(), "chunk_starts and chunk_cnts differ in length");
        auto same = [&](const OptTensor &t, const char *n, at::ScalarType dt) {
            if (!t) return;
            check_input(*t, n, dt);
            TORCH_CHECK(t->dim() == 1 && t->numel() == vals->numel(), n, " must be 1-D with as many elements as vals");
        };
        same(ray_indices, "ray_indices", at::kLong);
        same(is_left, "is_left", at::kBool);
        same(is_right, "is_right", at::kBool);
        same(is_valid, "is_valid", at::kBool);
    }

    nfa_ray_segments view() const {
        nfa_ray_segments s{};
        s.vals = ptr<float>(vals);
        s.n_edges = vals->numel();
        if (vals->dim() > 1) {
            s.n_edges_per_ray = vals->size(-1);
            s.n_rays = vals->numel() / std::max<int64_t>(vals->size(-1), 1);
        } else {
            s.chunk_starts = ptr<int64_t>(chunk_starts);
            s.chunk_cnts = ptr<int64_t>(chunk_cnts);
            s.ray_indices = ptr<int64_t>(ray_indices);
            s.n_rays = chunk_cnts->numel();
        }
        return s;
    }
};

// ---------------------------------------------------------------------------------------------------
// grid
// ---------------------------------------------------------------------------------------------------
std::vector<Tensor> ray_aabb_intersect(const Tensor &rays_o, const Tensor &rays_d, const Tensor &aabbs, double near_plane,
                                       double far_plane, double miss_value) {   // nerfacc.cpp:63-69
    check_input(rays_o, "rays_o", at::kFloat);
    check_input(rays_d, "rays_d", at::kFloat);
    check_input(aabbs, "aabbs", at::kFloat);
    const int64_t R = rays_o.size(0), G = aabbs.size(0);
    Tensor t_mins = at::empty({R, G}, rays_o.options()), t_maxs = at::empty({R, G}, rays_o.options());
    Tensor hits = at::empty({R, G}, opts(rays_o, at::kBool));
    Guard g(device_of(rays_o));
    check_rc(nfa_ray_aabb_intersect(ptr<float>(rays_o), ptr<float>(rays_d), R, ptr<float>(aabbs), G, (float)near_plane,
                                    (float)far_plane, (float)miss_value, ptr<float>(t_mins), ptr<float>(t_maxs), ptr<uint8_t>(hits),
                                    stream_of(rays_o)));
    return {t_mins, t_maxs, hits};
}

nfa_traverse_args traverse_args(const Tensor &rays_o, const Tensor &rays_d, const OptTensor &rays_mask, const Tensor &binaries,
                                const Tensor &aabbs, const OptTensor &t_sorted, const OptTensor &t_indices, const OptTensor &hits,
                                const OptTensor &near_planes, const OptTensor &far_planes, double step_size, double cone_angle,
                                int64_t limit, Tensor &bricks_keepalive, double near_plane = 0.0,
                                double far_plane = std::numeric_limits<double>::infinity(), const OptTensor &t_min = std::nullopt,
                                const OptTensor &t_max = std::nullopt, const OptTensor &jitter = std::nullopt, double jitter_scale = 0.0) {
    check_input(rays_o, "rays_o", at::kFloat);
    check_input(rays_d, "rays_d", at::kFloat);
    check_input(aabbs, "aabbs", at::kFloat);
    const int64_t R = rays_o.size(0), G = binaries.size(0);
    auto per_ray = [&](const OptTensor &t, const char *name) {
        if (!t) return;
        check_input(*t, name, at::kFloat);
        TORCH_CHECK(t->numel() == R, name, " must have n_rays elements");
    };
    per_ray(near_planes, "near_planes");
    per_ray(far_planes, "far_planes");
    per_ray(t_min, "t_min");
    per_ray(t_max, "t_max");
    per_ray(jitter, "jitter");
    TORCH_CHECK(rays_o.dim() == 2 && rays_o.size(1) == 3 && rays_d.sizes() == rays_o.sizes(), "rays_o / rays_d must have shape [n_rays, 3]");
    TORCH_CHECK(rays_d.device() == rays_o.device() && binaries.device() == rays_o.device() && aabbs.device() == rays_o.device(),
                "rays_o, rays_d, binaries and aabbs must live on the same device");
    TORCH_CHECK(aabbs.dim() == 2 && aabbs.size(0) == G && aabbs.size(1) == 6, "aabbs must have shape [n_grids, 6]");
    nfa_traverse_args a{};
    a.near_plane = (float)near_plane;
    a.far_plane = (float)far_plane;
    a.t_min = ptr<float>(t_min);
    a.t_max = ptr<float>(t_max);
    a.jitter = ptr<float>(jitter);
    a.jitter_scale = (float)jitter_scale;
    a.n_rays = R;
    a.rays_o = ptr<float>(rays_o);
    a.rays_d = ptr<float>(rays_d);
    if (rays_mask) {
        check_input(*rays_mask, "rays_mask", at::kBool);
        a.rays_mask = ptr<uint8_t>(*rays_mask);
    }
    auto be = brick_entry(binaries, true);
    bricks_keepalive = be.first;
    a.n_grids = (int32_t)G;
    a.res[0] = (int32_t)binaries.size(1); a.res[1] = (int32_t)binaries.size(2); a.res[2] = (int32_t)binaries.size(3);
    a.bricks = ptr<uint64_t>(be.first);
    a.n_nonempty_bricks = be.second;
    a.aabbs = ptr<float>(aabbs);
    if (t_sorted) {
        TORCH_CHECK(t_indices && hits, "t_sorted, t_indices and hits must be given together");
        check_input(*t_sorted, "t_sorted", at::kFloat);
        check_input(*t_indices, "t_indices", at::kLong);
        check_input(*hits, "hits", at::kBool);
        TORCH_CHECK(t_sorted->dim() == 2 && t_sorted->size(0) == R && t_sorted->size(1) == 2 * G && t_indices->sizes() == t_sorted->sizes()
                        && hits->dim() == 2 && hits->size(0) == R && hits->size(1) == G,
                    "t_sorted/t_indices must be [n_rays, 2*n_grids], hits [n_rays, n_grids]");
        a.t_sorted = ptr<float>(*t_sorted);
        a.t_indices = ptr<int64_t>(*t_indices);
        a.hits = ptr<uint8_t>(*hits);
    }
    a.near_planes = ptr<float>(near_planes);
    a.far_planes = ptr<float>(far_planes);
    a.step_size = (float)step_size;
    a.cone_angle = (float)cone_angle;
    a.traverse_steps_limit = (int32_t)limit;
    return a;
}

// nerfacc.cpp:71-98 / grid.cu:320-474.  Two-pass mode: count -> offsets (device) -> ONE readback -> allocate -> fill; as
// in the reference it ignores rays_mask (grid.cu:418,450).
std::tuple<RaySegmentsSpec, RaySegmentsSpec, OptTensor> traverse_grids(
    const Tensor &rays_o, const Tensor &rays_d, const Tensor &rays_mask, const Tensor &binaries, const Tensor &aabbs,
    const OptTensor &t_sorted, const OptTensor &t_indices, const OptTensor &hits, const Tensor &near_planes, const Tensor &far_planes,
    double step_size, double cone_angle, bool compute_intervals, bool compute_samples, bool compute_terminate_planes,
    int64_t traverse_steps_limit, bool over_allocate) {
    TORCH_CHECK(!over_allocate || traverse_steps_limit > 0, "traverse_steps_limit must be > 0 when over_allocate is true");   // grid.cu:345
    check_input(rays_o, "rays_o", at::kFloat);
    const int64_t R = rays_o.size(0);
    const auto i64 = opts(rays_o, at::kLong), f32 = rays_o.options(), b8 = opts(rays_o, at::kBool);
    RaySegmentsSpec intervals, samples;
    OptTensor terminate;
    if (compute_terminate_planes) terminate = at::empty({R}, f32);
    Guard g(device_of(rays_o));
    hipStream_t s = stream_of(rays_o);
    Tensor keep;
    nfa_traverse_args a = traverse_args(rays_o, rays_d, over_allocate ? OptTensor(rays_mask) : std::nullopt, binaries, aabbs, t_sorted,
                                        t_indices, hits, OptTensor(near_planes), OptTensor(far_planes), step_size, cone_angle,
                                        traverse_steps_limit, keep);
    Tensor iv_cnts, iv_starts, sm_cnts, sm_starts, ws;
    int64_t n_edges = 0, n_samples = 0, n_overflow = 0;
    if (over_allocate) {
        // grid.cu:364-404: fixed-size slots, single pass, then starts from the actual counts
        Tensor maskl = rays_mask.to(at::kLong);
        iv_cnts = maskl * (2 * traverse_steps_limit);
        sm_cnts = maskl * traverse_steps_limit;
        iv_starts = at::empty_like(iv_cnts);
        sm_starts = at::empty_like(sm_cnts);
        int64_t *h = host_ints(rays_o.device().index(), s);
        check_rc(nfa_exclusive_sum_i64(ptr<int64_t>(iv_cnts), R, ptr<int64_t>(iv_starts), h, s));
        check_rc(nfa_exclusive_sum_i64(ptr<int64_t>(sm_cnts), R, ptr<int64_t>(sm_starts), h + 1, s));
        wait_stream(s);
        n_edges = h[0];
        n_samples = h[1];
    } else {
        if (compute_intervals) { iv_cnts = at::empty({R}, i64); iv_starts = at::empty({R}, i64); }
        sm_cnts = at::empty({R}, i64);
        sm_starts = at::empty({R}, i64);
        a.workspace_bytes = nfa_traverse_workspace_bytes_for(&a);
        ws = at::empty({std::max<int64_t>(a.workspace_bytes, 16)}, opts(rays_o, at::kByte));
        int64_t *h = host_ints(rays_o.device().index(), s);
        a.iv_cnts = ptr<int64_t>(iv_cnts); a.iv_starts = ptr<int64_t>(iv_starts);
        a.sm_cnts = ptr<int64_t>(sm_cnts); a.sm_starts = ptr<int64_t>(sm_starts);
        a.totals = h;
        a.terminate_planes = ptr<float>(terminate);
        check_rc(nfa_traverse_count(&a, ws.data_ptr(), s));
        check_rc(nfa_traverse_offsets(&a, ws.data_ptr(), s));
        wait_stream(s);                                       // the one host sync (data_spec.hpp:91)
        n_edges = h[0]; n_samples = h[1]; n_overflow = h[2];
    }
    a.iv_cnts = ptr<int64_t>(iv_cnts); a.iv_starts = ptr<int64_t>(iv_starts);
    a.sm_cnts = ptr<int64_t>(sm_cnts); a.sm_starts = ptr<int64_t>(sm_starts);
    auto alloc = [&](int64_t n, const at::TensorOptions &o) { return over_allocate ? at::zeros({n}, o) : at::empty({n}, o); };
    if (compute_intervals) {
        intervals.vals = alloc(n_edges, f32);
        intervals.ray_indices = alloc(n_edges, i64);
        Tensor flags = at::zeros({2, n_edges}, b8);
        intervals.is_left = flags[0];
        intervals.is_right = flags[1];
        a.iv_vals = ptr<float>(intervals.vals); a.iv_ray_indices = ptr<int64_t>(intervals.ray_indices);
        a.iv_is_left = ptr<uint8_t>(intervals.is_left); a.iv_is_right = ptr<uint8_t>(intervals.is_right);
    }
    if (compute_samples) {
        samples.vals = alloc(n_samples, f32);
        samples.ray_indices = alloc(n_samples, i64);
        samples.is_valid = alloc(n_samples, b8);
        a.sm_vals = ptr<float>(samples.vals); a.sm_ray_indices = ptr<int64_t>(samples.ray_indices);
        a.sm_is_valid = ptr<uint8_t>(samples.is_valid);
    }
    a.terminate_planes = ptr<float>(terminate);
    if (over_allocate) {
        if (R > 0) check_rc(nfa_traverse_fill(&a, 0, 1, nullptr, 0, 0, s));
        check_rc(nfa_exclusive_sum_i64(ptr<int64_t>(iv_cnts), R, ptr<int64_t>(iv_starts), nullptr, s));
        check_rc(nfa_exclusive_sum_i64(ptr<int64_t>(sm_cnts), R, ptr<int64_t>(sm_starts), nullptr, s));
    } else if (R > 0 && (compute_intervals || compute_samples) && n_samples > 0) {
        check_rc(nfa_traverse_fill(&a, 1, 0, ws.data_ptr(), n_samples, n_overflow, s));
    }
    if (compute_intervals) { intervals.chunk_cnts = iv_cnts; intervals.chunk_starts = iv_starts; }
    if (compute_samples) { samples.chunk_cnts = sm_cnts; samples.chunk_starts = sm_starts; }
    return {intervals, samples, terminate};
}

// k float rows of n elements from one allocation, each row 16-byte aligned (the tiled kernels then use vector accesses: 2 or 4 elements per lane)
struct Rows {
    Tensor buf;
    int64_t pitch, n;
    Rows(int64_t k, int64_t n_, const at::TensorOptions &o) : pitch((n_ + 3) & ~int64_t(3)), n(n_) { buf = at::empty({k, pitch}, o); }
    Rows(Tensor adopted, int64_t n_) : buf(std::move(adopted)), pitch(buf.size(1)), n(n_) {}      // a [k, pitch] tensor allocated ahead
    float *p(int64_t r) const { return buf.data_ptr<float>() + r * pitch; }
    Tensor row(int64_t r) const { return buf[r].narrow(0, 0, n); }
};

// The traversal's scratch (run records, block sums: nfa_traverse_workspace_bytes_for, up to 256 MB) is the same few MB call after
// call: one slab per (host thread, device, stream), grown when a call needs more, instead of an allocation per call.  Everything that
// touches it — count, offsets, emit of one call, then the next call's — is enqueued on that one stream, in order.
// A call that needs more than kWorkspaceKeep gets its own allocation (stream-ordered by torch's caching allocator, returned to it when
// the call ends): one frame-sized call does not pin a quarter of a GB for the life of the thread.
constexpr int64_t kWorkspaceKeep = 64ll << 20;
using SlabMap = std::map<std::pair<int, hipStream_t>, Tensor>;
SlabMap &workspace_slabs() {
    thread_local SlabMap slabs;
    return slabs;
}
Tensor traverse_workspace(const Tensor &like, hipStream_t s, int64_t bytes) {
    if (bytes > kWorkspaceKeep) return at::empty({bytes}, opts(like, at::kByte));
    Tensor &t = workspace_slabs()[{(int)like.device().index(), s}];
    if (!t.defined() || t.numel() < bytes)
        t = at::empty({std::min<int64_t>(std::max<int64_t>(bytes + bytes / 4, 1 << 16), kWorkspaceKeep)}, opts(like, at::kByte));
    return t;
}
// drops this host thread's retained slabs (and with them any count pass launched ahead: its records lived there)
void release_workspace();

// The sync block of the single-launch forms (include/nerfacc_hip.h: NFA_SYNC_BYTES, zero before its first use, left zero by every
// kernel that uses it): one per (host thread, device, stream), allocated zero-filled once.  Everything that touches it is enqueued
// on that one stream, in order.
SlabMap &sync_blocks() {
    thread_local SlabMap blocks;
    return blocks;
}
// Outputs of the NEXT single-launch sampling call of the same kind, allocated while the current call's launch is counting: the
// single launch needs its outputs to exist before it starts (the three-kernel form allocates them behind the count kernel's launch,
// for free), and two allocations in front of the launch are 4-5 us on the step's critical path — host wake-up to host wake-up is what
// bounds a training step at this size (tools/path_ab.py).  One set per (host thread, device, kind of call), at most kNextOutKeep bytes.
constexpr int64_t kNextOutKeep = 64ll << 20;
struct NextOut {
    Tensor ray_indices, rows;      // [cap] int64, [2, pitch] float
    int64_t cap = 0;
    hipStream_t stream = nullptr;
};
std::map<int, NextOut> &next_outputs() {
    thread_local std::map<int, NextOut> m;
    return m;
}

void *sync_block(const Tensor &like, hipStream_t s) {
    Tensor &t = sync_blocks()[{(int)like.device().index(), s}];
    if (!t.defined()) t = at::zeros({(int64_t)NFA_SYNC_BYTES}, opts(like, at::kByte));
    return t.data_ptr();
}

// The reference's eval loop (examples/utils.py:80-88) calls `estimator.sampling` on consecutive 8192-ray SLICES of one ray array, and
// every call has to wait for its count pass before it can size its outputs (~40 us of 205 per chunk with the host idle, profiles/
// r03_host_share.md).  When a call's rays start exactly where the previous call's ended — the second consecutive slice — the count
// and offsets kernels of the NEXT slice are launched at the end of the call, behind its own work; the next call, if it asks for
// exactly those rays (same pointers and count, same tensor versions, same grid, same scalars), finds its totals waiting.  A guess
// that is not taken up costs one count pass of GPU time and nothing else: nothing it wrote is looked at.  Only plain calls qualify
// (no jitter, masks, per-ray planes or step limits: training batches are fresh tensors and never match).  `chunk_prefetch` = 0
// switches it off.
struct ChunkPrefetch {
    bool valid = false;
    const void *o_ptr = nullptr, *d_ptr = nullptr, *aabbs_ptr = nullptr;
    const void *bin_impl = nullptr;
    int64_t R = 0;
    uint32_t o_ver = 0, d_ver = 0, bin_ver = 0, aabbs_ver = 0;
    double step = 0, cone = 0, near_plane = 0, far_plane = 0;
    int64_t stamp = 0, ws_bytes = 0;
    Tensor ws_keep;                     // the workspace the guess's run records live in (held: the taking call uses exactly this one)
    Tensor packed;
    const void *expect_o = nullptr;     // where the next slice of the caller's loop would start (set by every qualifying call)
    // The guess is keyed on IDENTITY, not only on addresses: weak references to the storages of both ray arrays, of the boxes and to
    // the grid tensor.  A weak reference keeps the (small) StorageImpl / TensorImpl object alive, so a live guess's pointer cannot be
    // handed to another object, and an expired one says the memory behind the recorded address may have been freed and re-used (a
    // fresh tensor at the same address starts at version 0 again: ADVICE r4).  It does not keep the caller's memory alive.
    c10::weak_intrusive_ptr<c10::StorageImpl> o_st{c10::intrusive_ptr<c10::StorageImpl>()}, d_st{c10::intrusive_ptr<c10::StorageImpl>()},
        aabbs_st{c10::intrusive_ptr<c10::StorageImpl>()};
    c10::weak_intrusive_ptr<c10::TensorImpl> bin_ref{c10::intrusive_ptr<c10::TensorImpl>()};
    bool same_storage(const c10::weak_intrusive_ptr<c10::StorageImpl> &w, const Tensor &t) const {
        return !w.expired() && w._unsafe_get_target() == t.storage().unsafeGetStorageImpl();
    }
    void drop() {
        valid = false;
        packed = Tensor();
        ws_keep = Tensor();
        o_st = d_st = aabbs_st = c10::weak_intrusive_ptr<c10::StorageImpl>(c10::intrusive_ptr<c10::StorageImpl>());
        bin_ref = c10::weak_intrusive_ptr<c10::TensorImpl>(c10::intrusive_ptr<c10::TensorImpl>());
    }
};
std::map<std::pair<int, hipStream_t>, ChunkPrefetch> &prefetch_slots() {
    thread_local std::map<std::pair<int, hipStream_t>, ChunkPrefetch> slots;
    return slots;
}
ChunkPrefetch &chunk_prefetch(int device, hipStream_t s) {
    auto &slots = prefetch_slots();
    if (slots.size() > 64) slots.clear();
    return slots[{device, s}];
}
void release_workspace() {
    prefetch_slots().clear();
    workspace_slabs().clear();
    sync_blocks().clear();
    next_outputs().clear();
}
inline bool chunk_prefetch_allowed() {
    int64_t v = 1;
    int32_t is_set = 0;
    nfa_get_option("chunk_prefetch", &v, &is_set);
    return !is_set || v != 0;
}

// traverse_grids + the two is_left / is_right compactions of occ_grid.py:164-177 in one count pass and one emit pass:
// (ray_indices, t_starts, t_ends, packed_info[, terminate_planes]).  rays_mask / traverse_steps_limit give one round of the
// test-time marcher (examples/utils.py:349-372) with exactly sized outputs.
py::tuple sample_occgrid(const Tensor &rays_o, const Tensor &rays_d, const Tensor &binaries, const Tensor &aabbs,
                         const OptTensor &near_planes, const OptTensor &far_planes, double step_size, double cone_angle,
                         const OptTensor &rays_mask, int64_t traverse_steps_limit, bool with_terminate_planes, double near_plane,
                         double far_plane, const OptTensor &t_min, const OptTensor &t_max, const OptTensor &jitter, double jitter_scale) {
    check_input(rays_o, "rays_o", at::kFloat);
    const int64_t R = rays_o.size(0);
    const auto i64 = opts(rays_o, at::kLong), f32 = rays_o.options();
    Guard g(device_of(rays_o));
    hipStream_t s = stream_of(rays_o);
    Tensor keep;
    nfa_traverse_args a = traverse_args(rays_o, rays_d, rays_mask, binaries, aabbs, std::nullopt, std::nullopt, std::nullopt,
                                        near_planes, far_planes, step_size, cone_angle, traverse_steps_limit, keep, near_plane, far_plane,
                                        t_min, t_max, jitter, jitter_scale);
    // a plain call (what the reference's eval loop makes): may take up, and leave behind, a count pass launched ahead
    // (tensors created under torch.inference_mode() carry no version counter — `_version()` throws for them — so a call with any
    //  of them neither takes up nor leaves a guess: ADVICE r4.  Writes that bypass the version counter — `x.data.copy_`, raw-pointer
    //  kernels, DLPack consumers — are invisible to this key: such callers set chunk_prefetch = 0.)
    const bool plain = R > 0 && !rays_mask && !near_planes && !far_planes && !t_min && !t_max && !jitter && traverse_steps_limit <= 0 &&
                       !with_terminate_planes && !rays_o.is_inference() && !rays_d.is_inference() && !aabbs.is_inference() &&
                       !binaries.is_inference();
    ChunkPrefetch &pf = chunk_prefetch(rays_o.device().index(), s);
    a.workspace_bytes = nfa_traverse_workspace_bytes_for(&a);
    const bool taken = plain && pf.valid && pf.o_ptr == rays_o.data_ptr() && pf.d_ptr == rays_d.data_ptr() && pf.R == R &&
                       pf.same_storage(pf.o_st, rays_o) && pf.same_storage(pf.d_st, rays_d) && pf.same_storage(pf.aabbs_st, aabbs) &&
                       !pf.bin_ref.expired() && pf.bin_impl == (const void *)binaries.unsafeGetTensorImpl() &&
                       pf.o_ver == rays_o._version() && pf.d_ver == rays_d._version() && pf.bin_ver == binaries._version() &&
                       pf.aabbs_ptr == aabbs.data_ptr() && pf.aabbs_ver == aabbs._version() && pf.step == step_size &&
                       pf.cone == cone_angle && pf.near_plane == near_plane && pf.far_plane == far_plane &&
                       pf.ws_bytes == a.workspace_bytes && pf.ws_keep.defined();
    Tensor ws = taken ? pf.ws_keep : traverse_workspace(rays_o, s, a.workspace_bytes);
    Tensor packed = taken ? pf.packed : at::empty({2, R}, i64);           // [starts; cnts], handed out transposed as [R, 2]
    const int64_t pf_stamp = pf.stamp;
    pf.drop();                                        // taken up or stale: either way it is spent
    int64_t *h = host_ints(rays_o.device().index(), s, taken ? 1 : 0);
    a.sm_starts = ptr<int64_t>(packed);
    a.sm_cnts = ptr<int64_t>(packed) + R;
    a.totals = h;
    Tensor term;
    if (with_terminate_planes) {
        TORCH_CHECK(near_planes.has_value(), "with_terminate_planes needs near_planes as a tensor");
        term = near_planes->clone();
        a.terminate_planes = ptr<float>(term);
    }
    // The emit pass runs BEFORE the read-back, into outputs sized from the previous call on this device (training
    // steps draw about the same number of samples every time): the GPU goes from counting straight into it
    // instead of idling through the host's wake-up, allocation and launch (16 us, tools/step_timeline.py), and the host —
    // released by the stamp behind the totals, not by the end of the emit pass — prepares the caller's next kernels while it
    // runs.  A guess that is too small costs nothing but the second launch the old order always needed.
    // The guess is SAMPLES PER RAY of the previous call of the same kind on this device (training batches, eval chunks and
    // marcher rounds differ by orders of magnitude in size but much less in samples per ray), times this call's ray count.
    thread_local std::map<int, double> last_spr;
    const int dev = rays_o.device().index() * 4 + (rays_mask.has_value() ? 1 : 0) + (traverse_steps_limit > 0 ? 2 : 0);
    const bool speculate = R > 0 && last_spr.count(dev) && last_spr[dev] > 0.0 && speculative_emit_allowed();
    int64_t cap = speculate ? (int64_t)(1.25 * last_spr[dev] * (double)R) + 1024 : 0;
    Tensor ray_indices;
    Rows ts(2, 0, f32);
    auto speculative_outputs = [&]() {
        ts = Rows(2, cap, f32);
        if (!speculate) return;
        ray_indices = at::empty({cap}, i64);
        a.sm_ray_indices = ptr<int64_t>(ray_indices);
        a.t_starts = ts.p(0);
        a.t_ends = ts.p(1);
    };
    int64_t stamp;
    if (taken) {
        stamp = pf_stamp;                             // count and offsets of exactly this call ran behind the previous one
        speculative_outputs();
        a.terminate_planes = nullptr;
        if (speculate) {
            Timed t("traverse_fill", s);
            check_rc(nfa_traverse_emit_speculative(&a, ws.data_ptr(), cap, s));
        }
    } else if (nfa_traverse_sample_fused(&a)) {
        // ONE launch counts, forms the offsets (look-back over the count kernel's workgroups) and expands every wave's own rays
        // (round 6; three launches before): the outputs have to exist first — normally they do, allocated behind the previous
        // call's launch (next_outputs)
        NextOut &no = next_outputs()[dev];
        if (speculate && no.ray_indices.defined() && no.stream == s && no.cap >= cap && no.cap <= 2 * cap + 4096) {
            cap = no.cap;
            ray_indices = std::move(no.ray_indices);
            ts = Rows(std::move(no.rows), cap);
            a.sm_ray_indices = ptr<int64_t>(ray_indices);
            a.t_starts = ts.p(0);
            a.t_ends = ts.p(1);
        } else {
            speculative_outputs();
        }
        no = NextOut{};
        stamp = next_stamp();
        h[3] = 0;
        int32_t did = 0;
        {
            Timed t("traverse_sample", s);
            check_rc(nfa_traverse_sample(&a, ws.data_ptr(), cap, stamp, sync_block(rays_o, s), &did, s));
        }
        a.terminate_planes = nullptr;                 // written by the count pass only
        if (speculate && 16 * cap <= kNextOutKeep) {  // (while the launch counts)
            no.cap = cap;
            no.stream = s;
            no.ray_indices = at::empty({cap}, i64);
            no.rows = at::empty({2, (cap + 3) & ~int64_t(3)}, f32);
        }
    } else {
        {
            Timed t("traverse_count", s);
            check_rc(nfa_traverse_count(&a, ws.data_ptr(), s));
        }
        stamp = next_stamp();
        h[3] = 0;
        check_rc(nfa_traverse_offsets_stamped(&a, ws.data_ptr(), stamp, s));
        speculative_outputs();                        // (allocated while the count pass runs)
        a.terminate_planes = nullptr;                 // written by the count pass only
        if (speculate) {
            Timed t("traverse_fill", s);
            check_rc(nfa_traverse_emit_speculative(&a, ws.data_ptr(), cap, s));
        }
    }
    if (R > 0) wait_stamp(h + 3, stamp, s);
    else wait_stream(s);
    bool emitted = speculate;
    if (R > 0 && h[1] < 0) {
        // the fused launch's look-back gave up (a bounded wait: include/nerfacc_hip.h); counts and run records are complete —
        // offsets by their own kernel, emit pass below
        stamp = next_stamp();
        h[3] = 0;
        check_rc(nfa_traverse_offsets_stamped(&a, ws.data_ptr(), stamp, s));
        wait_stamp(h + 3, stamp, s);
        emitted = false;
    }
    const int64_t n = h[1], n_overflow = h[2];
    // (totals that cannot be — seen once in ~60 runs of eight processes sharing one GPU, when the count launch lost one XCD's share of
    //  its workgroups' stores, profiles/r06_oversubscription.md: an error the caller can see instead of outputs sized by garbage)
    TORCH_CHECK(n >= 0 && n_overflow >= 0 && n_overflow <= R && h[0] >= n,
                "nerfacc_amd: sample_occgrid read back inconsistent totals (edges ", h[0], ", samples ", n, ", overflow rays ", n_overflow, " of ", R,
                " rays): the count pass's outputs are corrupt");
    last_spr[dev] = R > 0 ? (double)n / (double)R : 0.0;
    if (emitted && n <= cap) {
        if (n_overflow > 0) check_rc(nfa_traverse_fill(&a, 1, 0, ws.data_ptr(), 0, n_overflow, s));     // the rays the count pass flagged
        ray_indices = ray_indices.narrow(0, 0, n);
        ts.n = n;
        if (cap > (1 << 20) && n < cap / 4) {
            // a guess far too large: the caller would keep the whole over-sized storage alive through these views
            // (and torch.save would write it) — hand out right-sized copies instead
            ray_indices = ray_indices.clone();
            Rows exact(2, n, f32);
            exact.row(0).copy_(ts.row(0));
            exact.row(1).copy_(ts.row(1));
            ts = exact;
        }
    } else {
        ray_indices = at::empty({n}, i64);
        ts = Rows(2, n, f32);
        a.sm_ray_indices = ptr<int64_t>(ray_indices);
        a.t_starts = ts.p(0);
        a.t_ends = ts.p(1);
        if (n > 0) {
            Timed t("traverse_fill", s);
            check_rc(nfa_traverse_fill(&a, 1, 0, ws.data_ptr(), n, n_overflow, s));
        }
    }
    if (plain) {
        // the next slice of the caller's loop, if this call continued one (its rays start where the previous call's ended) and both
        // ray arrays have rows left behind this slice
        const float *o_end = ptr<float>(rays_o) + 3 * R, *d_end = ptr<float>(rays_d) + 3 * R;
        const bool continued = pf.expect_o == rays_o.data_ptr();
        pf.expect_o = o_end;
        auto rows_left = [&](const Tensor &t) {
            const int64_t used = (t.storage_offset() + 3 * R) * 4;
            return ((int64_t)t.storage().nbytes() - used) / 12;
        };
        const int64_t Rn = std::min<int64_t>(R, std::min(rows_left(rays_o), rows_left(rays_d)));
        nfa_traverse_args a2 = a;
        a2.n_rays = Rn;
        a2.rays_o = o_end;
        a2.rays_d = d_end;
        a2.sm_ray_indices = nullptr; a2.t_starts = nullptr; a2.t_ends = nullptr; a2.terminate_planes = nullptr;
        if (Rn > 0) a2.workspace_bytes = nfa_traverse_workspace_bytes_for(&a2);
        // (a slice whose workspace is beyond what a thread retains gets a per-call allocation: a guess would pin it between calls and
        //  behind the loop's last slice — exactly what the cap is there to prevent, ADVICE r5: no count pass ahead for such slices)
        if (continued && Rn > 0 && chunk_prefetch_allowed() && a2.workspace_bytes <= kWorkspaceKeep) {
            pf.packed = at::empty({2, Rn}, i64);
            a2.sm_starts = ptr<int64_t>(pf.packed);
            a2.sm_cnts = ptr<int64_t>(pf.packed) + Rn;
            int64_t *h2 = host_ints(rays_o.device().index(), s, 1);
            a2.totals = h2;
            Tensor ws2 = traverse_workspace(rays_o, s, a2.workspace_bytes);      // (the same slab: this call's kernels are enqueued before)
            pf.stamp = next_stamp();
            h2[3] = 0;
            // count + offsets of the next slice: one launch where the fused form exists (capacity 0: nothing is expanded), else two
            check_rc(nfa_traverse_sample(&a2, ws2.data_ptr(), 0, pf.stamp, sync_block(rays_o, s), nullptr, s));
            pf.o_ptr = o_end; pf.d_ptr = d_end; pf.R = Rn;
            pf.o_ver = rays_o._version(); pf.d_ver = rays_d._version();
            pf.bin_impl = (const void *)binaries.unsafeGetTensorImpl(); pf.bin_ver = binaries._version();
            pf.aabbs_ptr = aabbs.data_ptr(); pf.aabbs_ver = aabbs._version();
            pf.step = step_size; pf.cone = cone_angle; pf.near_plane = near_plane; pf.far_plane = far_plane;
            pf.ws_bytes = a2.workspace_bytes;
            pf.ws_keep = ws2;
            pf.o_st = rays_o.storage().getWeakStorageImpl();
            pf.d_st = rays_d.storage().getWeakStorageImpl();
            pf.aabbs_st = aabbs.storage().getWeakStorageImpl();
            pf.bin_ref = c10::weak_intrusive_ptr<c10::TensorImpl>(binaries.getIntrusivePtr());
            pf.valid = true;
        }
    } else {
        pf.expect_o = nullptr;
    }
    if (with_terminate_planes) return py::make_tuple(ray_indices, ts.row(0), ts.row(1), packed.t(), term);
    return py::make_tuple(ray_indices, ts.row(0), ts.row(1), packed.t());
}

// ---------------------------------------------------------------------------------------------------
// scans (nerfacc.cpp:17-60)
// ---------------------------------------------------------------------------------------------------
Tensor scan_packed(const Tensor &chunk_starts, const Tensor &chunk_cnts, const Tensor &inputs, int op, bool inclusive, bool reverse,
                   bool normalize) {
    check_input(chunk_starts, "chunk_starts", at::kLong);
    check_input(chunk_cnts, "chunk_cnts", at::kLong);
    check_input(inputs, "inputs", at::kFloat);
    TORCH_CHECK(chunk_starts.dim() == 1 && chunk_cnts.dim() == 1 && inputs.dim() == 1, "chunk_starts, chunk_cnts and inputs must be 1-D");
    TORCH_CHECK(chunk_starts.size(0) == chunk_cnts.size(0), "chunk_starts and chunk_cnts differ in length");
    Tensor out = at::empty_like(inputs);
    Guard g(device_of(inputs));
    check_rc(nfa_scan_packed(ptr<int64_t>(chunk_starts), ptr<int64_t>(chunk_cnts), chunk_cnts.size(0), ptr<float>(inputs), ptr<float>(out),
                             inputs.size(0), op, inclusive, reverse, normalize, stream_of(inputs)));
    return out;
}

Tensor scan_keyed(const Tensor &indices, const Tensor &inputs, int op, bool inclusive, bool reverse) {
    check_input(indices, "indices", at::kLong);
    check_input(inputs, "inputs", at::kFloat);
    TORCH_CHECK(indices.dim() == 1 && inputs.dim() == 1 && indices.size(0) == inputs.size(0), "indices and inputs must be 1-D with the same length");
    TORCH_CHECK(indices.device() == inputs.device(), "indices and inputs must live on the same device");
    Tensor out = at::empty_like(inputs);
    Guard g(device_of(inputs));
    check_rc(nfa_scan_keyed(ptr<int64_t>(indices), ptr<float>(inputs), ptr<float>(out), inputs.size(0), op, inclusive, reverse, stream_of(inputs)));
    return out;
}

Tensor prod_bwd(const OptTensor &indices, const OptTensor &chunk_starts, const OptTensor &chunk_cnts, const Tensor &inputs,
                const Tensor &outputs, const Tensor &grad_outputs, bool inclusive) {
    check_input(inputs, "inputs", at::kFloat);
    const int64_t n = inputs.numel();
    check_len(indices, "indices", at::kLong, n, inputs);
    if (chunk_starts) check_input(*chunk_starts, "chunk_starts", at::kLong);
    if (chunk_cnts) check_input(*chunk_cnts, "chunk_cnts", at::kLong);
    TORCH_CHECK(chunk_starts.has_value() == chunk_cnts.has_value() && (!chunk_starts || chunk_starts->numel() == chunk_cnts->numel()),
                "chunk_starts and chunk_cnts must be given together with the same length");
    check_len(outputs, "outputs", at::kFloat, n, inputs);
    check_len(grad_outputs, "grad_outputs", at::kFloat, n, inputs);
    Tensor gin = at::empty_like(grad_outputs);
    Guard g(device_of(inputs));
    check_rc(nfa_prod_backward(ptr<int64_t>(indices), ptr<int64_t>(chunk_starts), ptr<int64_t>(chunk_cnts), chunk_cnts ? chunk_cnts->size(0) : 0,
                               ptr<float>(inputs), ptr<float>(outputs), ptr<float>(grad_outputs), ptr<float>(gin), inputs.size(0), inclusive,
                               stream_of(inputs)));
    return gin;
}

// ---------------------------------------------------------------------------------------------------
// pdf (nerfacc.cpp:100-117)
// ---------------------------------------------------------------------------------------------------
std::vector<RaySegmentsSpec> importance_sampling(const RaySegmentsSpec &seg, const Tensor &cdfs, const py::object &n_intervals, bool stratified) {
    seg.check();
    check_input(cdfs, "cdfs", at::kFloat);
    TORCH_CHECK(cdfs.numel() == seg.vals->numel(), "cdfs and ray_segments.vals must have the same number of elements");
    if (!py::isinstance<py::int_>(n_intervals)) {
        // per-ray counts (nerfacc.cpp:100-105): flattened outputs sized from the counts — one host read of their sum, as the
        // reference's memalloc_data_from_chunk does (data_spec.hpp:86-97)
        Tensor cnts = n_intervals.cast<Tensor>();
        TORCH_CHECK(cnts.is_cuda() && cnts.device() == cdfs.device(), "n_intervals_per_ray must live on the device of cdfs");
        nfa_ray_segments view = seg.view();
        TORCH_CHECK(at::isIntegralType(cnts.scalar_type(), /*includeBool=*/false), "n_intervals_per_ray must be an integer tensor, got ",
                    cnts.scalar_type(), " (a float count would be truncated silently)");
        cnts = cnts.to(at::kLong).reshape({-1}).contiguous();
        TORCH_CHECK(cnts.numel() == view.n_rays, "n_intervals_per_ray must hold one count per ray (", view.n_rays, "), got ", cnts.numel());
        RaySegmentsSpec samples, intervals;
        samples.chunk_cnts = cnts;
        Tensor cs = at::cumsum(cnts, 0);
        samples.chunk_starts = cs - cnts;
        intervals.chunk_cnts = (cnts + 1) * (cnts > 0).to(at::kLong);
        Tensor ics = at::cumsum(*intervals.chunk_cnts, 0);
        intervals.chunk_starts = ics - *intervals.chunk_cnts;
        // ONE read-back: the smallest count and the two totals together (three blocking .item() calls before)
        int64_t n_s = 0, n_e = 0;
        if (cnts.numel()) {
            const Tensor three = at::stack({cnts.min(), cs[-1], ics[-1]}).cpu();
            const int64_t *h3 = three.data_ptr<int64_t>();
            TORCH_CHECK(h3[0] >= 0, "n_intervals_per_ray must not be negative");
            n_s = h3[1];
            n_e = h3[2];
        }
        const auto i64 = opts(cdfs, at::kLong), b8 = opts(cdfs, at::kBool);
        samples.vals = at::empty({n_s}, cdfs.options());
        samples.ray_indices = at::empty({n_s}, i64);
        intervals.vals = at::empty({n_e}, cdfs.options());
        intervals.ray_indices = at::empty({n_e}, i64);
        intervals.is_left = at::empty({n_e}, b8);
        intervals.is_right = at::empty({n_e}, b8);
        Tensor jitter;
        if (stratified) jitter = at::rand({view.n_rays}, cdfs.options());
        Guard g(device_of(cdfs));
        check_rc(nfa_importance_sampling_ragged(&view, ptr<float>(cdfs), ptr<int64_t>(samples.chunk_starts), ptr<int64_t>(cnts),
                                                ptr<int64_t>(intervals.chunk_starts), n_s, ptr<float>(jitter), ptr<float>(samples.vals),
                                                ptr<int64_t>(samples.ray_indices), ptr<float>(intervals.vals), ptr<int64_t>(intervals.ray_indices),
                                                (uint8_t *)intervals.is_left->data_ptr(), (uint8_t *)intervals.is_right->data_ptr(), stream_of(cdfs)));
        return {intervals, samples};
    }
    const int64_t n = n_intervals.cast<int64_t>();
    nfa_ray_segments view = seg.view();
    std::vector<int64_t> lead;
    if (seg.vals->dim() > 1) lead.assign(seg.vals->sizes().begin(), seg.vals->sizes().end() - 1);
    else lead = {view.n_rays};
    auto with_last = [&](int64_t k) { auto v = lead; v.push_back(k); return v; };
    RaySegmentsSpec samples, intervals;
    samples.vals = at::empty(with_last(n), cdfs.options());
    intervals.vals = at::empty(with_last(n + 1), cdfs.options());
    Tensor jitter;
    // one uniform per ray from torch's generator (the reference draws it with Philox inside the kernel, pdf.cu:138-144)
    if (stratified) jitter = at::rand({view.n_rays}, cdfs.options());
    Guard g(device_of(cdfs));
    check_rc(nfa_importance_sampling(&view, ptr<float>(cdfs), n, ptr<float>(jitter), ptr<float>(intervals.vals), ptr<float>(samples.vals),
                                     stream_of(cdfs)));
    return {intervals, samples};
}

std::vector<Tensor> searchsorted(const RaySegmentsSpec &query, const RaySegmentsSpec &key) {
    query.check();
    key.check();
    Tensor l = at::empty(query.vals->sizes(), opts(*query.vals, at::kLong)), r = at::empty_like(l);
    nfa_ray_segments q = query.view(), k = key.view();
    Guard g(device_of(*query.vals));
    check_rc(nfa_searchsorted(&q, &k, ptr<int64_t>(l), ptr<int64_t>(r), stream_of(*query.vals)));
    return {l, r};
}

Tensor transform_stot(const Tensor &s_vals, double t_min, double t_max, bool lindisp) {
    check_input(s_vals, "s_vals", at::kFloat);
    Tensor t = at::empty_like(s_vals);
    Guard g(device_of(s_vals));
    // (the reference's scalars are Python floats: 1 / t is formed in double inside the library call)
    check_rc(nfa_transform_stot(ptr<float>(s_vals), s_vals.numel(), t_min, t_max, lindisp ? 1 : 0, ptr<float>(t), stream_of(s_vals)));
    return t;
}

// (cdfs [R, S + 1], trans [R, S] or undefined)
std::vector<Tensor> edge_cdfs_fwd(const Tensor &t_edges, const Tensor &sigmas, bool want_trans) {
    check_input(t_edges, "t_edges", at::kFloat);
    check_input(sigmas, "sigmas", at::kFloat);
    TORCH_CHECK(sigmas.dim() == 2 && t_edges.dim() == 2 && t_edges.size(0) == sigmas.size(0) && t_edges.size(1) == sigmas.size(1) + 1 &&
                sigmas.size(1) >= 1 && t_edges.device() == sigmas.device(), "edge_cdfs: t_edges must be [n_rays, n + 1] and sigmas [n_rays, n]");
    Tensor cdfs = at::empty_like(t_edges), trans;
    if (want_trans) trans = at::empty_like(sigmas);
    Guard g(device_of(sigmas));
    check_rc(nfa_edge_cdfs_fwd(ptr<float>(t_edges), ptr<float>(sigmas), sigmas.size(0), sigmas.size(1), ptr<float>(cdfs), ptr<float>(trans),
                               stream_of(sigmas)));
    return {cdfs, trans};
}

Tensor edge_cdfs_bwd(const Tensor &t_edges, const Tensor &trans, const Tensor &g_cdfs) {
    check_input(t_edges, "t_edges", at::kFloat);
    check_input(trans, "trans", at::kFloat);
    check_input(g_cdfs, "g_cdfs", at::kFloat);
    TORCH_CHECK(trans.dim() == 2 && t_edges.dim() == 2 && t_edges.size(0) == trans.size(0) && t_edges.size(1) == trans.size(1) + 1 &&
                g_cdfs.sizes() == t_edges.sizes(), "edge_cdfs_bwd: shape mismatch");
    Tensor g_sig = at::empty_like(trans);
    Guard g(device_of(trans));
    check_rc(nfa_edge_cdfs_bwd(ptr<float>(t_edges), ptr<float>(trans), ptr<float>(g_cdfs), trans.size(0), trans.size(1), ptr<float>(g_sig),
                               stream_of(trans)));
    return g_sig;
}

// (loss [R, Nq], ids_left, ids_right (int32), coef) — the last three undefined unless want_grad
std::vector<Tensor> pdf_loss_fwd(const Tensor &q, const Tensor &cq, const Tensor &k, const Tensor &ck, double eps, bool want_grad) {
    check_input(q, "query vals", at::kFloat);
    check_input(cq, "cdfs_query", at::kFloat);
    check_input(k, "key vals", at::kFloat);
    check_input(ck, "cdfs_key", at::kFloat);
    TORCH_CHECK(q.dim() == 2 && k.dim() == 2 && cq.sizes() == q.sizes() && ck.sizes() == k.sizes() && q.size(0) == k.size(0) && q.size(1) >= 2 &&
                k.size(1) >= 2, "pdf_loss: batched [n_rays, n + 1] edges and cdfs expected");
    const int64_t R = q.size(0), nq = q.size(1) - 1, nk = k.size(1) - 1;
    Tensor loss = at::empty({R, nq}, q.options()), il, ir, coef;
    if (want_grad) { il = at::empty({R, nq}, opts(q, at::kInt)); ir = at::empty_like(il); coef = at::empty_like(loss); }
    Guard g(device_of(q));
    check_rc(nfa_pdf_loss_fwd(ptr<float>(q), ptr<float>(cq), ptr<float>(k), ptr<float>(ck), R, nq, nk, (float)eps, ptr<float>(loss),
                              ptr<int32_t>(il), ptr<int32_t>(ir), ptr<float>(coef), stream_of(q)));
    return {loss, il, ir, coef};
}

Tensor pdf_loss_bwd(const Tensor &g_loss, const Tensor &il, const Tensor &ir, const Tensor &coef, int64_t n_key) {
    check_input(g_loss, "g_loss", at::kFloat);
    check_input(il, "ids_left", at::kInt);
    check_input(ir, "ids_right", at::kInt);
    check_input(coef, "coef", at::kFloat);
    TORCH_CHECK(g_loss.dim() == 2 && il.sizes() == g_loss.sizes() && ir.sizes() == g_loss.sizes() && coef.sizes() == g_loss.sizes() && n_key >= 1,
                "pdf_loss_bwd: shape mismatch");
    Tensor g_ck = at::empty({g_loss.size(0), n_key + 1}, g_loss.options());
    Guard g(device_of(g_loss));
    check_rc(nfa_pdf_loss_bwd(ptr<float>(g_loss), ptr<int32_t>(il), ptr<int32_t>(ir), ptr<float>(coef), g_loss.size(0), g_loss.size(1), n_key,
                              ptr<float>(g_ck), stream_of(g_loss)));
    return g_ck;
}

// ---------------------------------------------------------------------------------------------------
// pack / rendering (fused entry points: chains of ATen ops in the reference)
// ---------------------------------------------------------------------------------------------------
Tensor pack_info(const Tensor &ray_indices, int64_t n_rays) {
    check_input(ray_indices, "ray_indices", at::kLong);
    TORCH_CHECK(n_rays >= 0, "n_rays must be >= 0");
    Tensor out = at::empty({n_rays, 2}, ray_indices.options());
    Guard g(device_of(ray_indices));
    check_rc(nfa_pack_info(ptr<int64_t>(ray_indices), ray_indices.size(0), n_rays, ptr<int64_t>(out), stream_of(ray_indices)));
    return out;
}

Tensor unpack_info(const Tensor &chunk_starts, const Tensor &chunk_cnts, int64_t n) {
    check_input(chunk_starts, "chunk_starts", at::kLong);
    check_input(chunk_cnts, "chunk_cnts", at::kLong);
    TORCH_CHECK(chunk_starts.numel() == chunk_cnts.numel() && n >= 0, "chunk_starts and chunk_cnts differ in length (or n < 0)");
    Tensor out = at::empty({n}, chunk_starts.options());
    Guard g(device_of(chunk_starts));
    check_rc(nfa_unpack_info(ptr<int64_t>(chunk_starts), ptr<int64_t>(chunk_cnts), chunk_cnts.size(0), ptr<int64_t>(out), n, stream_of(chunk_starts)));
    return out;
}

py::tuple render_weight_from_density_fwd(const Tensor &ray_indices, const Tensor &t_starts, const Tensor &t_ends, const Tensor &sigmas,
                                         const OptTensor &prefix_trans) {
    check_input(sigmas, "sigmas", at::kFloat);
    const int64_t n = sigmas.numel();
    check_len(ray_indices, "ray_indices", at::kLong, n, sigmas);
    check_len(t_starts, "t_starts", at::kFloat, n, sigmas);
    check_len(t_ends, "t_ends", at::kFloat, n, sigmas);
    check_len(prefix_trans, "prefix_trans", at::kFloat, n, sigmas);
    Rows out(3, n, sigmas.options());
    Guard g(device_of(sigmas));
    check_rc(nfa_render_weight_from_density_fwd(ptr<int64_t>(ray_indices), ptr<float>(t_starts), ptr<float>(t_ends), ptr<float>(sigmas),
                                                ptr<float>(prefix_trans), n, out.p(0), out.p(1), out.p(2), stream_of(sigmas)));
    return py::make_tuple(out.row(0), out.row(1), out.row(2));
}

Tensor render_weight_from_density_bwd(const Tensor &ray_indices, const Tensor &t_starts, const Tensor &t_ends, const Tensor &sigmas,
                                      const Tensor &trans, const Tensor &alphas, const OptTensor &g_w, const OptTensor &g_T, const OptTensor &g_a) {
    check_input(sigmas, "sigmas", at::kFloat);
    const int64_t n = sigmas.numel();
    check_len(ray_indices, "ray_indices", at::kLong, n, sigmas);
    check_len(t_starts, "t_starts", at::kFloat, n, sigmas);
    check_len(t_ends, "t_ends", at::kFloat, n, sigmas);
    check_len(trans, "trans", at::kFloat, n, sigmas);
    check_len(alphas, "alphas", at::kFloat, n, sigmas);
    for (auto *t : {&g_w, &g_T, &g_a}) check_len(*t, "grad", at::kFloat, n, sigmas);
    Tensor gs = at::empty_like(sigmas);
    Guard g(device_of(sigmas));
    check_rc(nfa_render_weight_from_density_bwd(ptr<int64_t>(ray_indices), ptr<float>(t_starts), ptr<float>(t_ends), ptr<float>(sigmas),
                                                ptr<float>(trans), ptr<float>(alphas), ptr<float>(g_w), ptr<float>(g_T), ptr<float>(g_a),
                                                sigmas.size(0), ptr<float>(gs), stream_of(sigmas)));
    return gs;
}

py::object sample_positions(const Tensor &rays_o, const Tensor &rays_d, const Tensor &ray_indices, const Tensor &t_starts, const Tensor &t_ends,
                            bool with_dirs) {
    check_input(rays_o, "rays_o", at::kFloat);
    check_input(rays_d, "rays_d", at::kFloat);
    check_input(ray_indices, "ray_indices", at::kLong);
    check_input(t_starts, "t_starts", at::kFloat);
    check_input(t_ends, "t_ends", at::kFloat);
    const int64_t n = ray_indices.size(0);
    TORCH_CHECK(t_starts.size(0) == n && t_ends.size(0) == n && rays_o.sizes() == rays_d.sizes() && rays_o.dim() == 2 && rays_o.size(1) == 3,
                "sample_positions: rays [R,3] x2 and ray_indices / t_starts / t_ends [N] expected");
    Tensor pos = at::empty({n, 3}, rays_o.options()), dirs;
    if (with_dirs) dirs = at::empty({n, 3}, rays_o.options());
    Guard g(device_of(rays_o));
    check_rc(nfa_sample_positions(ptr<float>(rays_o), ptr<float>(rays_d), rays_o.size(0), ptr<int64_t>(ray_indices), ptr<float>(t_starts),
                                  ptr<float>(t_ends), n, ptr<float>(pos), ptr<float>(dirs), stream_of(rays_o)));
    if (with_dirs) return py::make_tuple(pos, dirs);
    return py::cast(pos);
}

// (ray_indices', t_starts', t_ends', mask or None); ONE host sync (the count)
py::tuple visibility_compact(const Tensor &ray_indices, const Tensor &t_starts, const Tensor &t_ends, const Tensor &dens, bool from_alpha,
                             double early_stop_eps, double alpha_thre, bool want_mask) {
    check_input(dens, "sigmas/alphas", at::kFloat);
    const int64_t n = dens.numel();
    check_len(ray_indices, "ray_indices", at::kLong, n, dens);
    check_len(t_starts, "t_starts", at::kFloat, n, dens);
    check_len(t_ends, "t_ends", at::kFloat, n, dens);
    Tensor o_idx = at::empty({n}, ray_indices.options());
    Rows o_t(2, n, dens.options());
    Tensor mask;
    if (want_mask) mask = at::empty({n}, opts(dens, at::kBool));
    Tensor ws = at::empty({std::max<int64_t>(nfa_visibility_workspace_bytes(n), 16)}, opts(dens, at::kByte));
    Guard g(device_of(dens));
    hipStream_t s = stream_of(dens);
    int64_t *h = host_ints(dens.device().index(), s);
    const int64_t stamp = next_stamp();
    {
        Timed t("visibility", s);
        h[1] = 0;
        // (one launch at the training size: mask pass, look-back over its workgroups and compaction — include/nerfacc_hip.h)
        check_rc(nfa_visibility_compact_sync(ptr<int64_t>(ray_indices), ptr<float>(t_starts), ptr<float>(t_ends), ptr<float>(dens), from_alpha,
                                             n, (float)early_stop_eps, (float)alpha_thre, ptr<int64_t>(o_idx), o_t.p(0), o_t.p(1),
                                             ptr<uint8_t>(mask), h, stamp, ws.data_ptr(), n > 0 ? sync_block(dens, s) : nullptr, s));
    }
    if (n > 0) wait_stamp(h + 1, stamp, s);
    else wait_stream(s);
    if (n > 0 && h[0] < 0) {
        // the single launch's look-back gave up (a bounded wait): the compaction kernel over what it left in the workspace
        const int64_t stamp2 = next_stamp();
        h[1] = 0;
        check_rc(nfa_visibility_compact_resume(ptr<int64_t>(ray_indices), ptr<float>(t_starts), ptr<float>(t_ends), ptr<float>(dens), from_alpha,
                                               n, (float)early_stop_eps, (float)alpha_thre, ptr<int64_t>(o_idx), o_t.p(0), o_t.p(1),
                                               ptr<uint8_t>(mask), h, stamp2, ws.data_ptr(), s));
        wait_stamp(h + 1, stamp2, s);
    }
    const int64_t k = n > 0 ? h[0] : 0;
    py::object m = want_mask ? py::cast(mask) : py::none();
    return py::make_tuple(o_idx.narrow(0, 0, k), o_t.row(0).narrow(0, 0, k), o_t.row(1).narrow(0, 0, k), m);
}

Tensor accumulate_along_rays(const Tensor &ray_indices, const Tensor &weights, const OptTensor &values, int64_t n_rays, const OptTensor &outputs) {
    check_input(weights, "weights", at::kFloat);
    const int64_t n = weights.numel();
    check_len(ray_indices, "ray_indices", at::kLong, n, weights);
    int64_t D = 1;
    if (values) {
        TORCH_CHECK(values->dim() >= 1, "values must be [n_samples, D]");
        D = values->size(-1);
        check_len(*values, "values", at::kFloat, n, weights, D);
    }
    Tensor out;
    if (outputs) {
        check_input(*outputs, "outputs", at::kFloat);
        TORCH_CHECK(outputs->dim() == 2 && outputs->size(1) == D && outputs->device() == weights.device(), "outputs must be [n_rays, ", D, "] on the inputs' device");
        out = *outputs;
    } else {
        TORCH_CHECK(n_rays >= 0, "n_rays must be >= 0");
        out = at::zeros({n_rays, D}, weights.options());
    }
    Guard g(device_of(weights));
    check_rc(nfa_accumulate_along_rays(ptr<int64_t>(ray_indices), ptr<float>(weights), ptr<float>(values), weights.size(0), (int32_t)D,
                                       out.size(0), ptr<float>(out), stream_of(weights)));
    return out;
}

py::tuple accumulate_along_rays_bwd(const Tensor &ray_indices, const Tensor &weights, const OptTensor &values, const Tensor &g_out, bool need_w,
                                    bool need_v) {
    check_input(g_out, "g_outputs", at::kFloat);
    check_input(weights, "weights", at::kFloat);
    TORCH_CHECK(g_out.dim() == 2, "g_outputs must be [n_rays, D]");
    const int64_t D = g_out.size(-1), n = weights.numel();
    check_len(ray_indices, "ray_indices", at::kLong, n, weights);
    check_len(values, "values", at::kFloat, n, weights, D);
    TORCH_CHECK(values.has_value() || D == 1, "g_outputs must be [n_rays, 1] when values is None");
    TORCH_CHECK(g_out.device() == weights.device(), "g_outputs must live on the inputs' device");
    Tensor g_w, g_v;
    if (need_w) g_w = at::empty_like(weights);
    if (need_v && values) g_v = at::empty_like(*values);
    Guard g(device_of(weights));
    check_rc(nfa_accumulate_along_rays_bwd(ptr<int64_t>(ray_indices), ptr<float>(weights), ptr<float>(values), ptr<float>(g_out), weights.size(0),
                                           (int32_t)D, g_out.size(0), ptr<float>(g_w), ptr<float>(g_v), stream_of(weights)));
    return py::make_tuple(g_w.defined() ? py::cast(g_w) : py::none(), g_v.defined() ? py::cast(g_v) : py::none());
}

std::vector<Tensor> rendering_fwd_core(const Tensor &ray_indices, const Tensor &t_starts, const Tensor &t_ends, const Tensor &sigmas,
                                       const Tensor &rgbs, int64_t n_rays, const OptTensor &bkgd, bool expected_depths) {
    check_input(sigmas, "sigmas", at::kFloat);
    const int64_t n = sigmas.numel();
    check_len(ray_indices, "ray_indices", at::kLong, n, sigmas);
    check_len(t_starts, "t_starts", at::kFloat, n, sigmas);
    check_len(t_ends, "t_ends", at::kFloat, n, sigmas);
    check_len(rgbs, "rgbs", at::kFloat, n, sigmas, 3);
    check_len(bkgd, "render_bkgd", at::kFloat, 1, sigmas, 3);
    TORCH_CHECK(n_rays >= 0, "n_rays must be >= 0");
    Rows per(3, n, sigmas.options());
    Tensor colors = at::empty({n_rays, 3}, sigmas.options());
    Tensor od = at::empty({2, n_rays, 1}, sigmas.options());
    Guard g(device_of(sigmas));
    hipStream_t s = stream_of(sigmas);
    {
        Timed t("rendering_fwd", s);
        check_rc(nfa_rendering_fwd(ptr<int64_t>(ray_indices), ptr<float>(t_starts), ptr<float>(t_ends), ptr<float>(sigmas), ptr<float>(rgbs), n,
                                   n_rays, ptr<float>(bkgd), expected_depths, per.p(0), per.p(1), per.p(2),
                                   ptr<float>(colors), ptr<float>(od), ptr<float>(od) + n_rays, s));
    }
    return {colors, od[0], od[1], per.row(0), per.row(1), per.row(2)};
}
py::tuple rendering_fwd(const Tensor &ray_indices, const Tensor &t_starts, const Tensor &t_ends, const Tensor &sigmas, const Tensor &rgbs,
                        int64_t n_rays, const OptTensor &bkgd, bool expected_depths) {
    const auto o = rendering_fwd_core(ray_indices, t_starts, t_ends, sigmas, rgbs, n_rays, bkgd, expected_depths);
    return py::make_tuple(o[0], o[1], o[2], o[3], o[4], o[5]);
}

std::pair<Tensor, Tensor> rendering_bwd_core(const Tensor &ray_indices, const Tensor &t_starts, const Tensor &t_ends, const Tensor &sigmas,
                                             const Tensor &rgbs, const Tensor &weights, const Tensor &trans, const Tensor &alphas,
                                             const Tensor &opacities, const Tensor &depths, int64_t n_rays, const OptTensor &bkgd,
                                             bool expected_depths, const OptTensor &g_colors, const OptTensor &g_opac, const OptTensor &g_depth,
                                             const OptTensor &g_w, const OptTensor &g_T, const OptTensor &g_a, bool need_sigma, bool need_rgb) {
    check_input(sigmas, "sigmas", at::kFloat);
    const int64_t n = sigmas.numel();
    check_len(ray_indices, "ray_indices", at::kLong, n, sigmas);
    check_len(t_starts, "t_starts", at::kFloat, n, sigmas);
    check_len(t_ends, "t_ends", at::kFloat, n, sigmas);
    check_len(rgbs, "rgbs", at::kFloat, n, sigmas, 3);
    check_len(weights, "weights", at::kFloat, n, sigmas);
    check_len(trans, "trans", at::kFloat, n, sigmas);
    check_len(alphas, "alphas", at::kFloat, n, sigmas);
    check_len(opacities, "opacities", at::kFloat, n_rays, sigmas);
    check_len(depths, "depths", at::kFloat, n_rays, sigmas);
    check_len(bkgd, "render_bkgd", at::kFloat, 1, sigmas, 3);
    check_len(g_colors, "g_colors", at::kFloat, n_rays, sigmas, 3);
    check_len(g_opac, "g_opacities", at::kFloat, n_rays, sigmas);
    check_len(g_depth, "g_depths", at::kFloat, n_rays, sigmas);
    for (auto *t : {&g_w, &g_T, &g_a}) check_len(*t, "grad", at::kFloat, n, sigmas);
    Tensor g_sig, g_rgb;
    if (need_sigma) g_sig = at::empty_like(sigmas);
    if (need_rgb) g_rgb = at::empty_like(rgbs);
    Guard g(device_of(sigmas));
    hipStream_t s = stream_of(sigmas);
    {
        Timed t("rendering_bwd", s);
        check_rc(nfa_rendering_bwd(ptr<int64_t>(ray_indices), ptr<float>(t_starts), ptr<float>(t_ends), ptr<float>(sigmas), ptr<float>(rgbs),
                                   ptr<float>(weights), ptr<float>(trans), ptr<float>(alphas), ptr<float>(opacities), ptr<float>(depths),
                                   sigmas.size(0), n_rays, ptr<float>(bkgd), expected_depths, ptr<float>(g_colors), ptr<float>(g_opac),
                                   ptr<float>(g_depth), ptr<float>(g_w), ptr<float>(g_T), ptr<float>(g_a), ptr<float>(g_sig), ptr<float>(g_rgb), s));
    }
    return {g_sig, g_rgb};
}
py::tuple rendering_bwd(const Tensor &ray_indices, const Tensor &t_starts, const Tensor &t_ends, const Tensor &sigmas, const Tensor &rgbs,
                        const Tensor &weights, const Tensor &trans, const Tensor &alphas, const Tensor &opacities, const Tensor &depths,
                        int64_t n_rays, const OptTensor &bkgd, bool expected_depths, const OptTensor &g_colors, const OptTensor &g_opac,
                        const OptTensor &g_depth, const OptTensor &g_w, const OptTensor &g_T, const OptTensor &g_a, bool need_sigma, bool need_rgb) {
    const auto r = rendering_bwd_core(ray_indices, t_starts, t_ends, sigmas, rgbs, weights, trans, alphas, opacities, depths, n_rays, bkgd,
                                      expected_depths, g_colors, g_opac, g_depth, g_w, g_T, g_a, need_sigma, need_rgb);
    return py::make_tuple(r.first.defined() ? py::cast(r.first) : py::none(), r.second.defined() ? py::cast(r.second) : py::none());
}

// ---------------------------------------------------------------------------------------------------
// rendering with its autograd node in C++ (volrend.py:104-164 after rgb_sigma_fn): what nerfacc.rendering calls on the
// training path.  The Python twin (nerfacc_amd/volrend.py::_Rendering) costs ~15 us of interpreter per forward and ~30 us
// per backward on a step whose GPU waits for the host there (tools/step_timeline.py).
// ---------------------------------------------------------------------------------------------------
struct RenderingFn : public torch::autograd::Function<RenderingFn> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext *ctx, Tensor ray_indices, Tensor t_starts, Tensor t_ends,
                                                  Tensor sigmas, Tensor rgbs, int64_t n_rays, OptTensor bkgd, bool expected_depths) {
        ctx->set_materialize_grads(false);      // of six outputs a loss usually touches one: no zero fills for the rest
        ray_indices = ray_indices.contiguous();
        t_starts = t_starts.contiguous();
        t_ends = t_ends.contiguous();
        sigmas = sigmas.contiguous();
        rgbs = rgbs.contiguous();
        OptTensor bk;
        if (bkgd && bkgd->defined()) bk = bkgd->detach().to(at::kFloat).contiguous();
        std::vector<Tensor> o = rendering_fwd_core(ray_indices, t_starts, t_ends, sigmas, rgbs, n_rays, bk, expected_depths);
        ctx->saved_data["n_rays"] = n_rays;
        ctx->saved_data["expected_depths"] = expected_depths;
        ctx->saved_data["has_bkgd"] = bk.has_value();
        std::vector<Tensor> saved = {ray_indices, t_starts, t_ends, sigmas, rgbs, o[3], o[4], o[5], o[1], o[2]};
        if (bk) saved.push_back(*bk);
        ctx->save_for_backward(saved);
        return o;
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::variable_list g) {
        const auto saved = ctx->get_saved_variables();
        const bool has_bkgd = ctx->saved_data["has_bkgd"].toBool();
        auto opt = [](const Tensor &t) -> OptTensor { return t.defined() ? OptTensor(t.contiguous()) : std::nullopt; };
        const bool need_sigma = ctx->needs_input_grad(3), need_rgb = ctx->needs_input_grad(4);
        Tensor g_sig, g_rgb;
        if (need_sigma || need_rgb) {            // (no Python object anywhere below: the autograd engine's thread does not hold the GIL)
            auto r = rendering_bwd_core(saved[0], saved[1], saved[2], saved[3], saved[4], saved[5], saved[6], saved[7], saved[8], saved[9],
                                        ctx->saved_data["n_rays"].toInt(), has_bkgd ? OptTensor(saved[10]) : std::nullopt,
                                        ctx->saved_data["expected_depths"].toBool(), opt(g[0]), opt(g[1]), opt(g[2]), opt(g[3]), opt(g[4]),
                                        opt(g[5]), need_sigma, need_rgb);
            g_sig = r.first;
            g_rgb = r.second;
        }
        return {Tensor(), Tensor(), Tensor(), g_sig, g_rgb, Tensor(), Tensor(), Tensor()};
    }
};

py::tuple rendering_autograd(const Tensor &ray_indices, const Tensor &t_starts, const Tensor &t_ends, const Tensor &sigmas, const Tensor &rgbs,
                             int64_t n_rays, const OptTensor &bkgd, bool expected_depths) {
    auto o = RenderingFn::apply(ray_indices, t_starts, t_ends, sigmas, rgbs, n_rays, bkgd, expected_depths);
    return py::make_tuple(o[0], o[1], o[2], o[3], o[4], o[5]);
}

// ---------------------------------------------------------------------------------------------------
// occupancy-grid maintenance (OccGridEstimator._update, occ_grid.py:366-404)
// ---------------------------------------------------------------------------------------------------
Tensor grid_cell_points(const OptTensor &cell_ids, const Tensor &jitter, const std::vector<int64_t> &resolution, const Tensor &aabb) {
    check_input(jitter, "jitter", at::kFloat);
    check_input(aabb, "aabb", at::kFloat);
    const int64_t n = jitter.size(0);
    if (cell_ids) {
        check_input(*cell_ids, "cell_ids", at::kLong);
        TORCH_CHECK(cell_ids->dim() == 1 && cell_ids->size(0) == n, "cell_ids must have shape [n] matching jitter [n, 3]");
    }
    TORCH_CHECK(jitter.dim() == 2 && jitter.size(1) == 3 && aabb.numel() == 6 && resolution.size() == 3, "jitter must be [n, 3] and aabb must hold 6 floats");
    Tensor points = at::empty({n, 3}, jitter.options());
    Guard g(device_of(jitter));
    hipStream_t s = stream_of(jitter);
    Timed t("grid_cell_points", s);
    check_rc(nfa_grid_cell_points(ptr<int64_t>(cell_ids), n, ptr<float>(jitter), (int32_t)resolution[0], (int32_t)resolution[1],
                                  (int32_t)resolution[2], ptr<float>(aabb), ptr<float>(points), s));
    return points;
}

void grid_ema_update(const Tensor &occs_level, const OptTensor &cell_ids, const Tensor &occ_new, double ema_decay) {
    check_input(occs_level, "occs", at::kFloat);
    check_input(occ_new, "occ_new", at::kFloat);
    const int64_t n = occ_new.numel();
    if (cell_ids) {
        check_input(*cell_ids, "cell_ids", at::kLong);
        TORCH_CHECK(cell_ids->numel() == n, "cell_ids and occ_new must have the same number of elements");
    } else {
        TORCH_CHECK(n <= occs_level.numel(), "occ_new has more elements than the level has cells");
    }
    Tensor scratch = at::empty({n}, occs_level.options());
    Guard g(device_of(occs_level));
    hipStream_t s = stream_of(occs_level);
    Timed t("grid_ema_update", s);
    check_rc(nfa_grid_ema_update(ptr<float>(occs_level), ptr<int64_t>(cell_ids), n, ptr<float>(occ_new), (float)ema_decay, ptr<float>(scratch), s));
}

void grid_mark_invisible(const Tensor &occs_level, const OptTensor &cell_ids, const std::vector<int64_t> &resolution, const Tensor &aabb,
                         const Tensor &w2c_R, const Tensor &w2c_T, const Tensor &K, double width, double height, double near_plane) {
    check_input(occs_level, "occs", at::kFloat);
    check_input(aabb, "aabb", at::kFloat);
    check_input(w2c_R, "w2c_R", at::kFloat);
    check_input(w2c_T, "w2c_T", at::kFloat);
    check_input(K, "K", at::kFloat);
    int64_t n = occs_level.numel();
    if (cell_ids) { check_input(*cell_ids, "cell_ids", at::kLong); n = cell_ids->numel(); }
    const int64_t C = w2c_R.size(0);
    TORCH_CHECK(w2c_R.numel() == 9 * C && w2c_T.numel() == 3 * C && (K.numel() == 9 || K.numel() == 9 * C) && resolution.size() == 3,
                "grid_mark_invisible: w2c_R [C,3,3], w2c_T [C,3,1], K [C or 1,3,3] expected");
    Guard g(device_of(occs_level));
    check_rc(nfa_grid_mark_invisible(ptr<float>(occs_level), ptr<int64_t>(cell_ids), n, (int32_t)resolution[0], (int32_t)resolution[1],
                                     (int32_t)resolution[2], ptr<float>(aabb), ptr<float>(w2c_R), ptr<float>(w2c_T), ptr<float>(K), (int32_t)C,
                                     (K.numel() == 9 && C != 1) ? 1 : 0, (float)width, (float)height, (float)near_plane, stream_of(occs_level)));
}

// shape = None: flat bool grid.  shape = (G, rx, ry, rz): the bool grid in that shape AND its bit-packed form, produced by
// the same pass and entered into the brick cache — the traversal that follows does not pack again.
py::tuple grid_threshold(const Tensor &occs, double occ_thre, const std::optional<std::vector<int64_t>> &shape) {
    check_input(occs, "occs", at::kFloat);
    const int64_t n = occs.numel();
    Tensor ws = at::empty({nfa_grid_threshold_workspace_bytes() / 8}, opts(occs, at::kDouble));
    Tensor thre = at::empty({1}, occs.options());
    Guard g(device_of(occs));
    hipStream_t s = stream_of(occs);
    Timed t("grid_threshold", s);
    if (shape) {
        TORCH_CHECK(shape->size() == 4 && (*shape)[0] * (*shape)[1] * (*shape)[2] * (*shape)[3] == n && n > 0,
                    "grid_threshold: shape must be (n_grids, resx, resy, resz) with as many cells as occs");
        const int G = (int)(*shape)[0], rx = (int)(*shape)[1], ry = (int)(*shape)[2], rz = (int)(*shape)[3];
        Tensor binaries = at::empty(*shape, opts(occs, at::kBool));
        Tensor bricks = at::empty({nfa_packed_grid_words(G, rx, ry, rz)}, opts(occs, at::kLong));
        check_rc(nfa_grid_threshold_packed(ptr<float>(occs), G, rx, ry, rz, (float)occ_thre, ws.data_ptr(), ptr<uint8_t>(binaries),
                                           ptr<float>(thre), ptr<uint64_t>(bricks), s));
        std::lock_guard<std::mutex> l(g_brick_mu);
        insert_brick_entry(binaries, bricks, s);
        return py::make_tuple(binaries, thre);
    }
    Tensor binaries = at::empty({n}, opts(occs, at::kBool));
    check_rc(nfa_grid_threshold(ptr<float>(occs), n, (float)occ_thre, ws.data_ptr(), ptr<uint8_t>(binaries), ptr<float>(thre), s));
    return py::make_tuple(binaries, thre);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "nerfacc_amd._hip: torch C++ extension over the C ABI of libnerfacc_hip.so (MI355X / gfx950)";
    m.def("version", []() { return std::string(nfa_version()); });

    py::class_<RaySegmentsSpec>(m, "RaySegmentsSpec")
        .def(py::init<>())
        .def_readwrite("vals", &RaySegmentsSpec::vals)
        .def_readwrite("is_left", &RaySegmentsSpec::is_left)
        .def_readwrite("is_right", &RaySegmentsSpec::is_right)
        .def_readwrite("is_valid", &RaySegmentsSpec::is_valid)
        .def_readwrite("chunk_starts", &RaySegmentsSpec::chunk_starts)
        .def_readwrite("chunk_cnts", &RaySegmentsSpec::chunk_cnts)
        .def_readwrite("ray_indices", &RaySegmentsSpec::ray_indices)
        .def("check", &RaySegmentsSpec::check);

    using namespace pybind11::literals;
    // ---- the reference boundary (nerfacc.cpp:126-163), same names and argument order
    m.def("is_cub_available", []() { return true; });     // scan-by-key is native (scan.hip): the keyed path is always there
    m.def("ray_aabb_intersect", &ray_aabb_intersect);
    m.def("traverse_grids", &traverse_grids);
    m.def("inclusive_sum", [](const Tensor &s, const Tensor &c, const Tensor &x, bool normalize, bool backward) {
        return scan_packed(s, c, x, NFA_OP_SUM, true, backward, normalize); });
    m.def("exclusive_sum", [](const Tensor &s, const Tensor &c, const Tensor &x, bool normalize, bool backward) {
        return scan_packed(s, c, x, NFA_OP_SUM, false, backward, normalize); });
    m.def("inclusive_prod_forward", [](const Tensor &s, const Tensor &c, const Tensor &x) { return scan_packed(s, c, x, NFA_OP_PROD, true, false, false); });
    m.def("exclusive_prod_forward", [](const Tensor &s, const Tensor &c, const Tensor &x) { return scan_packed(s, c, x, NFA_OP_PROD, false, false, false); });
    m.def("inclusive_prod_backward", [](const Tensor &s, const Tensor &c, const Tensor &x, const Tensor &y, const Tensor &gy) {
        return prod_bwd(std::nullopt, s, c, x, y, gy, true); });
    m.def("exclusive_prod_backward", [](const Tensor &s, const Tensor &c, const Tensor &x, const Tensor &y, const Tensor &gy) {
        return prod_bwd(std::nullopt, s, c, x, y, gy, false); });
    m.def("inclusive_sum_cub", [](const Tensor &k, const Tensor &x, bool backward) { return scan_keyed(k, x, NFA_OP_SUM, true, backward); });
    m.def("exclusive_sum_cub", [](const Tensor &k, const Tensor &x, bool backward) { return scan_keyed(k, x, NFA_OP_SUM, false, backward); });
    m.def("inclusive_prod_cub_forward", [](const Tensor &k, const Tensor &x) { return scan_keyed(k, x, NFA_OP_PROD, true, false); });
    m.def("exclusive_prod_cub_forward", [](const Tensor &k, const Tensor &x) { return scan_keyed(k, x, NFA_OP_PROD, false, false); });
    m.def("inclusive_prod_cub_backward", [](const Tensor &k, const Tensor &x, const Tensor &y, const Tensor &gy) {
        return prod_bwd(k, std::nullopt, std::nullopt, x, y, gy, true); });
    m.def("exclusive_prod_cub_backward", [](const Tensor &k, const Tensor &x, const Tensor &y, const Tensor &gy) {
        return prod_bwd(k, std::nullopt, std::nullopt, x, y, gy, false); });
    // the generic forms behind the 14 scan names (op, inclusive, reverse[, normalize]); tests drive every combination
    m.def("_packed", &scan_packed);
    m.def("_keyed", &scan_keyed);
    m.def("importance_sampling", &importance_sampling);
    m.def("searchsorted", &searchsorted);
    m.def("transform_stot", &transform_stot, "s_vals"_a, "t_min"_a, "t_max"_a, "lindisp"_a);
    m.def("edge_cdfs_fwd", &edge_cdfs_fwd, "t_edges"_a, "sigmas"_a, "want_trans"_a);
    m.def("edge_cdfs_bwd", &edge_cdfs_bwd, "t_edges"_a, "trans"_a, "g_cdfs"_a);
    m.def("pdf_loss_fwd", &pdf_loss_fwd, "query_vals"_a, "cdfs_query"_a, "key_vals"_a, "cdfs_key"_a, "eps"_a, "want_grad"_a);
    m.def("pdf_loss_bwd", &pdf_loss_bwd, "g_loss"_a, "ids_left"_a, "ids_right"_a, "coef"_a, "n_key"_a);
    m.def("opencv_lens_undistortion", [](py::args, py::kwargs) {
        raise_not_implemented("camera undistortion (camera.cu) is outside the OccGrid hot path and not built"); });
    m.def("opencv_lens_undistortion_fisheye", [](py::args, py::kwargs) {
        raise_not_implemented("camera undistortion (camera.cu) is outside the OccGrid hot path and not built"); });

    // ---- fused entry points of this implementation
    m.def("release_workspace", &release_workspace,
          "free this host thread's retained traversal workspace (<= 64 MB per device and stream) and any count pass launched ahead");
    m.def("sample_occgrid", &sample_occgrid, "rays_o"_a, "rays_d"_a, "binaries"_a, "aabbs"_a, "near_planes"_a, "far_planes"_a, "step_size"_a,
          "cone_angle"_a, "rays_mask"_a = py::none(), "traverse_steps_limit"_a = -1, "with_terminate_planes"_a = false, "near_plane"_a = 0.0,
          "far_plane"_a = std::numeric_limits<double>::infinity(), "t_min"_a = py::none(), "t_max"_a = py::none(), "jitter"_a = py::none(),
          "jitter_scale"_a = 0.0);
    m.def("pack_info", &pack_info, "ray_indices"_a, "n_rays"_a);
    m.def("unpack_info", &unpack_info, "chunk_starts"_a, "chunk_cnts"_a, "n"_a);
    m.def("render_weight_from_density_fwd", &render_weight_from_density_fwd, "ray_indices"_a, "t_starts"_a, "t_ends"_a, "sigmas"_a,
          "prefix_trans"_a = py::none());
    m.def("render_weight_from_density_bwd", &render_weight_from_density_bwd);
    m.def("sample_positions", &sample_positions, "rays_o"_a, "rays_d"_a, "ray_indices"_a, "t_starts"_a, "t_ends"_a, "with_dirs"_a = false);
    m.def("visibility_compact", &visibility_compact, "ray_indices"_a, "t_starts"_a, "t_ends"_a, "dens"_a, "from_alpha"_a, "early_stop_eps"_a,
          "alpha_thre"_a, "want_mask"_a = false);
    m.def("accumulate_along_rays", &accumulate_along_rays, "ray_indices"_a, "weights"_a, "values"_a, "n_rays"_a, "outputs"_a = py::none());
    m.def("accumulate_along_rays_bwd", &accumulate_along_rays_bwd);
    m.def("rendering_fwd", &rendering_fwd);
    m.def("rendering_bwd", &rendering_bwd, "ray_indices"_a, "t_starts"_a, "t_ends"_a, "sigmas"_a, "rgbs"_a, "weights"_a, "trans"_a, "alphas"_a,
          "opacities"_a, "depths"_a, "n_rays"_a, "bkgd"_a, "expected_depths"_a, "g_colors"_a, "g_opac"_a, "g_depth"_a, "g_w"_a, "g_T"_a, "g_a"_a,
          "need_sigma"_a = true, "need_rgb"_a = true);
    m.def("rendering", &rendering_autograd, "ray_indices"_a, "t_starts"_a, "t_ends"_a, "sigmas"_a, "rgbs"_a, "n_rays"_a, "render_bkgd"_a,
          "expected_depths"_a);
    m.def("grid_cell_points", &grid_cell_points);
    m.def("grid_ema_update", &grid_ema_update);
    m.def("grid_threshold", &grid_threshold, "occs"_a, "occ_thre"_a, "shape"_a = py::none());
    m.def("grid_mark_invisible", &grid_mark_invisible);
    m.def("packed_bricks", &packed_bricks);
    m.def("grid_occupied_counts", &grid_occupied_counts);
    m.def("grid_occupied_cells", &grid_occupied_cells);
    m.def("set_timing", &set_timing, "names"_a = py::none());
    m.def("timing_summary", &timing_summary);
}
