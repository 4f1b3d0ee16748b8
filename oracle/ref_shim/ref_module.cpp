// pybind face of oracle/_ref: hands CPU tensors to the REFERENCE's own host wrappers
// (traverse_grids, grid.cu:320-474; ray_aabb_intersect, grid.cu:477-519), which were compiled
// from /root/reference where they lie (see Makefile).  TEST INFRASTRUCTURE: used only by
// tests/golden/make_k2_golden.py to generate fixtures and by tests that happen to run where
// /root/reference exists.  Nothing here restates the algorithm.
#include <torch/extension.h>

#include "include/data_spec.hpp"  // the reference's RaySegmentsSpec (found through -I <reference csrc>)

std::tuple<RaySegmentsSpec, RaySegmentsSpec, torch::Tensor> traverse_grids(
    const torch::Tensor rays_o, const torch::Tensor rays_d, const torch::Tensor rays_mask,
    const torch::Tensor binaries, const torch::Tensor aabbs, const torch::Tensor t_sorted,
    const torch::Tensor t_indices, const torch::Tensor hits, const torch::Tensor near_planes,
    const torch::Tensor far_planes, const float step_size, const float cone_angle,
    const bool compute_intervals, const bool compute_samples, const bool compute_terminate_planes,
    const int32_t traverse_steps_limit, const bool over_allocate);

std::vector<torch::Tensor> ray_aabb_intersect(const torch::Tensor rays_o, const torch::Tensor rays_d,
                                              const torch::Tensor aabbs, const float near_plane,
                                              const float far_plane, const float miss_value);

// pdf.cu:294-425 (Tensor-count and int-count overloads) and :428-456
std::vector<RaySegmentsSpec> importance_sampling(RaySegmentsSpec ray_segments, torch::Tensor cdfs, torch::Tensor n_intervels_per_ray, bool stratified);
std::vector<RaySegmentsSpec> importance_sampling(RaySegmentsSpec ray_segments, torch::Tensor cdfs, int64_t n_intervels_per_ray, bool stratified);
std::vector<torch::Tensor> searchsorted(RaySegmentsSpec query, RaySegmentsSpec key);

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    // same attribute names as the reference's own binding (nerfacc.cpp:126-137), so that the reference's
    // PYTHON layer (nerfacc/grid.py, data_specs.py, estimators/occ_grid.py) runs unmodified on top of it.
    py::class_<RaySegmentsSpec>(m, "RaySegmentsSpec", py::module_local())
        .def(py::init<>())
        .def_readwrite("vals", &RaySegmentsSpec::vals)
        .def_readwrite("chunk_starts", &RaySegmentsSpec::chunk_starts)
        .def_readwrite("chunk_cnts", &RaySegmentsSpec::chunk_cnts)
        .def_readwrite("ray_indices", &RaySegmentsSpec::ray_indices)
        .def_readwrite("is_left", &RaySegmentsSpec::is_left)
        .def_readwrite("is_right", &RaySegmentsSpec::is_right)
        .def_readwrite("is_valid", &RaySegmentsSpec::is_valid);
    m.def("traverse_grids", &traverse_grids);
    m.def("ray_aabb_intersect", &ray_aabb_intersect);
    m.def("is_cub_available", []() { return false; });
    // the two overloads as the reference binds them (nerfacc.cpp:152-159)
    m.def("importance_sampling", py::overload_cast<RaySegmentsSpec, torch::Tensor, torch::Tensor, bool>(&importance_sampling));
    m.def("importance_sampling", py::overload_cast<RaySegmentsSpec, torch::Tensor, int64_t, bool>(&importance_sampling));
    m.def("searchsorted", &searchsorted);
}
