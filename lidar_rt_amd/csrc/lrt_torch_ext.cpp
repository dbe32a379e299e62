// lrt_torch_ext.cpp -- the PyTorch-ROCm C++ extension `diff_lidar_tracer._C_ext`: counterpart of the reference's pybind module
// (DLT/ext.cpp:18-23, built by DLT/setup.py:25-74), on the C ABI of liblrt_hip.so (include/lrt.h).
//
//   OptiXStateWrapper(pkg_dir)                                     DLT/optix_tracer/optix_wrapper.cpp:177-233, ext.cpp:19
//   build_acceleration_structure(state, vertices, triangles, rebuild)          DLT/trace_surfels.cpp:46-148,   ext.cpp:20
//   trace_surfels(state, training, ray_o, ..., debug) -> (out_f32, out_i32, accum)     trace_surfels.cpp:152-265, ext.cpp:21
//   trace_surfels_backward(state, ray_o, ..., dL_dout) -> 8 tensors                    trace_surfels.cpp:269-386, ext.cpp:22
//
// Same names, argument orders, return arities and shape-error messages.  What the reference's host code does per call --
// `.contiguous()` on every input, output allocation, the 8 zero-filled gradient tensors, `at::cuda::getCurrentCUDAStream()`
// (trace_surfels.cpp:175-245, 294-370) -- happens here in C++ as well (current HIP stream, device guard), minus its per-call
// cudaMalloc of Params and its cudaStreamSynchronize: the calls only enqueue.  Host-only translation unit (no device code): the
// kernels live in liblrt_hip.so, which this module links.
#include <torch/extension.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/lrt.h"

namespace py = pybind11;

namespace {

struct LrtFailure : std::runtime_error { using std::runtime_error::runtime_error; };

void check_rc(int rc, const char* what)
{
    if (rc != 0) throw LrtFailure(std::string(what) + " failed (" + std::to_string(rc) + "): " + lrt_last_error());
}

struct State {                                   // the reference's OptiXStateWrapper: one tracer state per device, created lazily
    std::string pkg_dir;
    std::map<int, lrt_state*> handles;
    std::map<int, bool> dirty;
    std::map<int, bool> refit_next;              // build_acceleration_structure(rebuild = 0): the next build of an unchanged P is an lrt_refit (trace_surfels.cpp:63-73)
    std::map<int, int> built_P, since_full, full_P;
    std::map<int, float> built_mod;
    std::map<std::string, int> options;
    int refit_interval = 0;                      // > 0: that many lrt_refit calls between full builds of an unchanged number of Gaussians
    bool stats_enabled = false, timing_enabled = false;
    py::object last_serial = py::none();

    explicit State(std::string dir) : pkg_dir(std::move(dir)) {}
    ~State() { for (auto& kv : handles) lrt_destroy(kv.second); }

    lrt_state* handle(int idx)
    {
        auto it = handles.find(idx);
        if (it != handles.end()) return it->second;
        lrt_state* h = lrt_create(idx);
        if (!h) throw LrtFailure(std::string("lrt_create failed: ") + lrt_last_error());
        handles[idx] = h; dirty[idx] = true;
        if (stats_enabled) lrt_enable_stats(h, 1);
        if (timing_enabled) lrt_enable_timing(h, 1);
        for (auto& kv : options) check_rc(lrt_set_option(h, kv.first.c_str(), kv.second), "lrt_set_option");
        return h;
    }
    void set_option(const std::string& name, int value)
    {
        options[name] = value;
        for (auto& kv : handles) check_rc(lrt_set_option(kv.second, name.c_str(), value), "lrt_set_option");
    }
    void mark_dirty(bool refit = false) { for (auto& kv : dirty) { kv.second = true; refit_next[kv.first] = refit; } }
};

int device_index(const at::Tensor& t)
{
    if (!t.is_cuda()) throw std::runtime_error("diff_lidar_tracer: tensors must be on a HIP (cuda) device; there is no CPU path");
    return t.get_device();
}

void check_f32_cuda(const at::Tensor& t, const char* name)
{
    TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor");                        // AT_ASSERTM(is_cuda), trace_surfels.cpp:33-35
    TORCH_CHECK(t.numel() == 0 || t.scalar_type() == at::kFloat, name, " must be float32");
}

const float* fptr(const at::Tensor& t) { return t.numel() > 0 ? t.data_ptr<float>() : nullptr; }
float* fptr_mut(at::Tensor& t) { return t.numel() > 0 ? t.data_ptr<float>() : nullptr; }

bool present(const c10::optional<at::Tensor>& t) { return t.has_value() && t->defined() && t->numel() > 0; }

[[noreturn]] void not_implemented(const char* msg)
{
    PyErr_SetString(PyExc_NotImplementedError, msg);
    throw py::error_already_set();
}

void build_acceleration_structure(State& st, const at::Tensor& vertices, const at::Tensor& triangles, unsigned rebuild)
{
    // shape checks and messages of DLT/trace_surfels.cpp:53-58
    if (vertices.dim() != 2 || vertices.size(1) != 3) AT_ERROR("vertices must have dimensions (num_vertices, 3)");
    if (triangles.dim() != 2 || triangles.size(1) != 3) AT_ERROR("triangles must have dimensions (num_triangles, 3)");
    TORCH_CHECK(vertices.is_cuda() && triangles.is_cuda(), "vertices/triangles must be CUDA tensors");
    // rebuild == 0 is the reference's OPTIX_BUILD_OPERATION_UPDATE (trace_surfels.cpp:63-73: same topology, new vertex positions): the
    // structure is refitted (lrt_refit) when the next trace finds the number of Gaussians of the last full build, rebuilt otherwise
    st.mark_dirty(rebuild == 0u);
    st.handle(device_index(vertices));            // create the per-device state eagerly (errors surface here)
}

// Fused fast path: LBVH straight from the Gaussian parameters (no vertices tensor).  cull_rays = (ray_o, ray_d): build only what
// these rays can reach (lrt_build_for_rays; the azimuth-sharded tracer).
void build_from_gaussians(State& st, const at::Tensor& means3D, const at::Tensor& scales, const at::Tensor& rotations,
                          const at::Tensor& opacities, double scale_modifier, const c10::optional<std::tuple<at::Tensor, at::Tensor>>& cull_rays)
{
    check_f32_cuda(means3D, "means3D"); check_f32_cuda(scales, "scales"); check_f32_cuda(rotations, "rotations"); check_f32_cuda(opacities, "opacities");
    if (means3D.dim() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");
    const int64_t P = means3D.size(0);
    TORCH_CHECK(scales.numel() == 2 * P && rotations.numel() == 4 * P && opacities.numel() == P,
                "scales (P,2), rotations (P,4), opacities (P,1) must match means3D (P,3)");
    const int idx = device_index(means3D);
    lrt_state* h = st.handle(idx);
    const at::Tensor m = means3D.detach().contiguous(), s = scales.detach().contiguous(), r = rotations.detach().contiguous(),
                     o = opacities.detach().contiguous();
    c10::hip::HIPGuard guard(idx);
    void* stream = (void*)c10::hip::getCurrentHIPStream(idx).stream();
    if (!cull_rays.has_value()) {
        auto since = st.since_full.find(idx);
        const bool can_refit = since != st.since_full.end() && since->second >= 0 && st.full_P[idx] == (int)P && P > 0;
        auto rn = st.refit_next.find(idx);
        const bool asked = rn != st.refit_next.end() && rn->second;
        if (rn != st.refit_next.end()) rn->second = false;
        if (can_refit && (asked || (st.refit_interval > 0 && since->second < st.refit_interval))) {
            check_rc(lrt_refit(h, (int)P, fptr(m), fptr(s), fptr(r), fptr(o), (float)scale_modifier, stream), "lrt_refit");
            since->second += 1;
        } else {
            check_rc(lrt_build(h, (int)P, fptr(m), fptr(s), fptr(r), fptr(o), (float)scale_modifier, stream), "lrt_build");
            st.since_full[idx] = 0; st.full_P[idx] = (int)P;
        }
    } else {
        st.since_full[idx] = -1;
        const at::Tensor& ro_ = std::get<0>(*cull_rays); const at::Tensor& rd_ = std::get<1>(*cull_rays);
        check_f32_cuda(ro_, "cull ray_o"); check_f32_cuda(rd_, "cull ray_d");
        const at::Tensor ro = ro_.detach().contiguous(), rd = rd_.detach().contiguous();
        TORCH_CHECK(ro.numel() == rd.numel() && rd.numel() % 3 == 0, "cull_rays must be two (...,3) tensors of the same size");
        if (rd.dim() == 3 && rd.size(2) == 3 && ro.sizes() == rd.sizes())        // a (H, W, 3) slab of a range image: the wedge between its edge columns culls too
            check_rc(lrt_build_for_slab(h, (int)P, fptr(m), fptr(s), fptr(r), fptr(o), (float)scale_modifier, (int)rd.size(0), (int)rd.size(1), fptr(ro), fptr(rd), stream),
                     "lrt_build_for_slab");
        else
        check_rc(lrt_build_for_rays(h, (int)P, fptr(m), fptr(s), fptr(r), fptr(o), (float)scale_modifier, (int)(rd.numel() / 3), fptr(ro), fptr(rd), stream),
                 "lrt_build_for_rays");
    }
    st.dirty[idx] = false; st.built_P[idx] = (int)P; st.built_mod[idx] = (float)scale_modifier;
}

int64_t prep(const at::Tensor& ray_o, const at::Tensor& ray_d, const at::Tensor& background, const at::Tensor& means3D, const at::Tensor& shs,
             const c10::optional<at::Tensor>& colors_precomp, const at::Tensor& opacities, const at::Tensor& scales, const at::Tensor& rotations,
             const c10::optional<at::Tensor>& transMat_precomp)
{
    if (means3D.dim() != 2 || means3D.size(1) != 3) AT_ERROR("means3D must have dimensions (num_points, 3)");     // trace_surfels.cpp:178-180
    check_f32_cuda(ray_o, "ray_o"); check_f32_cuda(ray_d, "ray_d"); check_f32_cuda(background, "background"); check_f32_cuda(means3D, "means3D");
    check_f32_cuda(shs, "shs"); check_f32_cuda(opacities, "opacities"); check_f32_cuda(scales, "scales"); check_f32_cuda(rotations, "rotations");
    if (present(colors_precomp))
        not_implemented("colors_precomp is not supported (the reference forward kernel ignores it and reads shs unconditionally, forward.cu:261-263); pass shs");
    if (present(transMat_precomp))
        not_implemented("cov3Ds_precomp / transMat_precomp is not supported (unused by the reference kernels)");
    TORCH_CHECK(ray_o.dim() == 3 && ray_o.size(2) == 3 && ray_d.sizes() == ray_o.sizes(), "ray_o and ray_d must have dimensions (H, W, 3)");
    const int64_t P = means3D.size(0);
    TORCH_CHECK(P == 0 || (shs.dim() == 3 && shs.size(0) == P && shs.size(2) == 3), "shs must have dimensions (num_points, M, 3)");
    return P;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor>
trace_surfels(State& st, bool training, const at::Tensor& ray_o, const at::Tensor& ray_d, const c10::optional<at::Tensor>& vertices,
              const at::Tensor& background, const at::Tensor& means3D, const at::Tensor& shs, int degree,
              const c10::optional<at::Tensor>& colors_precomp, const at::Tensor& opacities, const at::Tensor& scales, double scale_modifier,
              const at::Tensor& rotations, const c10::optional<at::Tensor>& transMat_precomp, const c10::optional<at::Tensor>& viewmatrix,
              const c10::optional<at::Tensor>& projmatrix, const c10::optional<at::Tensor>& campos, bool prefiltered, bool debug)
{
    (void)vertices; (void)viewmatrix; (void)projmatrix; (void)campos; (void)prefiltered; (void)debug;      // dead in the reference kernels too
    const int64_t P = prep(ray_o, ray_d, background, means3D, shs, colors_precomp, opacities, scales, rotations, transMat_precomp);
    const int64_t H = ray_o.size(0), W = ray_o.size(1), M = shs.numel() > 0 ? shs.size(1) : 0;
    const int idx = device_index(means3D);
    lrt_state* h = st.handle(idx);
    auto d_ = st.dirty.find(idx); auto bp = st.built_P.find(idx); auto bm = st.built_mod.find(idx);
    if (d_ == st.dirty.end() || d_->second || bp == st.built_P.end() || bp->second != (int)P || bm == st.built_mod.end() || bm->second != (float)scale_modifier)
        build_from_gaussians(st, means3D, scales, rotations, opacities, scale_modifier, c10::nullopt);     // also when the scale modifier changed
    const at::Tensor ro = ray_o.detach().contiguous(), rd = ray_d.detach().contiguous();                   // ray_o is an expanded view upstream
    const at::Tensor bg = background.detach().contiguous(), sh = shs.detach().contiguous();
    auto f32 = at::TensorOptions().dtype(at::kFloat).device(means3D.device());
    at::Tensor out = at::empty({H, W, 9}, f32), out_i = at::empty({H, W, 1}, f32.dtype(at::kInt)), accum = at::empty({P}, f32);
    c10::hip::HIPGuard guard(idx);
    check_rc(lrt_forward(h, (int)H, (int)W, fptr(ro), fptr(rd), (int)P, (int)M, degree, fptr(sh), fptr(bg), training ? 1 : 0, fptr_mut(out),
                         out_i.numel() > 0 ? out_i.data_ptr<int32_t>() : nullptr, fptr_mut(accum), (void*)c10::hip::getCurrentHIPStream(idx).stream()),
             "lrt_forward");
    st.last_serial = py::int_(lrt_forward_serial(h));
    return {out, out_i, accum};
}

py::tuple
trace_surfels_backward(State& st, const at::Tensor& ray_o, const at::Tensor& ray_d, const c10::optional<at::Tensor>& vertices,
                       const at::Tensor& background, const at::Tensor& means3D, const at::Tensor& shs, int degree,
                       const c10::optional<at::Tensor>& colors_precomp, const at::Tensor& opacities, const at::Tensor& scales, double scale_modifier,
                       const at::Tensor& rotations, const c10::optional<at::Tensor>& transMat_precomp, const c10::optional<at::Tensor>& viewmatrix,
                       const c10::optional<at::Tensor>& projmatrix, const c10::optional<at::Tensor>& campos, bool prefiltered, bool debug,
                       const at::Tensor& out_attr_float32, const c10::optional<at::Tensor>& out_attr_uint32, const at::Tensor& dL_dout_attr_float32,
                       const py::object& grads_out, const py::object& forward_serial, const c10::optional<at::Tensor>& accum_out)
{
    (void)vertices; (void)viewmatrix; (void)projmatrix; (void)campos; (void)prefiltered; (void)debug; (void)out_attr_uint32; (void)scale_modifier;
    const int64_t P = prep(ray_o, ray_d, background, means3D, shs, colors_precomp, opacities, scales, rotations, transMat_precomp);
    const int64_t H = ray_o.size(0), W = ray_o.size(1), M = shs.numel() > 0 ? shs.size(1) : 0;
    const int idx = device_index(means3D);
    lrt_state* h = st.handle(idx);
    auto bp = st.built_P.find(idx);
    TORCH_CHECK(bp != st.built_P.end() && bp->second == (int)P, "trace_surfels_backward: acceleration structure does not match (run forward first)");
    if (!forward_serial.is_none() && lrt_forward_serial(h) != forward_serial.cast<long long>())
        check_rc(lrt_set_option(h, "invalidate_record", 1), "lrt_set_option");   // another forward replaced our hit record: re-trace like the reference
    const at::Tensor ro = ray_o.detach().contiguous(), rd = ray_d.detach().contiguous(), bg = background.detach().contiguous();
    const at::Tensor m = means3D.detach().contiguous(), s = scales.detach().contiguous(), r = rotations.detach().contiguous(),
                     o = opacities.detach().contiguous(), sh = shs.detach().contiguous();
    const at::Tensor out = out_attr_float32.detach().contiguous(), dL = dL_dout_attr_float32.detach().contiguous().to(at::kFloat);
    auto f32 = at::TensorOptions().dtype(at::kFloat).device(means3D.device());
    at::Tensor d_means, d_shs, d_opac, d_scales, d_rot;
    const bool own = grads_out.is_none();
    if (own) {
        // one allocation, [means | shs | opacities | scales | rotations]: the library zero-fills adjacent buffers in one launch
        at::Tensor flat = at::empty({P * (10 + 3 * M)}, f32);
        int64_t o0 = 0;
        d_means = flat.narrow(0, o0, 3 * P).view({P, 3}); o0 += 3 * P;
        d_shs = flat.narrow(0, o0, 3 * M * P).view({P, M, 3}); o0 += 3 * M * P;
        d_opac = flat.narrow(0, o0, P).view({P, 1}); o0 += P;
        d_scales = flat.narrow(0, o0, 2 * P).view({P, 2}); o0 += 2 * P;
        d_rot = flat.narrow(0, o0, 4 * P).view({P, 4});
    } else {
        // extension: preallocated contiguous fp32 tensors to write into (views of one flat buffer for a fused exchange)
        py::dict g = grads_out.cast<py::dict>();
        d_means = g["means"].cast<at::Tensor>(); d_shs = g["shs"].cast<at::Tensor>(); d_opac = g["opacities"].cast<at::Tensor>();
        d_scales = g["scales"].cast<at::Tensor>(); d_rot = g["rotations"].cast<at::Tensor>();
        auto ok = [&](const at::Tensor& t, at::IntArrayRef shp) {
            return t.sizes() == shp && t.is_contiguous() && t.scalar_type() == at::kFloat && t.device() == means3D.device(); };
        TORCH_CHECK(ok(d_means, {P, 3}) && ok(d_shs, {P, M, 3}) && ok(d_opac, {P, 1}) && ok(d_scales, {P, 2}) && ok(d_rot, {P, 4}),
                    "grads_out tensors must be contiguous float32 device tensors of the gradient shapes");
    }
    // option deferred_accum: the (P,) accum tensor the forward returned all-zero is completed here (lrt_backward_accum)
    at::Tensor acc;
    if (accum_out.has_value() && accum_out->defined()) {
        acc = *accum_out;
        TORCH_CHECK(acc.dim() == 1 && acc.size(0) == P && acc.is_contiguous() && acc.scalar_type() == at::kFloat && acc.device() == means3D.device(),
                    "accum_out must be a contiguous float32 device tensor of shape (P,)");
    }
    {
        c10::hip::HIPGuard guard(idx);
        check_rc(lrt_backward_accum(h, (int)H, (int)W, fptr(ro), fptr(rd), (int)P, (int)M, degree, fptr(m), fptr(s), fptr(r), fptr(o), fptr(sh), fptr(bg),
                                    fptr(out), fptr(dL), fptr_mut(d_means), fptr_mut(d_shs), fptr_mut(d_opac), fptr_mut(d_scales), fptr_mut(d_rot),
                                    acc.defined() ? fptr_mut(acc) : nullptr, (void*)c10::hip::getCurrentHIPStream(idx).stream()), "lrt_backward");
    }
    if (!own) return py::make_tuple(d_means, d_shs, py::none(), d_opac, d_scales, d_rot, py::none(), py::none());
    // dead outputs of the reference (never written by backward.cu; trace_surfels.cpp:322-329 zero-fills them): zeros of the same shapes
    return py::make_tuple(d_means, d_shs, at::zeros({P, 3}, f32), d_opac, d_scales, d_rot, at::zeros({P, 9}, f32), at::zeros({P, 3}, f32));
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, mod)
{
    mod.doc() = "MI355X-native diff_lidar_tracer._C (PyTorch-ROCm C++ extension on liblrt_hip.so)";
    // a stale extension must not bind a newer (or older) library: the header this file was compiled against names the ABI it expects
    if (lrt_abi_version() != LRT_ABI_VERSION)
        throw std::runtime_error("diff_lidar_tracer._C_ext was built for liblrt_hip ABI " + std::to_string(LRT_ABI_VERSION) + " but the loaded library reports " +
                                 std::to_string(lrt_abi_version()) + ": rebuild with `python -m lidar_rt_amd.build --force`");
    // failures of the C ABI raise lidar_rt_amd._capi.LrtError (a RuntimeError), the class the ctypes binding raises
    py::register_exception_translator([](std::exception_ptr p) {
        try { if (p) std::rethrow_exception(p); }
        catch (const LrtFailure& e) {
            py::object cls = py::module_::import("lidar_rt_amd._capi").attr("LrtError");
            PyErr_SetString(cls.ptr(), e.what());
        }
    });
    py::class_<State>(mod, "OptiXStateWrapper", py::dynamic_attr())
        .def(py::init<std::string>(), py::arg("pkg_dir") = "")
        .def_readonly("pkg_dir", &State::pkg_dir)
        .def_readwrite("refit_interval", &State::refit_interval)
        .def_readwrite("stats_enabled", &State::stats_enabled)
        .def_readwrite("timing_enabled", &State::timing_enabled)
        .def_readwrite("last_serial", &State::last_serial)
        .def_readonly("options", &State::options)
        .def_readonly("_since_full", &State::since_full)
        .def("handle_ptr", [](State& s, int idx) { return (uintptr_t)s.handle(idx); }, "lrt_state* of device `idx` (created on first use)")
        .def("handles", [](State& s) { std::map<int, uintptr_t> m; for (auto& kv : s.handles) m[kv.first] = (uintptr_t)kv.second; return m; })
        .def("set_option", &State::set_option)
        .def("mark_dirty", &State::mark_dirty, py::arg("refit") = false);
    mod.def("build_acceleration_structure", &build_acceleration_structure, py::arg("state"), py::arg("vertices"), py::arg("triangles"), py::arg("rebuild") = 1u);
    mod.def("build_from_gaussians", &build_from_gaussians, py::arg("state"), py::arg("means3D"), py::arg("scales"), py::arg("rotations"), py::arg("opacities"),
            py::arg("scale_modifier") = 1.0, py::arg("cull_rays") = py::none());
    mod.def("trace_surfels", &trace_surfels);
    mod.def("trace_surfels_backward", &trace_surfels_backward, py::arg("state"), py::arg("ray_o"), py::arg("ray_d"), py::arg("vertices"), py::arg("background"),
            py::arg("means3D"), py::arg("shs"), py::arg("degree"), py::arg("colors_precomp"), py::arg("opacities"), py::arg("scales"), py::arg("scale_modifier"),
            py::arg("rotations"), py::arg("transMat_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("campos"), py::arg("prefiltered"),
            py::arg("debug"), py::arg("out_attr_float32"), py::arg("out_attr_uint32"), py::arg("dL_dout_attr_float32"), py::arg("grads_out") = py::none(),
            py::arg("forward_serial") = py::none(), py::arg("accum_out") = py::none());
}
