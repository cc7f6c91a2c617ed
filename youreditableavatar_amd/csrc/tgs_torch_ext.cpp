// tgs_torch_ext.cpp -- the compiled `_C` extension of the drop-in package: torch::Tensor <-> the C ABI of include/tgs_raster.h.
//
// Counterpart of the reference's pybind module (diff-gaussian-rasterization/ext.cpp:15-19) and its glue
// (rasterize_points.cu:35-217, signatures rasterize_points.h:18-67): the same three exports with the same positional
// arguments and return tuples.  Torch is plumbing only -- tensors for device memory (the three state buffers grow through
// resize_ like rasterize_points.cu:27-33), the current HIP stream of the tensors' device -- and everything that computes
// sits behind the C ABI in libtgs_raster.so, which this module links.  Built by plain g++ against the torch / pybind11
// headers (youreditableavatar_amd/build.py: no CUDAExtension, no hipify); host code only, there is no kernel in this file.
#include <torch/extension.h>
// torch on ROCm names its HIP devices "cuda": the guard / stream types that accept that device type are the *MasqueradingAsCUDA ones
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <tuple>

#include "../../include/tgs_raster.h"

namespace {

// resizeFunctional of rasterize_points.cu:27-33 behind the C ABI's one callback: ctx = the three byte tensors.  A FRESH tensor per
// request, not resize_: when the speculative forward asks for the binning buffer a second time (the guess was too small) resize_ would
// copy the whole old buffer -- hundreds of MB of dead data -- into the new one; dropping the old tensor frees it in stream order.
void* alloc_resize(void* ctx, int which, size_t bytes)
{
    torch::Tensor* bufs = static_cast<torch::Tensor*>(ctx);
    if (which < 0 || which > 2) return nullptr;
    bufs[which] = torch::empty({(long long)bytes}, bufs[which].options());
    return bufs[which].data_ptr();
}

[[noreturn]] void raise_last(long long code)
{
    const char* m = tgs_last_error();
    throw std::runtime_error(std::string(m && m[0] ? m : "tgs_raster error") + " (code " + std::to_string(code) + ")");
}

// contiguous fp32 tensor on `dev`; an empty tensor is the reference's "absent" (nullptr: rasterizer_impl.cu:321,389,411 --
// tested by numel() here, not by the data pointer of an empty tensor)
struct Arg {
    torch::Tensor t;
    const float* p = nullptr;
    Arg(const torch::Tensor& x, const c10::Device& dev, const char* name)
    {
        if (!x.defined() || x.numel() == 0) return;
        if (x.scalar_type() != torch::kFloat32) throw std::runtime_error(std::string("expected scalar type Float but found ") + c10::toString(x.scalar_type()) + " for " + name);
        t = (x.device() == dev ? x : x.to(dev)).contiguous();
        p = t.data_ptr<float>();
    }
};

c10::Device require_gpu(const torch::Tensor& means3D)
{
    if (!means3D.is_cuda()) throw std::runtime_error("diff_gaussian_rasterization (MI355X build) has no CPU path: means3D must be on a HIP device");
    return means3D.device();
}

int sh_coeffs(const torch::Tensor& sh) { return (sh.dim() > 1 && sh.size(0) != 0) ? (int)sh.size(1) : 0; }   // rasterize_points.cu:83-87

// keyword-only extensions -> tgs_options_t (None / 0 = library defaults)
tgs_options_t make_options(int64_t tile_bound, c10::optional<bool> pruning, c10::optional<bool> deterministic, int64_t sort_lds_cap, int64_t mid_bound,
                           c10::optional<bool> light_tiles)
{
    tgs_options_t o;
    memset(&o, 0, sizeof(o));
    o.struct_size = (uint32_t)sizeof(o);
    o.instance_pruning = pruning ? (*pruning ? 1 : 0) : -1;
    o.deterministic = deterministic ? (*deterministic ? 1 : 0) : -1;
    o.sort_lds_cap = sort_lds_cap > 0 ? (uint32_t)sort_lds_cap : 0u;
    o.tile_bound = tile_bound > 0 ? tile_bound : 0;
    o.mid_bound = mid_bound > 0 ? mid_bound : 0;
    o.light_tiles = light_tiles ? (*light_tiles ? 1 : 0) : -1;
    return o;
}

// RasterizeGaussiansCUDA (rasterize_points.cu:35-115).  r_capacity / r_guess: the sync-free / speculative forwards of
// include/tgs_raster.h (extensions; None = the reference's protocol).  With r_guess or info=True the tuple has two more elements: the true
// num_rendered and the pair (tiles with instances, tiles with >= 128 instances) (-1 where the call did not read the frame's Meta).  tile_bound / pruning /
// sort_lds_cap travel as an explicit tgs_options_t: nothing process-wide is touched, so threads with different options do not interfere.
py::tuple rasterize_gaussians(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors, const torch::Tensor& opacity,
                              const torch::Tensor& scales, const torch::Tensor& rotations, float scale_modifier, const torch::Tensor& cov3D_precomp,
                              const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, float tan_fovx, float tan_fovy, int image_height,
                              int image_width, const torch::Tensor& sh, int degree, const torch::Tensor& campos, bool prefiltered, bool debug,
                              c10::optional<int64_t> r_capacity, c10::optional<int64_t> r_guess, int64_t tile_bound, c10::optional<bool> pruning,
                              int64_t sort_lds_cap, bool info, int64_t mid_bound, c10::optional<bool> light_tiles)
{
    if (means3D.dim() != 2 || means3D.size(1) != 3) throw std::runtime_error("means3D must have dimensions (num_points, 3)");   // rasterize_points.cu:57-59
    const c10::Device dev = require_gpu(means3D);
    const int P = (int)means3D.size(0), H = image_height, W = image_width, M = sh_coeffs(sh);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
    torch::Tensor out_color = torch::empty({3, H, W}, f32);
    torch::Tensor radii = torch::empty({P}, f32.dtype(torch::kInt32));
    torch::Tensor bufs[3];
    for (auto& b : bufs) b = torch::empty({0}, f32.dtype(torch::kByte));
    const Arg bg(background, dev, "background"), means(means3D, dev, "means3D"), col(colors, dev, "colors"), op(opacity, dev, "opacity"),
        sc(scales, dev, "scales"), rot(rotations, dev, "rotations"), cov(cov3D_precomp, dev, "cov3D_precomp"), view(viewmatrix, dev, "viewmatrix"),
        proj(projmatrix, dev, "projmatrix"), shs(sh, dev, "sh"), cam(campos, dev, "campos");
    void* stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
    const tgs_options_t opt = make_options(tile_bound, pruning, c10::nullopt, sort_lds_cap, mid_bound, light_tiles);
    tgs_frame_info_t fi;
    const int mode = r_guess ? TGS_FWD_SPECULATIVE : (r_capacity ? TGS_FWD_ASYNC : TGS_FWD_SYNC);
    int64_t r;
    {
        py::gil_scoped_release nogil;                       // (the allocation callback uses the C++ tensor API only)
        r = tgs_forward_opt(&opt, mode, r_guess ? *r_guess : (r_capacity ? *r_capacity : 0), &fi, alloc_resize, bufs, stream, P, degree, M, bg.p, W, H, means.p, shs.p,
                            col.p, op.p, sc.p, scale_modifier, rot.p, cov.p, view.p, proj.p, cam.p, tan_fovx, tan_fovy, prefiltered, out_color.data_ptr<float>(),
                            P ? radii.data_ptr<int>() : nullptr, debug);
    }
    if (r < 0) raise_last(r);
    if (r_guess || info) return py::make_tuple(r, out_color, radii, bufs[0], bufs[1], bufs[2], fi.num_rendered, py::make_tuple(fi.nonempty_tiles, (int64_t)fi.mid_tiles));
    return py::make_tuple(r, out_color, radii, bufs[0], bufs[1], bufs[2]);
}

// RasterizeGaussiansBackwardCUDA (rasterize_points.cu:117-196); return order of :195.  The library writes every element, so the outputs
// are torch::empty (the reference needs nine torch::zeros memsets, :151-159).  with_conic (tests) appends the scratch dL_dconic[P,2,2].
py::tuple rasterize_gaussians_backward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii, const torch::Tensor& colors,
                                       const torch::Tensor& scales, const torch::Tensor& rotations, float scale_modifier, const torch::Tensor& cov3D_precomp,
                                       const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, float tan_fovx, float tan_fovy,
                                       const torch::Tensor& dL_dout_color, const torch::Tensor& sh, int degree, const torch::Tensor& campos,
                                       const torch::Tensor& geomBuffer, int64_t R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, bool debug,
                                       bool with_conic, int64_t tile_bound, c10::optional<bool> deterministic, int64_t mid_bound, c10::optional<bool> light_tiles,
                                       bool need_colors, bool need_cov3D)
{
    // need_colors / need_cov3D = false (extension keywords; the reference's signature has neither): the caller will discard dL_dcolors / dL_dcov3D --
    // GaussianRasterizer's backward does when the forward was given shs / scales + rotations (__init__.py:137-152 hands them to inputs that are None) --
    // so they are neither allocated nor written (empty tensors come back); dL_dconic, an intermediate of the reference's two kernels, only on request.
    const c10::Device dev = require_gpu(means3D);
    const int P = (int)means3D.size(0), H = (int)dL_dout_color.size(1), W = (int)dL_dout_color.size(2), M = sh_coeffs(sh);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    const auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
    const Arg sc(scales, dev, "scales"), rot(rotations, dev, "rotations");
    const bool has_sr = sc.p != nullptr;
    const bool want_col = need_colors || M == 0, want_cov = need_cov3D || !has_sr;      // (required outputs on the colors_precomp / cov3D_precomp paths)
    torch::Tensor dL_dmeans3D = torch::empty({P, 3}, f32), dL_dmeans2D = torch::empty({P, 3}, f32), dL_dcolors = torch::empty({want_col ? P : 0, 3}, f32),
                  dL_dconic = torch::empty({with_conic ? P : 0, 2, 2}, f32), dL_dopacity = torch::empty({P, 1}, f32), dL_dcov3D = torch::empty({want_cov ? P : 0, 6}, f32),
                  dL_dsh = torch::empty({P, M, 3}, f32);
    // the reference leaves these at zero on the cov3D_precomp path
    torch::Tensor dL_dscales = has_sr ? torch::empty({P, 3}, f32) : torch::zeros({P, 3}, f32), dL_drotations = has_sr ? torch::empty({P, 4}, f32) : torch::zeros({P, 4}, f32);
    if (P != 0) {
        const Arg bg(background, dev, "background"), means(means3D, dev, "means3D"), col(colors, dev, "colors"), cov(cov3D_precomp, dev, "cov3D_precomp"),
            view(viewmatrix, dev, "viewmatrix"), proj(projmatrix, dev, "projmatrix"), shs(sh, dev, "sh"), cam(campos, dev, "campos"),
            dL(dL_dout_color, dev, "dL_dout_color");
        const torch::Tensor radii_c = radii.contiguous();
        void* stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        const tgs_options_t opt = make_options(tile_bound, c10::nullopt, deterministic, 0, mid_bound, light_tiles);
        int r;
        {
            py::gil_scoped_release nogil;
            r = tgs_backward_opt(&opt, 0, stream, P, degree, M, R, bg.p, W, H, means.p, shs.p, col.p, sc.p, scale_modifier, rot.p, cov.p, view.p, proj.p, cam.p, tan_fovx,
                                 tan_fovy, radii_c.data_ptr<int>(), geomBuffer.data_ptr(), binningBuffer.data_ptr(), imageBuffer.data_ptr(), dL.p,
                                 dL_dmeans2D.data_ptr<float>(), with_conic ? dL_dconic.data_ptr<float>() : nullptr, dL_dopacity.data_ptr<float>(),
                                 want_col ? dL_dcolors.data_ptr<float>() : nullptr, dL_dmeans3D.data_ptr<float>(), want_cov ? dL_dcov3D.data_ptr<float>() : nullptr,
                                 M ? dL_dsh.data_ptr<float>() : nullptr,
                                 has_sr ? dL_dscales.data_ptr<float>() : nullptr, has_sr ? dL_drotations.data_ptr<float>() : nullptr, debug);
        }
        if (r < 0) raise_last(r);
    }
    if (with_conic) return py::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dconic);
    return py::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations);
}

// markVisible (rasterize_points.cu:198-217)
torch::Tensor mark_visible(const torch::Tensor& means3D, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix)
{
    const c10::Device dev = require_gpu(means3D);
    const int P = (int)means3D.size(0);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dev);
    torch::Tensor present = torch::zeros({P}, torch::TensorOptions().dtype(torch::kBool).device(dev));
    if (P != 0) {
        const Arg m(means3D, dev, "means3D"), v(viewmatrix, dev, "viewmatrix"), pj(projmatrix, dev, "projmatrix");
        void* stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
        const int r = tgs_mark_visible(stream, P, m.p, v.p, pj.p, (uint8_t*)present.data_ptr());
        if (r < 0) raise_last(r);
    }
    return present;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "compiled glue of the MI355X-native diff_gaussian_rasterization: torch::Tensor <-> C ABI (include/tgs_raster.h)";
    m.def("rasterize_gaussians", &rasterize_gaussians, py::arg("background"), py::arg("means3D"), py::arg("colors"), py::arg("opacity"), py::arg("scales"),
          py::arg("rotations"), py::arg("scale_modifier"), py::arg("cov3D_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("tan_fovx"),
          py::arg("tan_fovy"), py::arg("image_height"), py::arg("image_width"), py::arg("sh"), py::arg("degree"), py::arg("campos"), py::arg("prefiltered"),
          py::arg("debug"), py::arg("r_capacity") = py::none(), py::arg("r_guess") = py::none(), py::arg("tile_bound") = 0, py::arg("pruning") = py::none(),
          py::arg("sort_lds_cap") = 0, py::arg("info") = false, py::arg("mid_bound") = 0, py::arg("light_tiles") = py::none());
    m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward, py::arg("background"), py::arg("means3D"), py::arg("radii"), py::arg("colors"),
          py::arg("scales"), py::arg("rotations"), py::arg("scale_modifier"), py::arg("cov3D_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"),
          py::arg("tan_fovx"), py::arg("tan_fovy"), py::arg("dL_dout_color"), py::arg("sh"), py::arg("degree"), py::arg("campos"), py::arg("geomBuffer"),
          py::arg("R"), py::arg("binningBuffer"), py::arg("imageBuffer"), py::arg("debug"), py::arg("_with_conic") = false, py::arg("tile_bound") = 0,
          py::arg("deterministic") = py::none(), py::arg("mid_bound") = 0, py::arg("light_tiles") = py::none(), py::arg("need_colors") = true,
          py::arg("need_cov3D") = true);
    m.def("mark_visible", &mark_visible);
    m.def("abi_version", []() { return tgs_abi_version(); });
    m.def("compiled_abi_version", []() { return TGS_ABI_VERSION; });      // the header THIS module was compiled against
    m.def("sizeof_view", []() { return sizeof(tgs_view_t); });
}
