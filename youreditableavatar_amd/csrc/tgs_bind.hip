// tgs_bind.hip -- the binding-side per-step assembly of the rasterizer's inputs, fused (north_star: "tetgs_scene Gaussian model bindings").
//
// Every optimisation step of the reference turns its raw parameters into what the rasterizer takes through four properties of the model
// class, each a separate PyTorch kernel forward and one or two backward (Edit_core/tetgs_scene/tetgs_model.py):
//   strengths    :261-265   opacity  = sigmoid(all_densities)
//   scaling      :279-281   scales   = exp(_scales)                        (scale_activation = torch.exp, :16)
//   quaternions  :283-286   quats    = F.normalize(_quaternions, dim=-1)   (x / max(|x|, 1e-12))
//   points       :252-258   points   = ori_points + normals * _points      (mesh-bound Gaussians moving along their face normal: the learnable
//                                                                            _points is one offset per Gaussian, :168-170)
// tetgs_edit_2d.py:199-208 / tetgs_edit_3d.py store their flat Gaussians' scales the same way (log of (1e-8, r, r)), so they take the same path.
// Here: one kernel forward, one backward, one thread per Gaussian, 12 + 16 + 4 (+ 28) bytes in and out -- pure HBM streaming.
//
// The editing stages (EditTetGS tetgs_edit_2d.py:280-318, Edit3DTetGS tetgs_edit_3d.py:272-331 -- 16 800 + 2000 of the reference's ~22 800
// rasterizer iterations) hold TWO groups: the reconstructed Gaussians ("keep", frozen: requires_grad=False, tetgs_edit_2d.py:237-262) and
// the edited ones ("edit", learnable); every property is torch.cat([keep, edit]) followed by the activation, every step, and autograd
// slices the gradient apart again.  tgs_bind_groups_forward writes the activations of both groups into the concatenated [Pk + Pe] outputs
// in one launch (no cat, no [Pk + Pe] raw intermediates); tgs_bind_groups_backward walks the edit rows only.
#include "tgs_device.hpp"
#include "../../include/tgs_raster.h"

namespace tgs {

struct BindArgs {
    int P;
    const float *raw_density, *raw_scales, *raw_quats, *ori_points, *normals, *deltas;     // any group may be NULL
    float *opacity, *scales, *quats, *points;
    const float *g_opacity, *g_scales, *g_quats, *g_points;                                // backward
    float *d_density, *d_scales, *d_quats, *d_deltas;
    float* d_points;                                                                        // [P,3] (grouped backward with plain edit positions)
};

__global__ __launch_bounds__(256) void k_bind_fwd(const BindArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.P) return;
    if (a.raw_density) a.opacity[i] = 1.0f / (1.0f + expf(-a.raw_density[i]));
    if (a.raw_scales) {
#pragma unroll
        for (int c = 0; c < 3; c++) a.scales[3 * (size_t)i + c] = expf(a.raw_scales[3 * (size_t)i + c]);
    }
    if (a.raw_quats) {
        const float4 q = reinterpret_cast<const float4*>(a.raw_quats)[i];
        const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
        reinterpret_cast<float4*>(a.quats)[i] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    }
    if (a.ori_points) {
        const float d = a.deltas[i];
#pragma unroll
        for (int c = 0; c < 3; c++) a.points[3 * (size_t)i + c] = a.ori_points[3 * (size_t)i + c] + a.normals[3 * (size_t)i + c] * d;
    }
}

// Two groups into one set of outputs: rows [0, Pk) from the keep group, rows [Pk, Pk + Pe) from the edit group.  Positions: the keep
// group's are plain [Pk,3]; the edit group's are either plain [Pe,3] (EditTetGS.points, tetgs_edit_2d.py:281-283) or
// ori_edit_points + _edit_normals * _edit_points with one offset per Gaussian (Edit3DTetGS.points, tetgs_edit_3d.py:273-278).
struct BindGroupArgs {
    int Pk, Pe;
    const float *keep_density, *keep_scales, *keep_quats, *keep_points;
    const float *edit_density, *edit_scales, *edit_quats, *edit_points, *ori_edit_points, *edit_normals, *edit_offsets;
    float *opacity, *scales, *quats, *points;
};

__global__ __launch_bounds__(256) void k_bind_groups_fwd(const BindGroupArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.Pk + a.Pe) return;
    const bool ed = i >= a.Pk;
    const size_t j = ed ? (size_t)(i - a.Pk) : (size_t)i;                                   // row inside the group
    const float* dn = ed ? a.edit_density : a.keep_density;
    const float* sc = ed ? a.edit_scales : a.keep_scales;
    const float* qu = ed ? a.edit_quats : a.keep_quats;
    if (a.opacity) a.opacity[i] = 1.0f / (1.0f + expf(-dn[j]));
    if (a.scales) {
#pragma unroll
        for (int c = 0; c < 3; c++) a.scales[3 * (size_t)i + c] = expf(sc[3 * j + c]);
    }
    if (a.quats) {
        const float4 q = reinterpret_cast<const float4*>(qu)[j];
        const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
        reinterpret_cast<float4*>(a.quats)[i] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    }
    if (a.points) {
        if (!ed || a.edit_points) {
            const float* src = ed ? a.edit_points : a.keep_points;
#pragma unroll
            for (int c = 0; c < 3; c++) a.points[3 * (size_t)i + c] = src[3 * j + c];
        } else {
            const float d = a.edit_offsets[j];
#pragma unroll
            for (int c = 0; c < 3; c++) a.points[3 * (size_t)i + c] = a.ori_edit_points[3 * j + c] + a.edit_normals[3 * j + c] * d;
        }
    }
}

__global__ __launch_bounds__(256) void k_bind_bwd(const BindArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.P) return;
    if (a.d_points) {                                                                       // plain positions of a group: the gradient's rows
#pragma unroll
        for (int c = 0; c < 3; c++) a.d_points[3 * (size_t)i + c] = a.g_points[3 * (size_t)i + c];
    }
    if (a.d_density) { const float o = a.opacity[i]; a.d_density[i] = a.g_opacity[i] * o * (1.0f - o); }            // sigmoid' from its output
    if (a.d_scales) {
#pragma unroll
        for (int c = 0; c < 3; c++) a.d_scales[3 * (size_t)i + c] = a.g_scales[3 * (size_t)i + c] * a.scales[3 * (size_t)i + c];   // exp' = its output
    }
    if (a.d_quats) {    // d(x / |x|) = (g - n (n . g)) / |x|   (the max(., 1e-12) branch has zero measure)
        const float4 q = reinterpret_cast<const float4*>(a.raw_quats)[i], g = reinterpret_cast<const float4*>(a.g_quats)[i];
        const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
        const float nx = q.x * inv, ny = q.y * inv, nz = q.z * inv, nw = q.w * inv;
        const float dot = nx * g.x + ny * g.y + nz * g.z + nw * g.w;
        reinterpret_cast<float4*>(a.d_quats)[i] = make_float4((g.x - nx * dot) * inv, (g.y - ny * dot) * inv, (g.z - nz * dot) * inv, (g.w - nw * dot) * inv);
    }
    if (a.d_deltas) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) s += a.g_points[3 * (size_t)i + c] * a.normals[3 * (size_t)i + c];
        a.d_deltas[i] = s;
    }
}

}  // namespace tgs

extern "C" {

int tgs_bind_forward(void* stream, int P, const float* raw_density, const float* raw_scales, const float* raw_quats, const float* ori_points,
                     const float* normals, const float* deltas, float* opacity, float* scales, float* quats, float* points)
{
    using namespace tgs;
    if (P == 0) return TGS_OK;
    if (P < 0 || (raw_density && !opacity) || (raw_scales && !scales) || (raw_quats && !quats) || (ori_points && (!normals || !deltas || !points)))
        return set_error(TGS_ERR_INVALID, "tgs_bind_forward: every given input group needs its output (and ori_points needs normals and deltas)");
    BindArgs a{};
    a.P = P; a.raw_density = raw_density; a.raw_scales = raw_scales; a.raw_quats = raw_quats; a.ori_points = ori_points; a.normals = normals; a.deltas = deltas;
    a.opacity = opacity; a.scales = scales; a.quats = quats; a.points = points;
    hipLaunchKernelGGL(k_bind_fwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hip_status("tgs_bind_forward");
}

int tgs_bind_backward(void* stream, int P, const float* raw_quats, const float* normals, const float* opacity, const float* scales,
                      const float* g_opacity, const float* g_scales, const float* g_quats, const float* g_points,
                      float* d_density, float* d_scales, float* d_quats, float* d_deltas)
{
    using namespace tgs;
    if (P == 0) return TGS_OK;
    if (P < 0 || (d_density && (!opacity || !g_opacity)) || (d_scales && (!scales || !g_scales)) || (d_quats && (!raw_quats || !g_quats)) ||
        (d_deltas && (!normals || !g_points)))
        return set_error(TGS_ERR_INVALID, "tgs_bind_backward: every requested gradient needs its forward output / input and its incoming gradient");
    BindArgs a{};
    a.P = P; a.raw_quats = raw_quats; a.normals = normals; a.opacity = const_cast<float*>(opacity); a.scales = const_cast<float*>(scales);
    a.g_opacity = g_opacity; a.g_scales = g_scales; a.g_quats = g_quats; a.g_points = g_points;
    a.d_density = d_density; a.d_scales = d_scales; a.d_quats = d_quats; a.d_deltas = d_deltas;
    hipLaunchKernelGGL(k_bind_bwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hip_status("tgs_bind_backward");
}

// The two-group form (module comment).  Any output may be NULL (its inputs are then not read); the edit positions are EITHER edit_points
// [Pe,3] OR ori_edit_points [Pe,3] + edit_normals [Pe,3] + edit_offsets [Pe,1].
int tgs_bind_groups_forward(void* stream, int Pk, int Pe, const float* keep_density, const float* keep_scales, const float* keep_quats, const float* keep_points,
                            const float* edit_density, const float* edit_scales, const float* edit_quats, const float* edit_points,
                            const float* ori_edit_points, const float* edit_normals, const float* edit_offsets,
                            float* opacity, float* scales, float* quats, float* points)
{
    using namespace tgs;
    if (Pk < 0 || Pe < 0 || (long long)Pk + Pe > 0x7fffffffLL) return set_error(TGS_ERR_INVALID, "tgs_bind_groups_forward: bad group sizes");
    if (Pk + Pe == 0) return TGS_OK;
    const bool k = Pk > 0, e = Pe > 0;
    if ((opacity && ((k && !keep_density) || (e && !edit_density))) || (scales && ((k && !keep_scales) || (e && !edit_scales))) ||
        (quats && ((k && !keep_quats) || (e && !edit_quats))) ||
        (points && ((k && !keep_points) || (e && !edit_points && !(ori_edit_points && edit_normals && edit_offsets)))))
        return set_error(TGS_ERR_INVALID, "tgs_bind_groups_forward: every requested output needs its input in both groups "
                                          "(edit positions: edit_points, or ori_edit_points + edit_normals + edit_offsets)");
    BindGroupArgs a{};
    a.Pk = Pk; a.Pe = Pe; a.keep_density = keep_density; a.keep_scales = keep_scales; a.keep_quats = keep_quats; a.keep_points = keep_points;
    a.edit_density = edit_density; a.edit_scales = edit_scales; a.edit_quats = edit_quats; a.edit_points = edit_points;
    a.ori_edit_points = ori_edit_points; a.edit_normals = edit_normals; a.edit_offsets = edit_offsets;
    a.opacity = opacity; a.scales = scales; a.quats = quats; a.points = points;
    hipLaunchKernelGGL(k_bind_groups_fwd, dim3((unsigned)((Pk + Pe + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hip_status("tgs_bind_groups_forward");
}

// Gradients of the EDIT group only (the keep group is frozen in the reference).  opacity / scales and the four incoming gradients are
// the full [Pk + Pe] tensors of the forward; the outputs have Pe rows.  d_points [Pe,3] (plain edit positions) and d_offsets [Pe,1]
// (normal-bound edit positions, needs edit_normals) are alternatives.
int tgs_bind_groups_backward(void* stream, int Pk, int Pe, const float* edit_quats, const float* edit_normals, const float* opacity, const float* scales,
                             const float* g_opacity, const float* g_scales, const float* g_quats, const float* g_points,
                             float* d_density, float* d_scales, float* d_quats, float* d_points, float* d_offsets)
{
    using namespace tgs;
    if (Pk < 0 || Pe < 0) return set_error(TGS_ERR_INVALID, "tgs_bind_groups_backward: bad group sizes");
    if (Pe == 0) return TGS_OK;
    if ((d_density && (!opacity || !g_opacity)) || (d_scales && (!scales || !g_scales)) || (d_quats && (!edit_quats || !g_quats)) ||
        (d_offsets && (!edit_normals || !g_points)) || (d_points && !g_points))
        return set_error(TGS_ERR_INVALID, "tgs_bind_groups_backward: every requested gradient needs its forward output / input and its incoming gradient");
    const size_t o = (size_t)Pk;
    auto at = [o](const float* p, int cols) { return p ? p + o * cols : nullptr; };        // the edit rows of a concatenated tensor
    BindArgs a{};
    a.P = Pe; a.raw_quats = edit_quats; a.normals = edit_normals; a.opacity = const_cast<float*>(at(opacity, 1)); a.scales = const_cast<float*>(at(scales, 3));
    a.g_opacity = at(g_opacity, 1); a.g_scales = at(g_scales, 3); a.g_quats = at(g_quats, 4); a.g_points = at(g_points, 3);
    a.d_density = d_density; a.d_scales = d_scales; a.d_quats = d_quats; a.d_deltas = d_offsets; a.d_points = d_points;
    hipLaunchKernelGGL(k_bind_bwd, dim3((unsigned)((Pe + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hip_status("tgs_bind_groups_backward");
}
}
