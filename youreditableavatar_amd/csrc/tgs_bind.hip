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

// (All three kernels: every load first, then the arithmetic, then every store.  Written group by group -- load, compute, store, next group --
// each load is waited for before the store in front of the next one is issued (the pointers may alias as far as the compiler knows): ten
// memory round trips in a row per thread for 88 bytes.)
__global__ __launch_bounds__(256) void k_bind_fwd(const BindArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.P) return;
    const size_t i3 = 3 * (size_t)i;
    float dn = 0.f, sc[3] = {0.f, 0.f, 0.f}, op[3] = {0.f, 0.f, 0.f}, nr[3] = {0.f, 0.f, 0.f}, d = 0.f;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.raw_density) dn = a.raw_density[i];
    if (a.raw_scales) {
#pragma unroll
        for (int c = 0; c < 3; c++) sc[c] = a.raw_scales[i3 + c];
    }
    if (a.raw_quats) q = reinterpret_cast<const float4*>(a.raw_quats)[i];
    if (a.ori_points) {
        d = a.deltas[i];
#pragma unroll
        for (int c = 0; c < 3; c++) { op[c] = a.ori_points[i3 + c]; nr[c] = a.normals[i3 + c]; }
    }
    const float o = 1.0f / (1.0f + expf(-dn));
    const float e0 = expf(sc[0]), e1 = expf(sc[1]), e2 = expf(sc[2]);
    const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    const float p0 = op[0] + nr[0] * d, p1 = op[1] + nr[1] * d, p2 = op[2] + nr[2] * d;
    asm volatile("" :: "v"(o), "v"(e0), "v"(e1), "v"(e2), "v"(inv), "v"(p0), "v"(p1), "v"(p2), "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w));      // (evaluated HERE: sunk into the branches below, each waits for the store in front of it)
    if (a.raw_density) a.opacity[i] = o;
    if (a.raw_scales) { a.scales[i3] = e0; a.scales[i3 + 1] = e1; a.scales[i3 + 2] = e2; }
    if (a.raw_quats) reinterpret_cast<float4*>(a.quats)[i] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    if (a.ori_points) { a.points[i3] = p0; a.points[i3 + 1] = p1; a.points[i3 + 2] = p2; }
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
    const size_t j = ed ? (size_t)(i - a.Pk) : (size_t)i, j3 = 3 * j, i3 = 3 * (size_t)i;          // row inside the group
    const float* dn = ed ? a.edit_density : a.keep_density;
    const float* sc = ed ? a.edit_scales : a.keep_scales;
    const float* qu = ed ? a.edit_quats : a.keep_quats;
    const bool plain = !ed || a.edit_points;                                                      // positions copied / ori + normal * offset
    float dv = 0.f, sv[3] = {0.f, 0.f, 0.f}, pv[3] = {0.f, 0.f, 0.f}, nv[3] = {0.f, 0.f, 0.f}, off = 0.f;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.opacity) dv = dn[j];
    if (a.scales) {
#pragma unroll
        for (int c = 0; c < 3; c++) sv[c] = sc[j3 + c];
    }
    if (a.quats) q = reinterpret_cast<const float4*>(qu)[j];
    if (a.points) {
        if (plain) {
            const float* src = ed ? a.edit_points : a.keep_points;
#pragma unroll
            for (int c = 0; c < 3; c++) pv[c] = src[j3 + c];
        } else {
            off = a.edit_offsets[j];
#pragma unroll
            for (int c = 0; c < 3; c++) { pv[c] = a.ori_edit_points[j3 + c]; nv[c] = a.edit_normals[j3 + c]; }
        }
    }
    const float o = 1.0f / (1.0f + expf(-dv));
    const float e0 = expf(sv[0]), e1 = expf(sv[1]), e2 = expf(sv[2]);
    const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    asm volatile("" :: "v"(o), "v"(e0), "v"(e1), "v"(e2), "v"(inv), "v"(pv[0]), "v"(pv[1]), "v"(pv[2]), "v"(nv[0]), "v"(nv[1]), "v"(nv[2]), "v"(off));
    if (a.opacity) a.opacity[i] = o;
    if (a.scales) { a.scales[i3] = e0; a.scales[i3 + 1] = e1; a.scales[i3 + 2] = e2; }
    if (a.quats) reinterpret_cast<float4*>(a.quats)[i] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    if (a.points) {
#pragma unroll
        for (int c = 0; c < 3; c++) a.points[i3 + c] = plain ? pv[c] : pv[c] + nv[c] * off;
    }
}

__global__ __launch_bounds__(256) void k_bind_bwd(const BindArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.P) return;
    const size_t i3 = 3 * (size_t)i;
    const bool need_gp = a.d_points || a.d_deltas;
    float gp[3] = {0.f, 0.f, 0.f}, nr[3] = {0.f, 0.f, 0.f}, gs[3] = {0.f, 0.f, 0.f}, sv[3] = {0.f, 0.f, 0.f}, o = 0.f, go = 0.f;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f), g = q;
    if (need_gp) {
#pragma unroll
        for (int c = 0; c < 3; c++) gp[c] = a.g_points[i3 + c];
    }
    if (a.d_deltas) {
#pragma unroll
        for (int c = 0; c < 3; c++) nr[c] = a.normals[i3 + c];
    }
    if (a.d_density) { o = a.opacity[i]; go = a.g_opacity[i]; }
    if (a.d_scales) {
#pragma unroll
        for (int c = 0; c < 3; c++) { gs[c] = a.g_scales[i3 + c]; sv[c] = a.scales[i3 + c]; }
    }
    if (a.d_quats) { q = reinterpret_cast<const float4*>(a.raw_quats)[i]; g = reinterpret_cast<const float4*>(a.g_quats)[i]; }
    const float dd = go * o * (1.0f - o);                                                   // sigmoid' from its output
    const float ds0 = gs[0] * sv[0], ds1 = gs[1] * sv[1], ds2 = gs[2] * sv[2];             // exp' = its output
    // d(x / |x|) = (g - n (n . g)) / |x|   (the max(., 1e-12) branch has zero measure)
    const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    const float nx = q.x * inv, ny = q.y * inv, nz = q.z * inv, nw = q.w * inv;
    const float dot = nx * g.x + ny * g.y + nz * g.z + nw * g.w;
    const float dq0 = (g.x - nx * dot) * inv, dq1 = (g.y - ny * dot) * inv, dq2 = (g.z - nz * dot) * inv, dq3 = (g.w - nw * dot) * inv;
    const float dl = gp[0] * nr[0] + gp[1] * nr[1] + gp[2] * nr[2];
    asm volatile("" :: "v"(dd), "v"(ds0), "v"(ds1), "v"(ds2), "v"(dq0), "v"(dq1), "v"(dq2), "v"(dq3), "v"(dl), "v"(gp[0]), "v"(gp[1]), "v"(gp[2]));   // (evaluated here, not in the branches below)
    if (a.d_points) { a.d_points[i3] = gp[0]; a.d_points[i3 + 1] = gp[1]; a.d_points[i3 + 2] = gp[2]; }      // plain positions of a group: the gradient's rows
    if (a.d_density) a.d_density[i] = dd;
    if (a.d_scales) { a.d_scales[i3] = ds0; a.d_scales[i3 + 1] = ds1; a.d_scales[i3 + 2] = ds2; }
    if (a.d_quats) reinterpret_cast<float4*>(a.d_quats)[i] = make_float4(dq0, dq1, dq2, dq3);
    if (a.d_deltas) a.d_deltas[i] = dl;
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
