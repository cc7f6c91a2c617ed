"""CPU restatement of TetGS.get_points_rgb (TEST INFRASTRUCTURE ONLY): Edit_core/tetgs_scene/tetgs_model.py:413-442
with eval_sh of Edit_core/utils/spherical_harmonics.py:117-172, in plain PyTorch so autograd gives the gradients.
Pinned by tests/golden/ref_utils_fixture.npz (outputs and gradients of the reference's own eval_sh)."""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def eval_sh(deg, sh, dirs):
    """sh [..., C, (deg+1)^2], dirs [..., 3] (spherical_harmonics.py:117-172, degrees 0-3)."""
    result = C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            result = (result + C2[0] * xy * sh[..., 4] + C2[1] * yz * sh[..., 5] + C2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
                      + C2[3] * xz * sh[..., 7] + C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + C3[0] * y * (3 * xx - yy) * sh[..., 9] + C3[1] * xy * z * sh[..., 10]
                          + C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                          + C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + C3[5] * z * (xx - yy) * sh[..., 14]
                          + C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return result


def points_rgb(sh_coordinates, sh_levels, positions=None, camera_centers=None, directions=None):
    """tetgs_model.py:413-442."""
    if camera_centers is not None:
        render_directions = torch.nn.functional.normalize(positions - camera_centers, dim=-1)
    elif directions is not None:
        render_directions = directions
    else:
        raise ValueError("Either camera_centers or directions must be provided.")
    sh = sh_coordinates[:, :sh_levels ** 2]
    shs_view = sh.transpose(-1, -2).reshape(-1, 3, sh_levels ** 2)
    sh2rgb = eval_sh(sh_levels - 1, shs_view, render_directions)
    return torch.clamp_min(sh2rgb + 0.5, 0.0).view(-1, 3)
