"""Cone-beam ray geometry -- drop-in for the reference's ``train/proj_helpers.py`` (lines 50-90)."""
import numpy as np
import torch


def _rot(axis: str, angle: float) -> np.ndarray:
    c, s = np.cos(angle), np.sin(angle)
    m = np.identity(4)
    i, j = {"x": (1, 2), "y": (2, 0), "z": (0, 1)}[axis]
    m[i, i], m[i, j], m[j, i], m[j, j] = c, -s, s, c
    return m


def x_rotation_matrix(angle):
    return _rot("x", angle)


def y_rotation_matrix(angle):
    return _rot("y", angle)


def z_rotation_matrix(angle):
    return _rot("z", angle)


def translation_matrix(vec):
    m = np.identity(4)
    m[:3, 3] = vec[:3]
    return m


def get_rotation_matrix_tigre(theta, phi, larm=0):
    """Rz(-theta) . Rz(pi/2) . Rx(phi) . Rx(-pi/2)  (proj_helpers.py:50-57)."""
    inner = np.dot(z_rotation_matrix(np.pi / 2), x_rotation_matrix(np.deg2rad(phi)))
    return np.dot(np.dot(z_rotation_matrix(-np.deg2rad(theta)), inner), x_rotation_matrix(-np.pi / 2))


def source_matrix_tigre(source_pt, theta, phi, larm=0):
    return get_rotation_matrix_tigre(theta, phi, larm).dot(translation_matrix(source_pt))


def get_ray_values_tigre(theta, phi, larm, geo, device):
    """(origins, directions) f32[W,H,3] of one projection (proj_helpers.py:65-90).  Pixel (w,h):
    u=(w+.5-W/2)*dDet0+off0, v=(h+.5-H/2)*dDet1+off1, dir = R [u/DSD, v/DSD, 1] (not normalised)."""
    pose = torch.from_numpy(source_matrix_tigre(np.array([0, 0, -geo["DSO"]]), theta, phi, larm)).to(device).float()
    W, H = geo["nDetector"]
    gi, gj = torch.meshgrid(torch.linspace(0, W - 1, W).to(device), torch.linspace(0, H - 1, H).to(device), indexing="xy")
    u = (gi.t() + 0.5 - W / 2) * geo["dDetector"][0] + geo["offDetector"][0]
    v = (gj.t() + 0.5 - H / 2) * geo["dDetector"][1] + geo["offDetector"][1]
    local = torch.stack([u / geo["DSD"], v / geo["DSD"], torch.ones_like(u)], -1)
    dirs = torch.sum(torch.matmul(pose[:3, :3], local[..., None]), -1)
    orig = pose[:3, -1].expand(dirs.shape)
    return orig.cpu().numpy(), dirs.cpu().numpy()


def get_depth_values(near_thresh, far_thresh, depth_samples_per_ray, device, stratified=True):
    t = torch.linspace(0.0, 1.0, depth_samples_per_ray)
    z = near_thresh * (1.0 - t) + far_thresh * t
    if stratified:
        mid = 0.5 * (z[..., 1:] + z[..., :-1])
        hi, lo = torch.cat([mid, z[..., -1:]], -1), torch.cat([z[..., :1], mid], -1)
        z = lo + (hi - lo) * torch.rand(z.shape)
    return z.to(device)
