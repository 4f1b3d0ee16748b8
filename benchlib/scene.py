"""The workload of bench.py: BASELINE.json configs[1] with the dataset and the tiny-cuda-nn field replaced (bench.py's docstring) —
constants, the procedural scene, the two stand-in fields, the ray pool, and the reference examples' render function
(examples/utils.py:87-155) as the timed steps call it."""
import math

import torch
import torch.nn.functional as F

import nerfacc_amd as nerfacc

AABB = [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]
RENDER_STEP = 5e-3
TARGET_SAMPLES = 1 << 18
INIT_RAYS = 1024
GRID_RES = 128
HBM_PEAK_GBS = 8000.0


# ------------------------------------------------------------------------------------------
# procedural scene + torch-native field (stand-ins for nerf_synthetic/lego and tiny-cuda-nn)
# ------------------------------------------------------------------------------------------
def lego_like_density(x: torch.Tensor) -> torch.Tensor:
    """analytic occupancy of a bulldozer-ish union of boxes; x [..., 3] in world units -> bool"""
    def box(c, h):
        c = torch.tensor(c, device=x.device)
        h = torch.tensor(h, device=x.device)
        return ((x - c).abs() <= h).all(-1)

    body = box([0.0, 0.0, -0.25], [0.75, 0.45, 0.2])
    cabin = box([-0.25, 0.0, 0.2], [0.3, 0.35, 0.25]) & ~box([-0.25, 0.0, 0.25], [0.22, 0.4, 0.12])
    plate = box([0.0, 0.0, -0.55], [0.95, 0.7, 0.06])
    arm = box([0.65, 0.0, 0.1], [0.35, 0.08, 0.08]) | box([0.95, 0.0, -0.1], [0.06, 0.4, 0.25])
    studs = (torch.sin(x[..., 0] * 24.0) * torch.sin(x[..., 1] * 24.0) > 0.5) & box([0.0, 0.0, -0.45], [0.9, 0.65, 0.05])
    return body | cabin | plate | arm | studs


class DenseGridField(torch.nn.Module):
    """sigma = exp(g[0](x)), rgb = sigmoid(g[1:4](x)); one 4-channel voxel grid, trilinear lookups
    (one gather pass forward, one scatter pass backward per query)."""

    def __init__(self, aabb, res=128, occ_fn=None):
        super().__init__()
        self.register_buffer("aabb", torch.tensor(aabb, dtype=torch.float32))
        g = (torch.arange(res, dtype=torch.float32) + 0.5) / res
        lo, hi = self.aabb[:3], self.aabb[3:]
        X, Y, Z = torch.meshgrid(g, g, g, indexing="ij")
        pts = torch.stack([X, Y, Z], -1) * (hi - lo) + lo
        occ = (occ_fn or lego_like_density)(pts)
        dens = torch.where(occ, math.log(50.0), math.log(1e-4)).float()
        gen = torch.Generator().manual_seed(42)
        col = torch.randn((3, res, res, res), generator=gen) * 0.5 + (pts.permute(3, 0, 1, 2) * 1.5)
        # stored [1, 4, Z, Y, X] so that grid_sample's (x, y, z) coordinate order needs no shuffle
        vol = torch.cat([dens[None], col], 0).permute(0, 3, 2, 1)
        self.grid = torch.nn.Parameter(vol[None].contiguous())
        self.register_buffer("u_scale", 2.0 / (hi - lo))
        self.register_buffer("u_shift", -2.0 * lo / (hi - lo) - 1.0)

    def _lookup(self, grid, x):
        u = torch.addcmul(self.u_shift, x, self.u_scale).view(1, 1, 1, -1, 3)
        out = F.grid_sample(grid, u, mode="bilinear", padding_mode="border", align_corners=False)
        return out.view(grid.shape[1], -1).t()

    def query_density(self, x):
        return torch.exp(self._lookup(self.grid[:, :1], x))

    def forward(self, x, dirs=None):
        f = self._lookup(self.grid, x)
        return torch.sigmoid(f[:, 1:4]), torch.exp(f[:, :1])


class GridMlpField(torch.nn.Module):
    """`--field grid+mlp`: a dense FEATURE grid (8 channels) decoded by a two-layer MLP — the shape of the reference's NGP
    field (examples/radiance_fields/ngp.py:79-163: encoding + small MLPs) as far as the gradient exchange is concerned: seven
    parameter tensors from 4 floats to 8 x res^3, reached by autograd in reverse order.  Not the headline field: it exists so that
    the multi-GPU path (ExchangeAdam's hooks, chunk order, both exchange modes) is exercised on a multi-tensor graph."""

    def __init__(self, aabb, res=64, feat=8, hidden=32):
        super().__init__()
        self.register_buffer("aabb", torch.tensor(aabb, dtype=torch.float32))
        lo, hi = self.aabb[:3], self.aabb[3:]
        gen = torch.Generator().manual_seed(7)
        self.grid = torch.nn.Parameter(0.1 * torch.randn((1, feat, res, res, res), generator=gen))
        self.l1, self.l2, self.l3 = torch.nn.Linear(feat, hidden), torch.nn.Linear(hidden, hidden), torch.nn.Linear(hidden, 4)
        with torch.no_grad():                       # starts as fog and grey, like the dense-grid student
            self.l3.weight.mul_(0.1)
            self.l3.bias.copy_(torch.tensor([math.log(0.5), 0.0, 0.0, 0.0]))
        self.register_buffer("u_scale", 2.0 / (hi - lo))
        self.register_buffer("u_shift", -2.0 * lo / (hi - lo) - 1.0)

    def _decode(self, x):
        u = torch.addcmul(self.u_shift, x, self.u_scale).view(1, 1, 1, -1, 3)
        f = F.grid_sample(self.grid, u, mode="bilinear", padding_mode="border", align_corners=False).view(self.grid.shape[1], -1).t()
        return self.l3(torch.relu(self.l2(torch.relu(self.l1(f)))))

    def query_density(self, x):
        return torch.exp(self._decode(x)[:, :1].clamp(max=8.0))

    def forward(self, x, dirs=None):
        f = self._decode(x)
        return torch.sigmoid(f[:, 1:4]), torch.exp(f[:, :1].clamp(max=8.0))


def make_ray_pool(n_pool: int, seed: int, device) -> tuple:
    """random pixels of 100 cameras on a radius-4 sphere looking at the origin (OpenGL camera,
    800x800, focal 1111.1)."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    n_cams, W, focal = 100, 800, 0.5 * 800 / math.tan(0.5 * 0.6911112070083618)
    cam_pos = torch.randn((n_cams, 3), generator=gen)
    cam_pos[:, 2] = cam_pos[:, 2].abs() * 0.7 + 0.2                 # upper hemisphere like the dataset
    cam_pos = 4.0 * cam_pos / cam_pos.norm(dim=-1, keepdim=True)
    fwd = -cam_pos / cam_pos.norm(dim=-1, keepdim=True)
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm(dim=-1, keepdim=True)
    true_up = torch.linalg.cross(right, fwd)
    cam = torch.randint(0, n_cams, (n_pool,), generator=gen)
    px = torch.randint(0, W, (n_pool, 2), generator=gen).float() + 0.5
    dx, dy = (px[:, 0] - W / 2) / focal, -(px[:, 1] - W / 2) / focal
    d = fwd[cam] + dx[:, None] * right[cam] + dy[:, None] * true_up[cam]
    d = d / d.norm(dim=-1, keepdim=True)
    return cam_pos[cam].contiguous().to(device), d.contiguous().to(device)


def render_rays(field, est, rays_o, rays_d, bkgd, training: bool):
    """examples/utils.py:54-167 (render_image_with_occgrid), one chunk."""
    def sigma_fn(t_starts, t_ends, ray_indices):
        if t_starts.shape[0] == 0:
            return torch.empty((0,), device=t_starts.device)
        pos = nerfacc.sample_positions(rays_o, rays_d, ray_indices, t_starts, t_ends)
        return field.query_density(pos).squeeze(-1)

    def rgb_sigma_fn(t_starts, t_ends, ray_indices):
        if t_starts.shape[0] == 0:
            return torch.empty((0, 3), device=t_starts.device), torch.empty((0,), device=t_starts.device)
        pos = nerfacc.sample_positions(rays_o, rays_d, ray_indices, t_starts, t_ends)
        rgb, sigma = field(pos)            # (this stand-in field has no view dependence)
        return rgb, sigma.squeeze(-1)

    ray_indices, t_starts, t_ends = est.sampling(rays_o, rays_d, sigma_fn=sigma_fn, near_plane=0.0, far_plane=1e10,
                                                 render_step_size=RENDER_STEP, stratified=training, cone_angle=0.0,
                                                 alpha_thre=0.0)
    rgb, opacity, depth, _ = nerfacc.rendering(t_starts, t_ends, ray_indices, n_rays=rays_o.shape[0],
                                               rgb_sigma_fn=rgb_sigma_fn, render_bkgd=bkgd)
    return rgb, opacity, depth, t_starts.shape[0]


# candidate samples of the most recent estimator.sampling call of render_rays_reference_style (what its sigma_fn was handed: the
# traversal's output before the visibility filter) — the timed steps add it up, no extra read-back
LAST_CALL = {"candidates": 0}


def render_rays_reference_style(field, est, rays_o, rays_d, bkgd, training: bool):
    """examples/utils.py:87-155 as written there: the user-side closures index the rays with plain torch ops
    (`rays_o[ray_indices] + rays_d[ray_indices] * (t_starts + t_ends)[:, None] / 2.0`); only nerfacc's own
    calls (`estimator.sampling`, `nerfacc.rendering`) reach this package."""
    def sigma_fn(t_starts, t_ends, ray_indices):
        LAST_CALL["candidates"] = t_starts.shape[0]
        if t_starts.shape[0] == 0:
            return torch.empty((0,), device=t_starts.device)
        positions = rays_o[ray_indices] + rays_d[ray_indices] * (t_starts + t_ends)[:, None] / 2.0
        return field.query_density(positions).squeeze(-1)

    def rgb_sigma_fn(t_starts, t_ends, ray_indices):
        if t_starts.shape[0] == 0:
            return torch.empty((0, 3), device=t_starts.device), torch.empty((0,), device=t_starts.device)
        positions = rays_o[ray_indices] + rays_d[ray_indices] * (t_starts + t_ends)[:, None] / 2.0
        rgb, sigma = field(positions, rays_d[ray_indices])
        return rgb, sigma.squeeze(-1)

    LAST_CALL["candidates"] = 0
    ray_indices, t_starts, t_ends = est.sampling(rays_o, rays_d, sigma_fn=sigma_fn, near_plane=0.0, far_plane=1e10,
                                                 render_step_size=RENDER_STEP, stratified=training, cone_angle=0.0,
                                                 alpha_thre=0.0)
    rgb, opacity, depth, _ = nerfacc.rendering(t_starts, t_ends, ray_indices, n_rays=rays_o.shape[0],
                                               rgb_sigma_fn=rgb_sigma_fn, render_bkgd=bkgd)
    return rgb, opacity, depth, t_starts.shape[0]
