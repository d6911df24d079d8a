"""
CPU restatement of the Lagrangian transition model (TEST INFRASTRUCTURE — see oracle/__init__.py).

Follows /root/reference/models/transmodel.py:
  B1 integrate_pos_vel :100-104, update_pos_vel :144-148      (PINNED by tests/golden/b1_integrate.npz)
  B3 _window_poly6 :73-77                                      (PINNED, same fixture)
  B2 FixedRadiusSearch / B4 continuous_conv / B6 reduce_subarrays_sum (call sites :86-95, :116-118,
     :125, :135-138) — Open3D 0.15.2, NOT vendored and NOT importable here:  PARITY UNPINNED.
     Restated from Ummenhofer et al., "Lagrangian Fluid Simulation with Continuous Convolutions"
     (ICLR 2020) and the Open3D layer contract:
       radius = extent/2; neighbours with d^2 <= radius^2; identical-position points skipped
       (ignore_query_point); window input = d^2/radius^2; relative position * 2/extent -> unit ball;
       ball_to_cube_volume_preserving = sphere->cylinder->cube (Griepentrog et al.); align_corners=True:
       cube [-1,1] -> [0, size-1]; + offset (0); trilinear interpolation with index clamping; filter
       tensor (kz,ky,kx,Cin,Cout); normalize=False; + bias; no activation.
  B5 dense + residual chain :117-131, B7 forward :151-163.
"""
import math

import numpy as np
import torch

from . import neighbors

PARTICLE_RADIUS = 0.025
RADIUS_SCALE = 1.5
FILTER_EXTENT = float(np.float32(6 * RADIUS_SCALE * PARTICLE_RADIUS))   # transmodel.py:35
LAYER_CHANNELS = [32, 64, 64, 3]                                        # transmodel.py:26
KSIZE = 4


# --------------------------------------------------------------------------------------------
def integrate_pos_vel(pos, vel, gravity, dt):
    vel_new = vel + gravity * dt
    pos_new = pos + (vel + vel_new) / 2 * dt
    return pos_new, vel_new


def update_pos_vel(pos, pos_new, delta, dt):
    pos_c = pos_new + delta
    vel_c = (pos_c - pos) / dt
    return pos_c, vel_c


def window_poly6(r_sqr_norm):
    return torch.clamp((1 - r_sqr_norm) ** 3, 0, 1)


# --------------------------------------------------------------------------------------------
# coordinate mapping (each helper unit-tested on its own: SURVEY §8c "isolate the uncertain items")
# --------------------------------------------------------------------------------------------
def map_sphere_to_cylinder(p):
    """Volume-preserving unit ball -> unit cylinder (radius 1, z in [-1,1])."""
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    sq = x * x + y * y + z * z
    norm = torch.sqrt(sq)
    xy2 = x * x + y * y
    cap = (5.0 / 4.0) * z * z > xy2
    s_cap = torch.sqrt(3 * norm / (norm + z.abs()))
    s_side = norm / torch.sqrt(xy2)
    zero = sq < 1e-12
    sx = torch.where(cap, s_cap, s_side)
    ox = torch.where(zero, torch.zeros_like(x), x * sx)
    oy = torch.where(zero, torch.zeros_like(y), y * sx)
    oz = torch.where(cap, torch.sign(z) * norm, 1.5 * z)
    oz = torch.where(zero, torch.zeros_like(z), oz)
    return torch.stack([ox, oy, oz], -1)


def map_cylinder_to_cube(p):
    """Area-preserving disc -> square on (x,y); z untouched."""
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    xy2 = x * x + y * y
    nxy = torch.sqrt(xy2)
    zero = xy2 < 1e-12
    xmajor = y.abs() <= x.abs()
    four_over_pi = 4.0 / math.pi
    tx = torch.sign(x) * nxy
    ty = torch.sign(y) * nxy
    safe_x = torch.where(x == 0, torch.ones_like(x), x)
    safe_y = torch.where(y == 0, torch.ones_like(y), y)
    ox = torch.where(xmajor, tx, ty * four_over_pi * torch.atan(x / safe_y))
    oy = torch.where(xmajor, tx * four_over_pi * torch.atan(y / safe_x), ty)
    ox = torch.where(zero, torch.zeros_like(x), ox)
    oy = torch.where(zero, torch.zeros_like(y), oy)
    return torch.stack([ox, oy, z], -1)


def filter_coordinates(rel, extent, size=KSIZE):
    """relative positions (nnz,3) -> continuous filter-grid coordinates in [0,size-1]^3 (x,y,z order)."""
    unit = rel * (2.0 / extent)
    cube = map_cylinder_to_cube(map_sphere_to_cylinder(unit))
    return (cube + 1.0) * (0.5 * (size - 1))


def trilinear(coords, size=KSIZE):
    """-> (cell (nnz,8) int64 flattened as (z*size+y)*size+x, weight (nnz,8))."""
    c = torch.clamp(coords, 0.0, float(size - 1))
    c0 = torch.clamp(torch.floor(c), max=float(size - 2))
    f = c - c0
    i0 = c0.to(torch.int64)
    cells, wts = [], []
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                wx = f[:, 0] if dx else 1 - f[:, 0]
                wy = f[:, 1] if dy else 1 - f[:, 1]
                wz = f[:, 2] if dz else 1 - f[:, 2]
                cells.append(((i0[:, 2] + dz) * size + (i0[:, 1] + dy)) * size + (i0[:, 0] + dx))
                wts.append(wx * wy * wz)
    return torch.stack(cells, 1), torch.stack(wts, 1)


# --------------------------------------------------------------------------------------------
def radius_search(points, queries, radius, ignore_query_point=True):
    idx, rs, d2 = neighbors.fixed_radius_search(points.detach().numpy(), queries.detach().numpy(), radius,
                                                ignore_query_point)
    return torch.from_numpy(idx), torch.from_numpy(rs), torch.from_numpy(d2)


def cconv(feats, inp_pos, out_pos, extent, kernel, bias, nbr_idx, row_splits, d2, use_window=True, window_fn=None):
    """ContinuousConv forward.  kernel (4,4,4,Cin,Cout) indexed [z][y][x].  window_fn: the layer's
    window_function, called on d^2/radius^2 (default: poly6, transmodel.py:73-77)."""
    n_out = out_pos.shape[0]
    cout = kernel.shape[-1]
    counts = (row_splits[1:] - row_splits[:-1])
    rows = torch.repeat_interleave(torch.arange(n_out), counts)
    nbr = nbr_idx.to(torch.int64)
    radius = 0.5 * extent
    # Open3D's continuous_conv has gradients w.r.t. filter and input features only: the geometry is detached
    rel = inp_pos.detach()[nbr] - out_pos.detach()[rows]
    imp = (window_fn or window_poly6)(d2 / (radius * radius)) if use_window else torch.ones_like(d2)
    cell, w = trilinear(filter_coordinates(rel, extent))
    # G[j, cell, :] = feats[j] @ kernel[cell]  (transform-then-gather; equal to Open3D's
    # gather-then-GEMM up to summation order)
    G = torch.einsum("ni,cio->nco", feats, kernel.reshape(KSIZE ** 3, kernel.shape[-2], cout))
    out = torch.zeros(n_out, cout, dtype=feats.dtype)
    for c in range(8):
        contrib = G[nbr, cell[:, c]] * (imp * w[:, c]).unsqueeze(-1)
        out.index_add_(0, rows, contrib)
    return out + bias


def particle_net_forward(state, pos, vel, box, box_feats, gravity=None, dt=1 / 50, extent=FILTER_EXTENT,
                         return_debug=False, feats=None):
    """ParticleNet.forward (transmodel.py:151-163) -> (pos'', vel'', num_fluid_neighbors).  feats: the optional per-particle
    features of other_feats_channels > 0, appended to [1, v] (:111-114)."""
    g = state["gravity"] if gravity is None else gravity
    pos_new, vel_new = integrate_pos_vel(pos, vel, g, dt)
    radius = 0.5 * extent
    f_idx, f_rs, f_d2 = radius_search(pos_new, pos_new, radius, True)
    b_idx, b_rs, b_d2 = radius_search(box, pos_new, radius, True)
    fluid_feats = torch.cat([torch.ones_like(pos_new[:, 0:1]), vel_new] + ([feats] if feats is not None else []), -1)

    def conv(name, feats, inp_pos, idx, rs, d2):
        return cconv(feats, inp_pos, pos_new, extent, state[f"{name}.kernel"], state[f"{name}.bias"], idx, rs, d2)

    def dense(name, x):
        return torch.nn.functional.linear(x, state[f"{name}.weight"], state[f"{name}.bias"])

    a_fluid = conv("conv0_fluid", fluid_feats, pos_new, f_idx, f_rs, f_d2)
    a_dense = dense("dense0_fluid", fluid_feats)
    a_obst = conv("conv0_obstacle", box_feats, box, b_idx, b_rs, b_d2)
    ans = [torch.cat([a_obst, a_fluid, a_dense], -1)]
    for i in (1, 2, 3):
        x = torch.relu(ans[-1])
        y = conv(f"conv{i}", x, pos_new, f_idx, f_rs, f_d2) + dense(f"dense{i}", x)
        if y.shape[-1] == ans[-1].shape[-1]:
            y = y + ans[-1]
        ans.append(y)
    num_nbrs = (f_rs[1:] - f_rs[:-1]).to(torch.float32)
    delta = (1.0 / 128) * ans[-1]
    pos_c, vel_c = update_pos_vel(pos, pos_new, delta, dt)
    if return_debug:
        return pos_c, vel_c, num_nbrs, dict(ans=ans, f_idx=f_idx, f_rs=f_rs, f_d2=f_d2, b_idx=b_idx, b_rs=b_rs,
                                            b_d2=b_d2, pos_new=pos_new, vel_new=vel_new)
    return pos_c, vel_c, num_nbrs


# --------------------------------------------------------------------------------------------
def conv_shapes(other_feats_channels=0):
    shapes = {"conv0_fluid": (4 + other_feats_channels, 32), "conv0_obstacle": (3, 32)}
    dense = {"dense0_fluid": (4 + other_feats_channels, 32)}
    for i in range(1, 4):
        cin = LAYER_CHANNELS[i - 1] * (3 if i == 1 else 1)
        shapes[f"conv{i}"] = (cin, LAYER_CHANNELS[i])
        dense[f"dense{i}"] = (cin, LAYER_CHANNELS[i])
    return shapes, dense


def deterministic_transition_state(gravity=(0.0, 0.0, -9.81), other_feats_channels=0):
    convs, denses = conv_shapes(other_feats_channels)
    st = {"gravity": torch.tensor(gravity, dtype=torch.float32)}
    for l, (name, (ci, co)) in enumerate(convs.items()):
        n = KSIZE ** 3 * ci * co
        k = torch.arange(n, dtype=torch.float64)
        st[f"{name}.kernel"] = (0.05 * torch.sin(0.61 * k + 0.9 * l)).float().view(KSIZE, KSIZE, KSIZE, ci, co)
        st[f"{name}.bias"] = (0.01 * torch.cos(0.3 * torch.arange(co, dtype=torch.float64) + l)).float()
        st[f"{name}.offset"] = torch.zeros(3)
    for l, (name, (ci, co)) in enumerate(denses.items()):
        k = torch.arange(ci * co, dtype=torch.float64).view(co, ci)
        st[f"{name}.weight"] = (math.sqrt(1.5 / ci) * torch.sin(0.43 * k + 0.7 * l)).float()
        st[f"{name}.bias"] = (0.01 * torch.sin(0.2 * torch.arange(co, dtype=torch.float64) + l)).float()
    return st


def watercube_box(spacing=0.05):
    """Synthetic container: the 6 faces of x,y in [-1,1], z in [-1,2.4552]
    (trainer/basetrainer.py:58-62) sampled on a 0.05 grid, inward normals."""
    lo = np.array([-1.0, -1.0, -1.0]); hi = np.array([1.0, 1.0, 2.4552])
    pts, nrm = [], []
    for ax in range(3):
        o = [a for a in range(3) if a != ax]
        u = np.arange(lo[o[0]], hi[o[0]] + 1e-6, spacing)
        v = np.arange(lo[o[1]], hi[o[1]] + 1e-6, spacing)
        uu, vv = np.meshgrid(u, v, indexing="ij")
        for side, val in ((0, lo[ax]), (1, hi[ax])):
            p = np.zeros((uu.size, 3)); p[:, o[0]] = uu.ravel(); p[:, o[1]] = vv.ravel(); p[:, ax] = val
            n = np.zeros((uu.size, 3)); n[:, ax] = 1.0 if side == 0 else -1.0
            pts.append(p); nrm.append(n)
    return (torch.from_numpy(np.concatenate(pts).astype(np.float32)),
            torch.from_numpy(np.concatenate(nrm).astype(np.float32)))
