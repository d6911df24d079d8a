"""Synthetic "watercube 400^2" scene and closed-form weights used by bench.py, tools/ and smoke runs.
Host-side data generation only (numpy / torch CPU); nothing here is on the timed path.  tests/test_host_logic.py
checks that these definitions coincide with the oracle's."""
import math

import numpy as np
import torch


def watercube_particles(n_side=17, spacing=0.05, corner=(-0.40, -0.40, -0.975), jitter=0.005, seed=10):
    """17^3 lattice (2 x particle_radius spacing), index order x-slowest, + RandomState(10) jitter."""
    ax = [corner[d] + spacing * np.arange(n_side) for d in range(3)]
    g = np.stack(np.meshgrid(ax[0], ax[1], ax[2], indexing="ij"), -1).reshape(-1, 3)
    g = g + np.random.RandomState(seed).uniform(-jitter, jitter, g.shape)
    return torch.from_numpy(g.astype(np.float32))


def shaped_particles(kind, spacing=0.05, jitter=0.005, seed=10, order="random", centre=(0.0, 0.0, -0.25)):
    """Synthetic stand-ins for the fluid bodies of BASELINE configs 4 / 5 (the released data sets are not available):
    a jittered 0.05 lattice clipped to the shape, in an index order that is NOT lattice order.
      kind "bunny":     union of ellipsoids (body, head, two ears, tail)  — 4 774 particles
      kind "honeycone": a cone standing on its apex, top radius 0.65, height 1.3 — 4 350 particles
      kind "cube":      the watercube block (for comparison with the same code path)
      order "random":   a seeded permutation — no spatial coherence of the index at all: the hardest regime of the
                        first-K-by-index search (the K lowest indices in radius are scattered over the whole ball)
      order "scan":     lattice scan order restricted to the shape (what a volume sampler emits)
      order "shells":   sorted by distance from the shape's centre (an emitter-like order)"""
    c = np.asarray(centre, np.float64)
    ax = [np.arange(-1.0, 1.0 + 1e-9, spacing) for _ in range(3)]
    g = np.stack(np.meshgrid(ax[0], ax[1], ax[2], indexing="ij"), -1).reshape(-1, 3)

    def ell(p, ctr, rad):
        return (((p - np.asarray(ctr)) / np.asarray(rad)) ** 2).sum(-1) <= 1.0

    if kind == "bunny":
        inside = (ell(g, (0, 0, 0), (0.60, 0.43, 0.47)) | ell(g, (0.52, 0, 0.43), (0.29, 0.25, 0.26)) |
                  ell(g, (0.58, 0.13, 0.80), (0.09, 0.07, 0.29)) | ell(g, (0.58, -0.13, 0.80), (0.09, 0.07, 0.29)) |
                  ell(g, (-0.63, 0, 0.07), (0.13, 0.13, 0.13)))
    elif kind == "honeycone":
        h = g[:, 2] + 0.65                                   # apex at z = -0.65, top at z = +0.65
        inside = (h >= 0) & (h <= 1.3) & (np.hypot(g[:, 0], g[:, 1]) <= 0.5 * h + 1e-9)
    elif kind == "cube":
        inside = (np.abs(g) <= 0.4 + 1e-9).all(-1)
    else:
        raise ValueError(kind)
    p = g[inside]
    rng = np.random.RandomState(seed)
    p = p + rng.uniform(-jitter, jitter, p.shape)
    if order == "random":
        p = p[rng.permutation(p.shape[0])]
    elif order == "shells":
        p = p[np.argsort((p ** 2).sum(-1), kind="stable")]
    elif order != "scan":
        raise ValueError(order)
    return torch.from_numpy((p + c).astype(np.float32))


def watercube_box(spacing=0.05):
    """The 6 faces of x,y in [-1,1], z in [-1,2.4552] (trainer/basetrainer.py:58-62) on a 0.05 grid, inward normals."""
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


def eval_camera():
    """c2w (3,4): the pose values of the reference's evaluation camera (eval_renderer.py:67-92)."""
    return torch.tensor([
        [0.3597943186759949, 0.09052024036645889, -0.18696719408035278, -4.842308521270752],
        [-0.2077273577451706, 0.15678563714027405, -0.32383665442466736, -8.387124061584473],
        [0.0, 0.37393447756767273, 0.181040421128273, 4.688809871673584]], dtype=torch.float32)


def camera_focal(W, camera_angle_x=0.323):
    return 0.5 * W / math.tan(0.5 * camera_angle_x)


def nerf_layer_shapes(cx=198, cd=54, W=256):
    shapes = {}
    for i in range(8):
        shapes[f"xyz_encoding_{i + 1}.0"] = (W, cx if i == 0 else (W + cx if i == 4 else W))
    shapes["xyz_encoding_final"] = (W, W)
    shapes["dir_encoding.0"] = (W // 2, W + cd)
    shapes["sigma"] = (1, W)
    shapes["rgb.0"] = (3, W // 2)
    return shapes


def deterministic_nerf_state(prefixes=("nerf_coarse", "nerf_fine"), cx=198, cd=54):
    """w[o,i] = sqrt(2.4/fan_in) sin(0.37 (o fan_in + i) + 1.3 layer + 0.5 net); small cosine biases, sigma bias + 2."""
    state = {}
    for n, prefix in enumerate(prefixes):
        for l, (name, (o, i)) in enumerate(nerf_layer_shapes(cx, cd).items()):
            k = torch.arange(o * i, dtype=torch.float64).view(o, i)
            w = math.sqrt(2.4 / i) * torch.sin(0.37 * k + 1.3 * l + 0.5 * n)
            b = 0.05 * torch.cos(0.11 * torch.arange(o, dtype=torch.float64) + l + n)
            if name == "sigma":
                b = b + 2.0
            state[f"{prefix}.{name}.weight"] = w.float()
            state[f"{prefix}.{name}.bias"] = b.float()
    return state


def deterministic_transition_state(gravity=(0.0, 0.0, -9.81)):
    convs = {"conv0_fluid": (4, 32), "conv0_obstacle": (3, 32), "conv1": (96, 64), "conv2": (64, 64), "conv3": (64, 3)}
    denses = {"dense0_fluid": (4, 32), "dense1": (96, 64), "dense2": (64, 64), "dense3": (64, 3)}
    st = {"gravity": torch.tensor(gravity, dtype=torch.float32)}
    for l, (name, (ci, co)) in enumerate(convs.items()):
        k = torch.arange(64 * ci * co, dtype=torch.float64)
        st[f"{name}.kernel"] = (0.05 * torch.sin(0.61 * k + 0.9 * l)).float().view(4, 4, 4, ci, co)
        st[f"{name}.bias"] = (0.01 * torch.cos(0.3 * torch.arange(co, dtype=torch.float64) + l)).float()
        st[f"{name}.offset"] = torch.zeros(3)
    for l, (name, (ci, co)) in enumerate(denses.items()):
        k = torch.arange(ci * co, dtype=torch.float64).view(co, ci)
        st[f"{name}.weight"] = (math.sqrt(1.5 / ci) * torch.sin(0.43 * k + 0.7 * l)).float()
        st[f"{name}.bias"] = (0.01 * torch.sin(0.2 * torch.arange(co, dtype=torch.float64) + l)).float()
    return st


def watercube_scene(H=400, W=400):
    """Everything bench.py needs, on the CPU."""
    from . import ray_utils
    c2w = eval_camera()
    rays = ray_utils.get_rays_cpu(H, W, camera_focal(W), c2w).view(-1, 6)
    box, bn = watercube_box()
    return dict(P=watercube_particles(), c2w=c2w, rays=rays, box=box, bn=bn, nerf_state=deterministic_nerf_state(),
                trans_state=deterministic_transition_state())
