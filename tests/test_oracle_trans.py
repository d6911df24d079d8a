"""Known-answer / property tests that pin the transition-model ORACLE itself (CPU).  Open3D cannot be
imported here, so these analytic properties are what the ContinuousConv restatement rests on."""
import math

import numpy as np
import torch

from oracle import trans_oracle as to


def test_mapping_is_volume_preserving():
    """ball_to_cube_volume_preserving: |det J| must equal vol(cube)/vol(ball) = 8 / (4 pi / 3) everywhere."""
    g = torch.Generator().manual_seed(0)
    p = torch.randn(4000, 3, generator=g, dtype=torch.float64)
    p = p / p.norm(dim=1, keepdim=True) * torch.rand(4000, 1, generator=g, dtype=torch.float64) ** (1 / 3) * 0.97
    f = lambda q: to.map_cylinder_to_cube(to.map_sphere_to_cylinder(q))
    eps = 1e-6
    J = torch.stack([(f(p + eps * e) - f(p - eps * e)) / (2 * eps) for e in torch.eye(3, dtype=torch.float64)], -1)
    det = torch.linalg.det(J).abs()
    # drop samples next to the piecewise seams where the finite difference straddles two branches
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    seam = ((1.25 * z * z - (x * x + y * y)).abs() < 1e-3) | ((x.abs() - y.abs()).abs() < 1e-3)
    det = det[~seam]
    assert det.numel() > 3000
    torch.testing.assert_close(det, torch.full_like(det, 6 / math.pi), rtol=1e-4, atol=1e-4)


def test_mapping_range_and_axes():
    pts = torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [-1.0, 0, 0], [0, 0, -1.0], [0, 0, 0]])
    out = to.map_cylinder_to_cube(to.map_sphere_to_cylinder(pts))
    torch.testing.assert_close(out, pts)           # axis points and the centre are fixed points
    g = torch.Generator().manual_seed(1)
    s = torch.randn(2000, 3, generator=g)
    s = s / s.norm(dim=1, keepdim=True)            # unit sphere -> cube surface
    c = to.map_cylinder_to_cube(to.map_sphere_to_cylinder(s))
    assert float(c.abs().max()) <= 1 + 1e-5
    torch.testing.assert_close(c.abs().max(dim=1)[0], torch.ones(2000), rtol=0, atol=1e-5)


def test_trilinear_partition_of_unity_and_clamp():
    g = torch.Generator().manual_seed(2)
    c = torch.rand(500, 3, generator=g) * 3
    c[0] = torch.tensor([3.0, 0.0, 1.5])
    cell, w = to.trilinear(c)
    torch.testing.assert_close(w.sum(1), torch.ones(500))
    assert int(cell.min()) >= 0 and int(cell.max()) <= 63
    # coordinate reconstruction: sum w * cell coordinate == c
    xs = (cell % 4).float(); ys = ((cell // 4) % 4).float(); zs = (cell // 16).float()
    torch.testing.assert_close(torch.stack([(w * xs).sum(1), (w * ys).sum(1), (w * zs).sum(1)], 1), c, rtol=1e-5, atol=1e-5)


def test_cconv_single_neighbour_on_axis():
    """One neighbour at distance d on +x: out = window(d^2/r^2) * trilinear mix of kernel cells * feat + bias."""
    extent = to.FILTER_EXTENT
    r = extent / 2
    d = 0.4 * r
    out_pos = torch.zeros(1, 3)
    inp_pos = torch.tensor([[d, 0.0, 0.0]])
    feats = torch.tensor([[2.0, -1.0]])
    g = torch.Generator().manual_seed(3)
    kernel = torch.randn(4, 4, 4, 2, 5, generator=g)
    bias = torch.randn(5, generator=g)
    idx, rs, d2 = to.radius_search(inp_pos, out_pos, r, True)
    out = to.cconv(feats, inp_pos, out_pos, extent, kernel, bias, idx, rs, d2)
    imp = max(0.0, min(1.0, (1 - (d * d) / (r * r)) ** 3))
    cx = (0.4 + 1) * 1.5               # on-axis points are fixed points of the mapping
    x0, fx = int(math.floor(cx)), cx - math.floor(cx)
    exp = torch.zeros(5)
    for dz in (1, 2):
        for dy in (1, 2):              # y = z = 1.5 -> cells 1 and 2 with weight 0.5 each
            for xi, wx in ((x0, 1 - fx), (x0 + 1, fx)):
                exp += 0.25 * wx * (feats[0] @ kernel[dz, dy, xi])
    torch.testing.assert_close(out[0], imp * exp + bias, rtol=1e-5, atol=1e-5)


def test_cconv_rotation_equivariance():
    """Rotating all positions by 90 deg about z equals permuting/flipping the kernel's (x,y) axes."""
    g = torch.Generator().manual_seed(4)
    P = torch.rand(60, 3, generator=g) * 0.3
    feats = torch.randn(60, 3, generator=g)
    kernel = torch.randn(4, 4, 4, 3, 4, generator=g)
    bias = torch.zeros(4)
    extent = to.FILTER_EXTENT
    idx, rs, d2 = to.radius_search(P, P, extent / 2, True)
    base = to.cconv(feats, P, P, extent, kernel, bias, idx, rs, d2)
    R = torch.tensor([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])   # (x,y) -> (-y,x)
    Pr = P @ R.T
    idx2, rs2, d22 = to.radius_search(Pr, Pr, extent / 2, True)
    # new x' = -y, y' = x  ->  kernel'[z][y'][x'] = kernel[z][y = 3 - x'][x = y']
    kr = kernel.permute(0, 2, 1, 3, 4).flip(2)
    rot = to.cconv(feats, Pr, Pr, extent, kr, bias, idx2, rs2, d22)
    torch.testing.assert_close(rot, base, rtol=1e-4, atol=1e-5)


def test_radius_search_contract():
    P = torch.tensor([[0.0, 0, 0], [0.05, 0, 0], [0.1125, 0, 0], [0.2, 0, 0], [0.0, 0, 0]])
    idx, rs, d2 = to.radius_search(P, P[:1], 0.1125, True)
    # identical positions (points 0 and 4) are skipped; the point exactly on the radius is included (<=)
    assert idx.tolist() == [1, 2]
    idx, rs, d2 = to.radius_search(P, P[:1], 0.1125, False)
    assert idx.tolist() == [0, 1, 2, 4]


def test_particle_net_shapes_and_rest_state():
    from oracle import render_oracle as ro
    P = ro.watercube_particles()[:800]
    box, bn = to.watercube_box()
    st = to.deterministic_transition_state()
    p, v, n = to.particle_net_forward(st, P, torch.zeros_like(P), box, bn)
    assert p.shape == P.shape and v.shape == P.shape and n.shape == (800,)
    # with zero conv/dense weights the step is pure ballistic integration
    z = {k: torch.zeros_like(t) if k != "gravity" else t for k, t in st.items()}
    p0, v0, _ = to.particle_net_forward(z, P, torch.zeros_like(P), box, bn)
    dt = 1 / 50
    # update_pos_vel re-derives the velocity from the displacement: (p'' - p)/dt = g*dt/2  (transmodel.py:144-148)
    torch.testing.assert_close(v0, (0.5 * torch.tensor([0, 0, -9.81]) * dt).expand_as(P), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(p0, P + 0.5 * torch.tensor([0, 0, -9.81]) * dt * dt, rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------
# Convention known-answer tests (tests/kat_conventions.py): expected values come from the operators' published contracts,
# not from this oracle — the same cases run against the HIP path in tests/test_gpu_trans.py / test_gpu_render.py.
# ------------------------------------------------------------------------------------------------
def test_kat_filter_axis_order_and_sign():
    import kat_conventions as kat
    out_pos = torch.tensor([kat.OUT_POS])
    for off, (kz, ky, kx), want in kat.axis_cases():
        inp_pos = out_pos + torch.tensor([off])
        kernel = torch.zeros(4, 4, 4, 1, 1)
        kernel[kz, ky, kx, 0, 0] = 1.0
        idx, rs, d2 = to.radius_search(inp_pos, out_pos, kat.RADIUS, True)
        assert idx.tolist() == [0]
        out = to.cconv(torch.ones(1, 1), inp_pos, out_pos, kat.EXTENT, kernel, torch.zeros(1), idx, rs, d2)
        assert abs(float(out[0, 0]) - want) <= 2e-6, (off, (kz, ky, kx), float(out[0, 0]), want)


def test_kat_ball_to_cube_closed_forms():
    import kat_conventions as kat
    for p, cube in kat.MAPPING_CASES:
        got = to.map_cylinder_to_cube(to.map_sphere_to_cylinder(torch.tensor([p], dtype=torch.float64)))[0]
        assert max(abs(float(g) - c) for g, c in zip(got, cube)) <= 1e-8, (p, got.tolist(), cube)
        rel = torch.tensor([p], dtype=torch.float32) * (kat.EXTENT / 2)
        coords = to.filter_coordinates(rel, kat.EXTENT)[0]
        assert max(abs(float(g) - c) for g, c in zip(coords, kat.filter_coordinate(cube))) <= 2e-6


def test_kat_radius_inclusivity():
    import kat_conventions as kat
    from oracle import neighbors
    pts = kat.radius_points()
    q = pts[:1]
    idx, rs, d2 = neighbors.fixed_radius_search(pts, q, kat.RADIUS, True)
    assert idx.tolist() == kat.FIXED_RADIUS_EXPECTED_IGNORE and float(d2[1]) == kat.RADIUS ** 2
    idx, rs, d2 = neighbors.fixed_radius_search(pts, q, kat.RADIUS, False)
    assert idx.tolist() == kat.FIXED_RADIUS_EXPECTED_KEEP
    d, i, nn = neighbors.ball_query_firstk(q, pts, kat.RADIUS, 5)
    assert i[0].tolist() == kat.BALL_QUERY_EXPECTED + [-1, -1]
