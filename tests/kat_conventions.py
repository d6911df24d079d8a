"""Known-answer cases that fix the CONVENTIONS of the two third-party operators on the hot path — written from the
operators' published contracts, NOT from oracle/ or the HIP kernels, and run against BOTH (tests/test_oracle_trans.py on
the CPU, tests/test_gpu_trans.py / test_gpu_render.py on the device).  A convention error shared by the oracle and the HIP
code (both restated by the same hand: transposed filter axes, a flipped branch of the ball -> cube map, `<` vs `<=` at the
radius, neighbour-minus-centre vs centre-minus-neighbour) passes every HIP-vs-oracle test; it cannot pass these.

Sources of the expected values
  * Open3D ml3d.layers.ContinuousConv (0.15; call sites /root/reference/models/transmodel.py:86-95, :116-118, :125):
      - relative position = inp_position - out_position (neighbour minus query), scaled by 2 / extent into the unit ball;
      - coordinate_mapping "ball_to_cube_volume_preserving" (Griepentrog, Hoehne, Ummenhofer): ball -> cylinder
            5/4 z^2 > x^2 + y^2 (cap):  (x, y) *= sqrt(3 n / (n + |z|)),  z = sign(z) n          n = |p|
            otherwise (side):           (x, y) *= n / sqrt(x^2 + y^2),    z *= 3/2
        then cylinder -> cube on (x, y):
            |y| <= |x|:  t = sign(x) sqrt(x^2 + y^2),  (x, y) = (t, t 4/pi atan(y / x));   else the same with x <-> y
      - align_corners=True: cube coordinate c in [-1, 1] -> filter coordinate (c + 1) / 2 * (size - 1) in [0, 3];
        interpolation "linear" = trilinear between the 8 surrounding filter nodes;
      - the filter tensor is (kernel_size[0], kernel_size[1], kernel_size[2], Cin, Cout) = (depth, height, width, ...): the
        LAST spatial dim is indexed by world x, the first by world z  ->  kernel[z][y][x];
      - window_function is called on d^2 / radius^2, radius = extent / 2; normalize=False: plain sum over neighbours; + bias.
  * Open3D FixedRadiusSearch (metric L2): neighbour <=> d^2 <= radius^2 (fp32 squared distance); with
    ignore_query_point=True a point at the IDENTICAL position as the query is skipped.
  * pytorch3d.ops.ball_query (0.6.1; call site models/renderer.py:116-118): neighbour <=> d^2 < radius^2 (STRICT), first K
    in index order, padding idx -1 / dists 0.
"""
import math

import numpy as np

EXTENT = 0.25                    # radius 0.125: exactly representable, radius^2 = 2^-6 too
RADIUS = 0.125
OUT_POS = (0.3, -0.2, 0.1)       # the query is NOT at the origin: a kernel that forgot the subtraction fails


def poly6(r2_over_radius2):
    return max(0.0, min(1.0, (1.0 - r2_over_radius2) ** 3))


# ---- (i) axis order and sign: ONE neighbour at half the radius along +/- one world axis, Cin = Cout = 1, feature 1,
# bias 0, a filter that is 1 in exactly one node [kz][ky][kx].  Along +x the unit-ball point is (0.5, 0, 0): a fixed point
# of the map (side branch: n / sqrt(x^2) = 1, z * 3/2 = 0; cylinder -> cube: atan(0) = 0), filter coordinate
# ((0.5 + 1) * 1.5, 1.5, 1.5) = (2.25, 1.5, 1.5): trilinear weights 0.75 / 0.25 on x = 2 / 3, 0.5 / 0.5 on y = 1 / 2 and on
# z = 1 / 2.  Window: poly6(0.25) = 0.75^3 = 0.421875.  Along -x: coordinate 0.75 -> weights 0.25 / 0.75 on x = 0 / 1.
def axis_cases():
    """-> list of (neighbour offset (3,), (kz, ky, kx), expected output)."""
    w = poly6(0.25)
    half = 0.5 * RADIUS
    cases = []
    for axis, name in enumerate("xyz"):
        for sign in (+1, -1):
            off = [0.0, 0.0, 0.0]
            off[axis] = sign * half
            near, far_ = (2, 3) if sign > 0 else (1, 0)          # node that gets 0.75, node that gets 0.25
            for node, wt in ((near, 0.75), (far_, 0.25)):
                k = [1, 1, 1]                                    # the other two axes sit between nodes 1 and 2 (0.5 each)
                k[axis] = node
                kx, ky, kz = k                                   # world (x, y, z) node indices
                cases.append((tuple(off), (kz, ky, kx), w * wt * 0.25))
                # the same node index put on a DIFFERENT filter axis must give 0 when that axis' coordinate is 1.5
                if node in (0, 3):
                    other = (axis + 1) % 3
                    k2 = [1, 1, 1]
                    k2[other] = node
                    cases.append((tuple(off), (k2[2], k2[1], k2[0]), 0.0))
    return cases


# ---- (ii) closed-form values of ball_to_cube_volume_preserving (unit ball -> cube [-1, 1]^3), by hand:
#  a) axis points are fixed points.
#  b) on the cap / side seam 5/4 z^2 = x^2 + y^2 with p = (sqrt(0.2), 0, 0.4): n = sqrt(0.36) = 0.6; the test is STRICT (>), so
#     the side branch: (x, y) *= 0.6 / sqrt(0.2) -> x = 0.6, z * 1.5 = 0.6 (the cap branch gives the same point: the map is
#     continuous across the seam) -> cylinder (0.6, 0, 0.6) -> cube (0.6, 0, 0.6).
#  c) sphere point on the cube diagonal (1, 1, 1) / sqrt(3): 5/4 * 1/3 < 2/3 -> side: (x, y) *= 1 / sqrt(2/3) -> (x, y) =
#     (1, 1) / sqrt(2), z = sqrt(3) / 2; cylinder -> cube: |y| <= |x| -> t = 1, y = 4/pi * atan(1) = 1 -> (1, 1, sqrt(3)/2).
#  d) cap region, p = (0, 0.3, 0.6): n = sqrt(0.45); s = sqrt(3 n / (n + 0.6)) = 1.2584086; -> (0, 0.3775226, 0.6708204);
#     cylinder -> cube: |y| > |x|: t = 0.3775226, x = t 4/pi atan(0) = 0 -> (0, 0.3775226, 0.6708204).
#  e) side region, p = (0.6, 0.2, 0.1): n = sqrt(0.41) = 0.6403124; (x, y) scaled to radius n; z = 0.15; cylinder -> cube:
#     t = n, y = n 4/pi atan(1/3) = 0.6403124 * 1.2732395 * 0.3217506 = 0.2623139 -> (0.6403124, 0.2623139, 0.15).
#  f) mixed signs, p = (-0.2, 0.5, -0.3): side; n = sqrt(0.38) = 0.6164414; z = -0.45; |y| > |x|: t = +n,
#     x = n 4/pi atan(-0.4) = -0.2986509 -> (-0.2986509, 0.6164414, -0.45).
MAPPING_CASES = [
    ((1.0, 0.0, 0.0), (1.0, 0.0, 0.0)),
    ((0.0, 0.0, 1.0), (0.0, 0.0, 1.0)),
    ((0.0, -1.0, 0.0), (0.0, -1.0, 0.0)),
    ((math.sqrt(0.2), 0.0, 0.4), (0.6, 0.0, 0.6)),
    ((1 / math.sqrt(3),) * 3, (1.0, 1.0, math.sqrt(3) / 2)),
    ((0.0, 0.3, 0.6), (0.0, 0.377522572, 0.670820393)),
    ((0.6, 0.2, 0.1), (0.640312424, 0.262313928, 0.15)),
    ((-0.2, 0.5, -0.3), (-0.29865092, 0.6164414, -0.45)),
]


def filter_coordinate(cube):
    """align_corners=True, size 4."""
    return tuple((c + 1.0) * 1.5 for c in cube)


# ---- (iii) / (iv) radius inclusivity in fp32.  Query at the origin, points on the +x axis: index 0 identical to the
# query, 1 well inside, 2 at EXACTLY the radius (d^2 = 2^-6 = radius^2 exactly), 3 one ulp beyond, 4 one ulp inside.
def radius_points():
    r = np.float32(RADIUS)
    xs = [np.float32(0.0), np.float32(0.05), r, np.nextafter(r, np.float32(1.0)), np.nextafter(r, np.float32(0.0))]
    pts = np.zeros((5, 3), np.float32)
    pts[:, 0] = xs
    return pts


FIXED_RADIUS_EXPECTED_IGNORE = [1, 2, 4]          # <= radius^2, the identical point skipped
FIXED_RADIUS_EXPECTED_KEEP = [0, 1, 2, 4]
BALL_QUERY_EXPECTED = [0, 1, 4]                   # STRICT <: the point at exactly the radius is out; d^2 = 0 is a hit
