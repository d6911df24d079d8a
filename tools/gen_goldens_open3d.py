#!/usr/bin/env python3
"""Pin-upgrade script for the Open3D half of the hot path.

Run on ANY machine that has `open3d` (0.15.2, the version /root/reference/README.md:32-33 pins; built with the ML
torch ops) and torch — CPU is enough:

    python tools/gen_goldens_open3d.py            # writes tests/golden/thirdparty/open3d_*.npz

It calls the REAL `open3d.ml.torch.layers.ContinuousConv` (constructed with exactly the arguments of
/root/reference/models/transmodel.py:86-95), its internal `FixedRadiusSearch` (read back through `.nns`, as
:136-138 does) and `ops.reduce_subarrays_sum`, on seeded synthetic inputs of the shapes the model uses, and stores
(inputs, kernel, bias, outputs) triples.  `tests/test_oracle_thirdparty.py` then checks oracle/trans_oracle.py and
oracle/csrc/nf_oracle.c against those files; with them committed the "parity unpinned" status of rows B2/B4-B8 is
lifted without touching product code.  Nothing of Open3D is copied: the files hold numeric arrays only.

This container has no open3d (no network): the script has NOT been run here, and the fixtures are absent until
somebody runs it — the test reports that as "unpinned" (skip with reason).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "thirdparty")
FILTER_EXTENT = float(np.float32(6 * 1.5 * 0.025))          # models/transmodel.py:35


def window_poly6(r_sqr):
    import torch
    return torch.clamp((1 - r_sqr) ** 3, 0, 1)               # models/transmodel.py:73-77


def lattice(n_side, seed, spacing=0.05, jitter=0.005):
    ax = spacing * np.arange(n_side)
    g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    return (g + np.random.RandomState(seed).uniform(-jitter, jitter, g.shape)).astype(np.float32)


def plane(n, z, seed, spacing=0.05):
    ax = spacing * np.arange(-2, n + 2)
    g = np.stack(np.meshgrid(ax, ax, indexing="ij"), -1).reshape(-1, 2)
    p = np.concatenate([g, np.full((g.shape[0], 1), z)], 1)
    nrm = np.tile(np.array([[0, 0, 1.0]]), (p.shape[0], 1))
    return p.astype(np.float32), nrm.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=OUT)
    args = ap.parse_args()
    try:
        import torch
        import open3d as o3d
        import open3d.ml.torch as ml3d
    except Exception as e:          # noqa: BLE001
        sys.exit(f"open3d with the ML torch ops is required to generate these fixtures: {e}")
    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(0)
    fluid = lattice(9, 1)                                   # 729 particles, ~40 neighbours in the interior
    wall, wall_n = plane(9, -0.045, 2)                      # a container face just below the block
    far = np.array([[3.0, 3.0, 3.0]], np.float32)           # an isolated particle: empty neighbourhood
    fluid = np.concatenate([fluid, far, fluid[:1] + 0.0])   # ... and a duplicate of particle 0 (identical position)
    rng = np.random.RandomState(3)
    cases = {
        # name: (Cin, Cout, input positions, input features)        — the five convs of the model + one odd shape
        "conv0_fluid": (4, 32, fluid, np.concatenate([np.ones((fluid.shape[0], 1), np.float32),
                                                      rng.normal(0, 0.3, (fluid.shape[0], 3)).astype(np.float32)], 1)),
        "conv0_obstacle": (3, 32, wall, wall_n),
        "conv1": (96, 64, fluid, rng.normal(0, 1, (fluid.shape[0], 96)).astype(np.float32)),
        "conv2": (64, 64, fluid, rng.normal(0, 1, (fluid.shape[0], 64)).astype(np.float32)),
        "conv3": (64, 3, fluid, rng.normal(0, 1, (fluid.shape[0], 64)).astype(np.float32)),
        "odd_5_7": (5, 7, fluid, rng.normal(0, 1, (fluid.shape[0], 5)).astype(np.float32)),
    }
    for name, (cin, cout, inp_pos, feats) in cases.items():
        conv = ml3d.layers.ContinuousConv(kernel_size=[4, 4, 4], activation=None, interpolation='linear',
                                          coordinate_mapping='ball_to_cube_volume_preserving', normalize=False,
                                          window_function=window_poly6, radius_search_ignore_query_points=True,
                                          in_channels=cin, filters=cout)
        with torch.no_grad():
            conv.kernel.copy_(torch.from_numpy(rng.normal(0, 0.1, tuple(conv.kernel.shape)).astype(np.float32)))
            conv.bias.copy_(torch.from_numpy(rng.normal(0, 0.1, (cout,)).astype(np.float32)))
        x = torch.from_numpy(feats).requires_grad_(True)
        conv.kernel.requires_grad_(True)
        out = conv(x, torch.from_numpy(inp_pos), torch.from_numpy(fluid), FILTER_EXTENT)
        g_out = torch.from_numpy(rng.normal(0, 1, tuple(out.shape)).astype(np.float32))
        (out * g_out).sum().backward()
        nns = conv.nns
        counts = ml3d.ops.reduce_subarrays_sum(torch.ones_like(nns.neighbors_index, dtype=torch.float32),
                                               nns.neighbors_row_splits)              # models/transmodel.py:135-138
        state = {k: v.detach().numpy() for k, v in conv.state_dict().items()}
        np.savez_compressed(
            os.path.join(args.out, f"open3d_{name}.npz"),
            open3d_version=o3d.__version__, extent=FILTER_EXTENT, inp_positions=inp_pos, out_positions=fluid,
            inp_features=feats, kernel=conv.kernel.detach().numpy(), bias=conv.bias.detach().numpy(),
            output=out.detach().numpy(), grad_output=g_out.numpy(), grad_kernel=conv.kernel.grad.numpy(),
            grad_features=x.grad.numpy(),
            neighbors_index=nns.neighbors_index.numpy(), neighbors_row_splits=nns.neighbors_row_splits.numpy(),
            neighbors_distance=nns.neighbors_distance.numpy(), neighbor_counts=counts.numpy(),
            state_dict_keys=np.array(sorted(state)), **{f"state__{k}": v for k, v in state.items()})
        print(f"{name}: out {tuple(out.shape)}, nnz {nns.neighbors_index.shape[0]}")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
