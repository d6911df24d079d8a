#!/usr/bin/env python3
"""Pin-upgrade script for `pytorch3d.ops.ball_query` (v0.6.1, /root/reference/README.md:41-42; call site
/root/reference/models/renderer.py:116-118).  Run on any machine with pytorch3d + torch (CPU is enough):

    python tools/gen_goldens_pytorch3d.py         # writes tests/golden/thirdparty/pytorch3d_ball_query.npz

Stores (p1, p2, radius, K) -> (dists, idx, nn) of the REAL op for the regimes the renderer meets: more than K points in
radius (first-K-by-index cut), fewer than K (padding -1 / 0 / 0), no point at all, queries that coincide with points
(d2 == 0 slots, which the renderer's `dists != 0` mask treats as empty), points in a shuffled index order, K larger than
the cloud, and a batch of 2 clouds.  tests/test_oracle_thirdparty.py checks oracle/csrc/nf_oracle.c against it.
Not run in this container (pytorch3d is not installable here): until the file exists the test skips with "unpinned"."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "thirdparty")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=OUT)
    args = ap.parse_args()
    try:
        import torch
        import pytorch3d
        from pytorch3d.ops import ball_query
    except Exception as e:          # noqa: BLE001
        sys.exit(f"pytorch3d is required to generate this fixture: {e}")
    os.makedirs(args.out, exist_ok=True)
    rng = np.random.RandomState(7)
    ax = 0.05 * np.arange(13)
    cloud = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3) + rng.uniform(-0.005, 0.005, (13 ** 3, 3))
    cloud = cloud.astype(np.float32)
    shuffled = cloud[rng.permutation(cloud.shape[0])]
    q = np.concatenate([cloud[rng.randint(0, cloud.shape[0], 600)] + rng.normal(0, 0.08, (600, 3)),
                        rng.uniform(-1, 2, (200, 3)), cloud[:40]]).astype(np.float32)
    out = {"pytorch3d_version": pytorch3d.__version__}
    cases = {"lattice_r0.225_k20": (q, cloud, 0.225, 20), "shuffled_r0.225_k20": (q, shuffled, 0.225, 20),
             "small_r0.1_k8": (q, cloud[:300], 0.1, 8), "k_gt_cloud": (q[:50], cloud[:5], 0.5, 20)}
    for name, (p1, p2, r, K) in cases.items():
        d, i, nn = ball_query(p1=torch.from_numpy(p1)[None], p2=torch.from_numpy(p2)[None], radius=r, K=K)
        out.update({f"{name}__p1": p1, f"{name}__p2": p2, f"{name}__radius": r, f"{name}__K": K,
                    f"{name}__dists": d[0].numpy(), f"{name}__idx": i[0].numpy(), f"{name}__nn": nn[0].numpy()})
        print(name, "full rows:", int((i[0, :, -1] >= 0).sum()), "empty rows:", int((i[0, :, 0] < 0).sum()))
    # the renderer's call shape: p2 replicated over the batch (models/renderer.py:113)
    p1b = torch.from_numpy(q[:128]).view(2, 64, 3)
    p2b = torch.from_numpy(cloud)[None].repeat(2, 1, 1)
    d, i, nn = ball_query(p1=p1b, p2=p2b, radius=0.225, K=20)
    out.update({"batched__p1": p1b.numpy(), "batched__p2": p2b.numpy(), "batched__radius": 0.225, "batched__K": 20,
                "batched__dists": d.numpy(), "batched__idx": i.numpy(), "batched__nn": nn.numpy()})
    np.savez_compressed(os.path.join(args.out, "pytorch3d_ball_query.npz"), **out)
    print("wrote", os.path.join(args.out, "pytorch3d_ball_query.npz"))


if __name__ == "__main__":
    main()
