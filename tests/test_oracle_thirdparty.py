"""Upgrades the pin of the third-party half of the path when fixtures recorded from the REAL libraries are present
(tests/golden/thirdparty/, written by tools/gen_goldens_open3d.py / tools/gen_goldens_pytorch3d.py on a machine that
has Open3D 0.15.2 / PyTorch3D 0.6.1).  Without them every test here SKIPS with the reason "parity unpinned": the
oracle's restatement of pytorch3d.ops.ball_query and of Open3D's ContinuousConv / FixedRadiusSearch is then pinned
only by analytic known-answer tests (tests/test_oracle_trans.py) — SURVEY §8c."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

TP = os.environ.get("NF_THIRDPARTY_GOLDEN", os.path.join(GOLDEN, "thirdparty"))     # env override: dry-running this file
O3D = sorted(glob.glob(os.path.join(TP, "open3d_*.npz")))
P3D = os.path.join(TP, "pytorch3d_ball_query.npz")
UNPINNED_O3D = "parity unpinned: no Open3D fixtures (run tools/gen_goldens_open3d.py where open3d 0.15.2 is installed)"
UNPINNED_P3D = "parity unpinned: no PyTorch3D fixture (run tools/gen_goldens_pytorch3d.py where pytorch3d 0.6.1 is installed)"


def pin_status():
    """What tests/… and DESIGN.md may claim about the third-party half."""
    return {"open3d": "pinned" if O3D else "unpinned", "pytorch3d": "pinned" if os.path.exists(P3D) else "unpinned"}


def _rows(idx, rs):
    return [np.sort(idx[rs[i]:rs[i + 1]]) for i in range(len(rs) - 1)]


@pytest.mark.skipif(not O3D, reason=UNPINNED_O3D)
@pytest.mark.parametrize("path", O3D or ["-"])
def test_oracle_vs_open3d_continuous_conv(path):
    """B2 / B4 / B6 (+ B8 gradients): oracle radius search, cconv forward and its autograd vs the real layer."""
    from oracle import trans_oracle as to
    g = dict(np.load(path))
    extent = float(g["extent"])
    inp, outp = torch.from_numpy(g["inp_positions"]), torch.from_numpy(g["out_positions"])
    idx, rs, d2 = to.radius_search(inp, outp, extent / 2, True)
    # B2: same neighbour SETS per query (Open3D's order inside a row is hash-table order; sums are order-independent
    # up to rounding), same squared distances
    assert np.array_equal(rs.numpy(), g["neighbors_row_splits"])
    got_rows, ref_rows = _rows(idx.numpy(), rs.numpy()), _rows(g["neighbors_index"], g["neighbors_row_splits"])
    assert all(np.array_equal(a, b) for a, b in zip(got_rows, ref_rows))
    for i in range(0, len(ref_rows), 37):
        a = np.sort(d2.numpy()[rs[i]:rs[i + 1]]); b = np.sort(g["neighbors_distance"][rs[i]:rs[i + 1]])
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)
    # B6
    assert np.array_equal((rs[1:] - rs[:-1]).float().numpy(), g["neighbor_counts"])
    # B4 forward
    feats = torch.from_numpy(g["inp_features"]).requires_grad_(True)
    kernel = torch.from_numpy(g["kernel"]).requires_grad_(True)
    out = to.cconv(feats, inp, outp, extent, kernel, torch.from_numpy(g["bias"]), idx, rs, d2)
    torch.testing.assert_close(out.detach(), torch.from_numpy(g["output"]), rtol=1e-4, atol=2e-5)
    # B8: gradients w.r.t. filter and input features
    (out * torch.from_numpy(g["grad_output"])).sum().backward()
    for got, ref in ((kernel.grad, g["grad_kernel"]), (feats.grad, g["grad_features"])):
        ref = torch.from_numpy(ref)
        assert float((got - ref).norm() / ref.norm()) < 1e-4
    # checkpoint surface: parameter / buffer names of the real layer must load into the build's layer
    from neurofluid_amd.transmodel import ContinuousConv
    mine = ContinuousConv(kernel_size=[4, 4, 4], in_channels=g["kernel"].shape[-2], filters=g["kernel"].shape[-1],
                          window_function=to.window_poly6)
    sd = {k: torch.from_numpy(g[f"state__{k}"]) for k in g["state_dict_keys"]}
    mine.load_state_dict(sd, strict=True)


@pytest.mark.skipif(not os.path.exists(P3D), reason=UNPINNED_P3D)
def test_oracle_vs_pytorch3d_ball_query():
    """A2: first-K-by-index, strict `<` on squared distance, padding -1 / 0 / 0, idx int64."""
    from oracle import neighbors
    g = dict(np.load(P3D))
    names = sorted({k.split("__")[0] for k in g if "__" in k})
    assert names
    for name in names:
        p1, p2 = g[f"{name}__p1"], g[f"{name}__p2"]
        r, K = float(g[f"{name}__radius"]), int(g[f"{name}__K"])
        if p1.ndim == 2:
            p1, p2 = p1[None], p2[None]
            ref = [g[f"{name}__{k}"][None] for k in ("dists", "idx", "nn")]
        else:
            ref = [g[f"{name}__{k}"] for k in ("dists", "idx", "nn")]
        for b in range(p1.shape[0]):
            d, i, nn = neighbors.ball_query_firstk(p1[b], p2[b], r, K)
            assert i.dtype == np.int64 and np.array_equal(i, ref[1][b]), name
            assert np.array_equal(d, ref[0][b]) and np.array_equal(nn, ref[2][b]), name


def test_pin_status_is_reported():
    """Always runs: states which half is pinned, so a green suite cannot be mistaken for a pinned oracle."""
    st = pin_status()
    print("third-party pin status:", st)
    assert set(st) == {"open3d", "pytorch3d"} and set(st.values()) <= {"pinned", "unpinned"}
