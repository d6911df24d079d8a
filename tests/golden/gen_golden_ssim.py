"""Generates tests/golden/f4_ssim.npz by EXECUTING the metric classes of the reference's notebook
(/root/reference/utils/evaluate_images.ipynb, cells 3-5: MSE / PSNR / SSIM) on seeded images.  Runs in the build container
only (reads /root/reference); commits inputs and the values the reference returned — no reference source."""
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference/utils/evaluate_images.ipynb"
assert os.path.exists(REF), "the reference is only present in the build container"
nb = json.load(open(REF))
ns = {}
exec("import math\nimport torch\nimport torch.nn as nn\nimport torch.nn.functional as F\n", ns)
for cell in nb["cells"]:
    src = "".join(cell["source"])
    if cell["cell_type"] == "code" and src.lstrip().startswith("#") and ("class MSE" in src or "class PSNR" in src or "class SSIM" in src):
        exec(src, ns)
SSIM, PSNR = ns["SSIM"], ns["PSNR"]

g = torch.Generator().manual_seed(4)
out = {}
cases = {
    "a": (1, 3, 32, 32, 0.0, 1.0),      # [0, 1] images
    "b": (2, 3, 48, 37, 0.0, 1.0),      # ragged tile edges, two images
    "c": (1, 1, 11, 11, 0.0, 1.0),      # a single valid position
    "d": (1, 3, 40, 40, 0.0, 255.0),    # 8-bit range: L = 255
    "e": (1, 3, 33, 50, -1.0, 1.0),     # tanh range: L = 2
}
for name, (B, C, H, W, lo, hi) in cases.items():
    gt = torch.rand(B, C, H, W, generator=g) * (hi - lo) + lo
    pred = (gt + 0.1 * (hi - lo) * torch.randn(B, C, H, W, generator=g)).clamp(lo, hi)
    out[f"{name}_pred"], out[f"{name}_gt"] = pred.numpy(), gt.numpy()
    out[f"{name}_ssim"] = np.float32(SSIM()(pred, gt).item())
    out[f"{name}_ssim_per_image"] = SSIM()(pred, gt, size_average=False).numpy()
    out[f"{name}_psnr"] = np.float32(PSNR()(pred, gt).item())
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "f4_ssim.npz"), **out)
print({k: v for k, v in out.items() if k.endswith("ssim") or k.endswith("psnr")})
