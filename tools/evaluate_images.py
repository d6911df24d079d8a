"""PSNR / SSIM over dumped PNGs, as /root/reference/utils/evaluate_images.ipynb cells 7-9 do (per view: <res>/<view>/GT/*.png
vs <res>/<view>/Pred/*.png, numeric file order, `--rollout` = the last 10 frames, otherwise all but the last 10), with the
metrics of neurofluid_amd/metrics.py on the GPU.  LPIPS (cell 6) is computed when the state dict of lpips.LPIPS(net='vgg') is handed over
(--lpips-weights=FILE; see neurofluid_amd.metrics.LPIPS: the pretrained weights are third-party data).
usage: python tools/evaluate_images.py <res_dir> view_6 [view_7 ...] [--rollout] [--lpips-weights=lpips_vgg.pt]"""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

from neurofluid_amd import metrics


def read_images_in_dir(imgs_dir, rollout=False):
    fnames = sorted(glob.glob(os.path.join(imgs_dir, '*.png')), key=lambda x: int(os.path.basename(x)[:-4]))
    fnames = fnames[-10:] if rollout else fnames[:-10]
    imgs = [np.transpose((np.array(Image.open(f).convert('RGB')) / 255.).astype(np.float32), (2, 0, 1)) for f in fnames]
    return np.stack(imgs)


def main(argv):
    rollout = '--rollout' in argv
    args = [a for a in argv if not a.startswith('--')]
    res_dir, views = args[0], args[1:]
    dev = torch.device('cuda:0')
    wts = [a.split('=', 1)[1] for a in argv if a.startswith('--lpips-weights=')]
    lp = metrics.LPIPS(wts[0], device=dev) if wts else None
    all_errors = {'psnr': [], 'ssim': []}
    if lp is not None:
        all_errors['lpips'] = []
    for view in views:
        files_dir = os.path.join(res_dir, view)
        gt = torch.from_numpy(read_images_in_dir(os.path.join(files_dir, 'GT'), rollout)).to(dev)
        pred = torch.from_numpy(read_images_in_dir(os.path.join(files_dir, 'Pred'), rollout)).to(dev)
        errors = {k: [] for k in all_errors}
        for i in range(gt.shape[0]):          # per frame, like the notebook's "if OOM" path
            errors['psnr'].append(metrics.psnr(pred[i:i + 1], gt[i:i + 1]).item())
            errors['ssim'].append(metrics.ssim(pred[i:i + 1], gt[i:i + 1]).item())
            if lp is not None:
                errors['lpips'].append(lp(pred[i:i + 1], gt[i:i + 1]).item())
        with open(os.path.join(files_dir, 'metrics.txt'), 'w') as f:
            f.write(str(errors))
        for k, v in errors.items():
            all_errors[k].append(v)
    for k, v in all_errors.items():
        print(k, np.mean(v))
    return all_errors


if __name__ == '__main__':
    main(sys.argv[1:])
