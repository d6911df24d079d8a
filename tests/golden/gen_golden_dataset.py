#!/usr/bin/env python3
"""
Golden vectors for the on-disk DATA FORMATS.  Runs ONLY in the authoring container, where
/root/reference exists: a tiny synthetic data set is written with neurofluid_amd.datasets.write_synthetic_dataset (fixed
arguments, recorded in the fixture), then read with the REFERENCE's own readers

    datasets/dataset.py                       BlenderDataset   (views x frames of rays / rgb / poses, particles, box)
    datasets/dataset_splishsplash_rawdata.py  ParticleDataset  (sliding windows, 'blender' layout, with and without rotation)

and what they return is stored as tests/golden/f1_dataset.npz.  tests/test_host_logic.py then re-writes the same data set and
requires the BUILD's readers to return the same arrays: the writer, the layout and the readers are pinned against the
reference's code, not against each other.  Only numeric arrays are stored.

Stand-ins needed for the import (nothing of them runs in the recorded values, except ToTensor's existence):
  * kornia.create_meshgrid  -> the 6-line pixel grid of gen_golden.py
  * torchvision.transforms  -> a module with a ToTensor class (the reference instantiates it and never calls it)
  * PIL.Image.ANTIALIAS     -> Image.LANCZOS (the constant was removed in Pillow 10; same filter)
Usage:  python tests/golden/gen_golden_dataset.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

WRITER_ARGS = dict(n_frames=4, img=8, n_side=3, views=("view_0", "view_1"), splits=("train",), camera_angle_x=0.323, seed=10)


def main():
    assert os.path.isdir(REF), "the reference tree is needed to record these vectors"
    sys.path.insert(0, HERE)
    from gen_golden import install_standins
    install_standins()
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class ToTensor:
        pass
    tvt.ToTensor = ToTensor
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    from PIL import Image
    if not hasattr(Image, "ANTIALIAS"):
        Image.ANTIALIAS = Image.LANCZOS
    sys.path.insert(0, REF)
    import importlib.util

    def load(name, rel):      # by file: the reference's `datasets` is a namespace package and loses to an installed `datasets`
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    RefBlender = load("ref_dataset", "datasets/dataset.py").BlenderDataset
    RefParticles = load("ref_dataset_particles", "datasets/dataset_splishsplash_rawdata.py").ParticleDataset
    from neurofluid_amd.datasets import write_synthetic_dataset

    cfg = types.SimpleNamespace(data_type="splishsplash")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        root = os.path.join(tmp, "watercube")
        write_synthetic_dataset(root, **WRITER_ARGS)
        ds = RefBlender(root, cfg, imgW=8, imgH=8, start_index=0, end_index=4, imgscale=1.0, viewnames=list(WRITER_ARGS["views"]),
                        split="train")
        out["blender_len"] = len(ds)
        for idx in (0, 2):
            item = ds[idx]
            for k, v in item.items():
                out[f"blender_{idx}__{k}"] = np.asarray(v if not torch.is_tensor(v) else v.numpy())
        # half-resolution variant (imgscale = 2: the PNGs are resized with the ANTIALIAS / LANCZOS filter)
        ds2 = RefBlender(root, cfg, imgW=8, imgH=8, start_index=1, end_index=3, imgscale=2.0, viewnames=["view_1"], split="train")
        item = ds2[0]
        out["blender_half__rgb"], out["blender_half__rays"] = item["rgb"].numpy(), item["rays"].numpy()
        out["blender_half__focal"] = np.asarray(item["focal"])
        pd = RefParticles(root, "blender", 0, 4, random_rot=False, window=3)
        out["particles_len"] = len(pd)
        for k, v in pd[1].items():
            out[f"particles_1__{k}"] = v.numpy()
        np.random.seed(123)
        pr = RefParticles(root, "blender", 0, 4, random_rot=True, window=2)
        out["particles_rot_len"] = len(pr)
        for k, v in pr[0].items():
            out[f"particles_rot0__{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "f1_dataset.npz"), **out)
    print("wrote f1_dataset.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "f1_dataset.npz")), "bytes")


if __name__ == "__main__":
    main()
