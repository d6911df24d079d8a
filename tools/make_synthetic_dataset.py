#!/usr/bin/env python3
"""Writes a synthetic data set in the reference's on-disk format (datasets/dataset.py:62-149: transforms_<split>.json,
RGBA PNGs, per-frame particle .npz, joblib box.pt) under data/synthetic/<name>, which is where configs/dataset.yaml
points.  The released NeuroFluid data is not available offline; these stand-ins have the right shapes and file
layout (image CONTENT is an analytic pattern), so every entry point runs end to end:

    python tools/make_synthetic_dataset.py --dataset watercube --img 400 --frames 61
    python tools/make_synthetic_dataset.py --dataset bunny     --img 800 --frames 61      # BASELINE config 4
    python tools/make_synthetic_dataset.py --dataset honeycone --img 800 --frames 201     # BASELINE config 5
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="watercube", choices=["watercube", "bunny", "honeycone", "watersphere"])
    ap.add_argument("--out", default="")
    ap.add_argument("--img", type=int, default=400)
    ap.add_argument("--frames", type=int, default=61)
    ap.add_argument("--order", default="random", choices=["random", "scan", "shells"],
                    help="index order of the bunny / honeycone particles (the first-K search depends on it)")
    args = ap.parse_args()
    from neurofluid_amd.datasets import write_synthetic_dataset
    root = args.out or os.path.join("data", "synthetic", args.dataset)
    shape = {"watercube": "watercube", "watersphere": "watercube"}.get(args.dataset, args.dataset)
    write_synthetic_dataset(root, n_frames=args.frames, img=args.img, n_side=17, shape=shape, order=args.order)
    print("wrote", root)


if __name__ == "__main__":
    main()
