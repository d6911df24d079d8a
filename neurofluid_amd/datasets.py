"""Dataset readers with the reference's on-disk formats.

* ``BlenderDataset``  — /root/reference/datasets/dataset.py:21-174:  ``<root>/<view>/transforms_<split>.json``
  (``camera_angle_x``, ``frames[].{file_path, particle_path, transform_matrix}``, ``bounding_box``), RGBA PNG
  blended on white, per-frame particle ``.npz`` (``pos``/``vel``) or ``.pkl`` (``location``/``velocity``), and the
  joblib ``box.pt`` (``box``/``box_normals``).  Items pair frame ``i`` with frame ``i+1`` exactly like :152-171.
  Rays are generated ON DEVICE per view (nf_get_rays) when a device is given, instead of the reference's CPU
  precomputation of all (V,T,H,W,6) rays.
* ``ParticleDataset`` — /root/reference/datasets/dataset_splishsplash_rawdata.py:18-143 (``blender`` / ``raw`` layouts,
  sliding windows, optional random z-rotation).
* ``write_synthetic_dataset`` — writes that format from the synthetic watercube scene (tests, smoke runs).
"""
import glob
import json
import os
import os.path as osp
import pickle

import numpy as np
import torch
from torch.utils.data import Dataset


def _load_box(path):
    import joblib
    info = joblib.load(path)
    return np.asarray(info['box'], np.float32), np.asarray(info['box_normals'], np.float32)


def _read_particles(path, data_type):
    if data_type == 'blender':
        with open(path, 'rb') as fp:
            info = pickle.load(fp)
        return np.array(info['location']).reshape(-1, 3), np.array(info['velocity']).reshape(-1, 3)
    if data_type == 'splishsplash':
        info = np.load(path)
        return info['pos'], info['vel']
    raise NotImplementedError('please enter correct data type')


class BlenderDataset(Dataset):
    def __init__(self, root_dir, cfg, imgW, imgH, start_index, end_index, imgscale, viewnames, split='train',
                 ray_fn=None):
        assert imgW == imgH, 'image width should be equal to image height'
        self.root_dir, self.cfg, self.split = root_dir, cfg, split
        self.data_type = cfg['data_type'] if isinstance(cfg, dict) else cfg.data_type
        self.viewnames = viewnames
        self.W, self.H = int(imgW // imgscale), int(imgH // imgscale)
        self.start_index, self.end_index = start_index, end_index
        from . import ray_utils
        self._ray_fn = ray_fn or ray_utils.get_rays_cpu
        rays, rgbs, cws, self.focal_mv = [], [], [], []
        self.particles_pos, self.particles_vel = None, None
        for vi, view in enumerate(viewnames):
            vdir = osp.join(root_dir, view)
            with open(osp.join(vdir, f'transforms_{split}.json')) as f:
                self.meta = json.load(f)
            focal = .5 * self.W / np.tan(0.5 * self.meta['camera_angle_x'])
            self.focal_mv.append(focal)
            v_rays, v_rgbs, v_cw, pos, vel = [], [], [], [], []
            for frame in self.meta['frames'][start_index:end_index]:
                if vi == 0:
                    p, v = _read_particles(osp.join(vdir, split, frame['particle_path']), self.data_type)
                    pos.append(p); vel.append(v)
                pose = np.array(frame['transform_matrix'], dtype=np.float64)[:3, :4]
                v_cw.append(pose)
                v_rays.append(self._ray_fn(self.H, self.W, focal, torch.FloatTensor(pose)).numpy())
                v_rgbs.append(self._read_image(osp.join(vdir, '{}.png'.format(frame['file_path']))))
            rays.append(np.stack(v_rays)); rgbs.append(np.stack(v_rgbs)); cws.append(np.stack(v_cw))
            if vi == 0:
                self.particles_pos, self.particles_vel = np.stack(pos), np.stack(vel)
        self.all_rays_mv, self.all_rgbs_mv, self.all_cw_mv = np.stack(rays), np.stack(rgbs), np.stack(cws)
        self.box, self.box_normals = _load_box(osp.join(root_dir, self.meta['bounding_box']))

    def _read_image(self, path):
        from PIL import Image
        img = Image.open(path)
        if img.size != (self.W, self.H):
            img = img.resize((self.W, self.H), Image.LANCZOS)     # Image.ANTIALIAS of dataset.py:106 (removed in Pillow 10)
        a = np.asarray(img) / 255.
        a = a.reshape(-1, 4)
        return a[:, :3] * a[:, -1:] + (1 - a[:, -1:])           # blend on white

    def __getitem__(self, index):
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()   # noqa: E731
        d = {'cw': f(self.all_cw_mv[:, index]), 'rgb': f(self.all_rgbs_mv[:, index]), 'rays': f(self.all_rays_mv[:, index]),
             'box': f(self.box), 'box_normals': f(self.box_normals),
             'particles_pos': f(self.particles_pos[index]), 'particles_vel': f(self.particles_vel[index]),
             'focal': self.focal_mv,
             'cw_1': f(self.all_cw_mv[:, index + 1]), 'rays_1': f(self.all_rays_mv[:, index + 1]),
             'rgb_1': f(self.all_rgbs_mv[:, index + 1]),
             'particles_pos_1': f(self.particles_pos[index + 1]), 'particles_vel_1': f(self.particles_vel[index + 1])}
        return d

    def __len__(self):
        return self.all_rgbs_mv.shape[1] - 1


class ParticleDataset(Dataset):
    def __init__(self, data_path, data_type, start, end, random_rot=True, window=3):
        self.random_rot, self.window, self.root_dir, self.start, self.end = random_rot, window, data_path, start, end
        if data_type == 'raw':
            self.dataitems = self._collect(glob.glob(osp.join(data_path, 'sim*')), 'output/fluid_*.npz',
                                           lambda p: int(p.split('_')[-1][:-4]), lambda d: osp.join(d, 'box.pt'), 0)
        elif data_type == 'blender':
            self.dataitems = self._collect([osp.join(data_path, 'view_0')], 'train/particles/*.npz',
                                           lambda p: int(osp.basename(p)[:-4]), lambda d: osp.join(data_path, 'box.pt'), 1)
        elif data_type == 'blender_all':
            self.dataitems = self._collect(glob.glob(osp.join(data_path, '*')), 'train/particles/*.npz',
                                           lambda p: int(osp.basename(p)[:-4]), lambda d: osp.join(data_path, 'box.pt'), 1)
        else:
            raise NotImplementedError(data_type)

    def _collect(self, dirs, pattern, key, box_of, extra):
        samples = []
        for d in dirs:
            if not osp.isdir(d):
                continue
            paths = sorted(glob.glob(osp.join(d, pattern)), key=key)[self.start:self.end]
            if not paths:
                continue
            box, normals = _load_box(box_of(d))
            for i in range(len(paths) - self.window + extra):
                s = {'box': box, 'box_normals': normals}
                for k in range(self.window):
                    info = np.load(paths[i + k])
                    s[f'particles_pos_{k}'], s[f'particles_vel_{k}'] = info['pos'], info['vel']
                samples.append(s)
        return samples

    def __getitem__(self, index):
        data = self.dataitems[index]
        if self.random_rot:
            a = np.random.uniform(0, 2 * np.pi)
            s, c = np.sin(a), np.cos(a)
            R = np.array([c, -s, 0, s, c, 0, 0, 0, 1], dtype=np.float32).reshape(3, 3)
            return {k: torch.from_numpy(np.matmul(v, R)).float() for k, v in data.items()}
        return {k: torch.from_numpy(np.asarray(v)).float() for k, v in data.items()}

    def __len__(self):
        return len(self.dataitems)


# ------------------------------------------------------------------------------------------------
def write_synthetic_dataset(root, n_frames=4, img=16, n_side=9, views=('view_0', 'view_1', 'view_2', 'view_3', 'view_4', 'view_5'),
                            splits=('train', 'test'), camera_angle_x=0.323, seed=10, shape='watercube', order='random'):
    """Synthetic scene in the reference's on-disk layout: a particle body falling under gravity (analytic frames), a
    sampled box, RGBA images with an analytic pattern (content is irrelevant to the kernels; PSNR-vs-paper needs the
    released data).  shape: 'watercube' (n_side^3 jittered lattice, lattice index order) or 'bunny' / 'honeycone'
    (synthetic.shaped_particles: ~4.8 k / 4.4 k particles in `order` = random | scan | shells index order)."""
    import joblib
    from PIL import Image
    rng = np.random.RandomState(seed)
    if shape == 'watercube':
        ax = [c + 0.05 * np.arange(n_side) for c in (-0.05 * (n_side - 1) / 2, -0.05 * (n_side - 1) / 2, -0.975)]
        pos0 = np.stack(np.meshgrid(*ax, indexing='ij'), -1).reshape(-1, 3) + rng.uniform(-0.005, 0.005, (n_side ** 3, 3))
    else:
        from .synthetic import shaped_particles
        pos0 = shaped_particles(shape, seed=seed, order=order).numpy().astype(np.float64)
    dt, g = 1 / 50, np.array([0, 0, -9.81])
    lo, hi = np.array([-1.0, -1.0, -1.0]), np.array([1.0, 1.0, 2.4552])
    pts, nrm = [], []
    for axis in range(3):
        o = [a for a in range(3) if a != axis]
        u, v = np.arange(lo[o[0]], hi[o[0]] + 1e-6, 0.05), np.arange(lo[o[1]], hi[o[1]] + 1e-6, 0.05)
        uu, vv = np.meshgrid(u, v, indexing='ij')
        for side, val in ((0, lo[axis]), (1, hi[axis])):
            p = np.zeros((uu.size, 3)); p[:, o[0]] = uu.ravel(); p[:, o[1]] = vv.ravel(); p[:, axis] = val
            n = np.zeros((uu.size, 3)); n[:, axis] = 1.0 if side == 0 else -1.0
            pts.append(p); nrm.append(n)
    os.makedirs(root, exist_ok=True)
    joblib.dump({'box': np.concatenate(pts).astype(np.float32), 'box_normals': np.concatenate(nrm).astype(np.float32)},
                osp.join(root, 'box.pt'))
    base = np.array([[0.3597943186759949, 0.09052024036645889, -0.18696719408035278, -4.842308521270752],
                     [-0.2077273577451706, 0.15678563714027405, -0.32383665442466736, -8.387124061584473],
                     [0.0, 0.37393447756767273, 0.181040421128273, 4.688809871673584], [0, 0, 0, 1.0]])
    for vi, view in enumerate(views):
        ang = 2 * np.pi * vi / max(len(views), 1)
        Rz = np.array([[np.cos(ang), -np.sin(ang), 0, 0], [np.sin(ang), np.cos(ang), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
        pose = Rz @ base
        for split in splits:
            pdir = osp.join(root, view, split, 'particles')
            os.makedirs(pdir, exist_ok=True)
            os.makedirs(osp.join(root, view, split, 'images'), exist_ok=True)
            frames = []
            for t in range(n_frames):
                tt = t * dt
                p = pos0 + 0.5 * g * tt * tt
                p[:, 2] = np.maximum(p[:, 2], -0.975)
                v = np.tile(g * tt, (p.shape[0], 1))
                np.savez(osp.join(pdir, f'{t}.npz'), pos=p.astype(np.float32), vel=v.astype(np.float32))
                yy, xx = np.mgrid[0:img, 0:img]
                rgba = np.zeros((img, img, 4), np.uint8)
                rgba[..., 0] = (xx * 255 // max(img - 1, 1)); rgba[..., 1] = (yy * 255 // max(img - 1, 1))
                rgba[..., 2] = (40 * t) % 256
                rgba[..., 3] = np.where((xx - img / 2) ** 2 + (yy - img / 2) ** 2 < (img / 3) ** 2, 255, 0)
                Image.fromarray(rgba, 'RGBA').save(osp.join(root, view, split, 'images', f'r_{t}.png'))
                frames.append({'file_path': f'{split}/images/r_{t}', 'particle_path': f'particles/{t}.npz',
                               'transform_matrix': pose.tolist()})
            with open(osp.join(root, view, f'transforms_{split}.json'), 'w') as f:
                json.dump({'camera_angle_x': camera_angle_x, 'frames': frames, 'bounding_box': 'box.pt'}, f)
    return root
