"""Thin counterparts of the reference's callers of the hot path:
  BaseTrainer        trainer/basetrainer.py:17-343      (seeding, dirs, ckpt loaders, losses, chunk loop, PNG dump)
  RendererTrainer    trainer/trainer_renderer.py:22-175  (warm-up: GT particles, 4 views x ray_chunk rays / step)
  E2ETrainer         trainer/trainer_e2e.py:26-371       (transition step -> render predicted particles -> RGB + boundary)
  TransModelTrainer  trainer/trainer_transmodel.py       (supervised 2-step unroll)
  E2EEvaluator / RendererEvaluation / TransModelEvaluation   eval_e2e.py, eval_renderer.py, eval_transmodel.py
Host plumbing only; every heavy op goes through RenderNet / ParticleNet (HIP).  TensorBoard is optional.
One process per GPU: when launched through torch.distributed.run the full-image loops shard ray chunks over ranks
and training all-reduces gradients (neurofluid_amd/dist.py)."""
import json
import os
import sys
import os.path as osp
import random

import numpy as np
import torch

from . import dist as nfdist
from .datasets import BlenderDataset, ParticleDataset
from .point_eval import FluidErrors
from .render_loop import render_image as _render_image
from .renderer import RenderNet
from .train_step import (ExponentialLR, PixelSampler, random_sample_coords, _upload, choice_without_replacement, gather_view_pixels,
                         make_adam, summed_view_mse, e2e_loss, portable_optimizer_state, load_optimizer_state, GraphedRendererStep)
from .transmodel import ParticleNet, PairCapacityExceeded

to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)   # noqa: E731  trainer/basetrainer.py:16
img2mse = lambda x, y: torch.mean((x - y) ** 2)              # noqa: E731
mse2psnr = lambda x: -10. * torch.log(x) / np.log(10.)       # noqa: E731


class _NullWriter:
    def __getattr__(self, name):
        return lambda *a, **k: None


def _writer(log_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except Exception:
        return _NullWriter()


def record2obj(pos, fp, color=(255, 0, 0)):
    """utils/particles_utils.py:38-42: one coloured vertex per particle."""
    p = pos.detach().cpu().numpy() if isinstance(pos, torch.Tensor) else np.asarray(pos)
    for x, y, z in p:
        fp.write('v {} {} {} {} {} {}\n'.format(x, y, z, *color))


class BaseTrainer:
    def __init__(self, options):
        self.options = options
        self.rank, self.world, self.local_rank = nfdist.init_from_env()
        self.seed_everything(options.TRAIN.seed if 'TRAIN' in options and 'seed' in options.TRAIN else 10)
        self.exppath = osp.join(options.expdir, options.expname)
        self.imgpath = osp.join(self.exppath, 'images')
        self.particlepath = osp.join(self.exppath, 'particles')
        for d in ('models', 'images', 'particles'):
            os.makedirs(osp.join(self.exppath, d), exist_ok=True)
        self.summary_writer = _writer(self.exppath) if self.rank == 0 else _NullWriter()
        if not torch.cuda.is_available():
            raise RuntimeError('neurofluid_amd needs an MI355X: there is no CPU fallback for the hot path')
        self.device = torch.device('cuda', self.local_rank)
        torch.cuda.set_device(self.device)
        self.init_fn()
        self.init_box_boundary()
        self.seed_data_streams()
        if self.options.resume_from != '':
            self.resume(self.options.resume_from)

    def init_fn(self):
        raise NotImplementedError()

    def seed_data_streams(self):
        """Data-parallel training: the models were initialised from the SAME seed on every rank (identical replicas);
        the streams that draw the training data (pixel selections: np.random.choice, trainer_renderer.py:119; sample
        order / z-rotation: np.random, dataset_splishsplash_rawdata.py:128-135) are re-seeded with seed + rank, so that
        N ranks average N different batches.  With one process the reference's single stream is kept untouched."""
        if self.world > 1:
            np.random.seed(self._seed + self.rank)
            random.seed(self._seed + self.rank)

    def resume(self, ckpt_file):
        raise NotImplementedError()

    def seed_everything(self, seed):
        self._seed = int(seed)
        random.seed(seed)
        os.environ['PYTHONHASHSEED'] = str(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)

    def init_box_boundary(self, particle_radius=0.025):
        self.x_bound = [1 - particle_radius, -1 + particle_radius]
        self.y_bound = [1 - particle_radius, -1 + particle_radius]
        self.z_bound = [2.4552 - particle_radius, -1 + particle_radius]

    def strict_clip_particles(self, pos):
        """basetrainer.py:108-116: per-axis clamp to the box.  One clamp against (3,) bound tensors instead of three
        select / clamp nodes and a stack: the same values and the same gradient mask, 5 kernels instead of 28 in the
        forward + backward of the boundary loss (the e2e step is launch-bound around its loss)."""
        assert pos.dim() == 2
        key = (pos.device, pos.dtype)
        if getattr(self, "_bounds_key", None) != key:
            self._bounds_lo = torch.tensor([self.x_bound[1], self.y_bound[1], self.z_bound[1]], device=pos.device, dtype=pos.dtype)
            self._bounds_hi = torch.tensor([self.x_bound[0], self.y_bound[0], self.z_bound[0]], device=pos.device, dtype=pos.dtype)
            self._bounds_key = key
        return torch.clamp(pos, self._bounds_lo, self._bounds_hi)

    # ---- checkpoints (key names are the compatibility contract)
    def load_pretained_transition_model(self, path):
        ckpt = torch.load(path, map_location=self.device)
        ckpt = ckpt.get('transition_model_state_dict', ckpt.get('model_state_dict', ckpt))
        ckpt = {k: v for k, v in ckpt.items() if 'gravity' not in k}
        sd = self.transition_model.state_dict()
        sd.update(ckpt)
        self.transition_model.load_state_dict(sd, strict=True)

    def load_pretained_renderer_model(self, path, partial_load=False):
        ckpt = torch.load(path, map_location=self.device)['renderer_state_dict']
        if partial_load:
            ckpt = {k: v for k, v in ckpt.items() if 'sigma' in k or 'xyz_encoding' in k}
        sd = self.renderer.state_dict()
        sd.update(ckpt)
        self.renderer.load_state_dict(sd, strict=True)

    # ---- losses
    def set_RGB_criterion(self):
        self.rgb_criterion = torch.nn.MSELoss()

    def set_L1_criterion(self):
        self.L1_criterion = torch.nn.L1Loss()

    def cal_boundary_loss(self, pos):
        return self.L1_criterion(pos, self.strict_clip_particles(pos))

    def weighted_mse_loss(self, pred_pos, gt_pos, num_fluid_neighbors, gamma=0.5, neighbor_scale=1 / 40):
        importance = torch.exp(-neighbor_scale * num_fluid_neighbors)
        dist = torch.sqrt(torch.sum((pred_pos - gt_pos) ** 2, dim=-1) + 1e-12)
        return torch.mean(importance * dist ** gamma)

    def cal_grad_norm(self, model):
        return np.array([p.grad.detach().norm(2).item() for p in model.parameters() if p.grad is not None])

    def get_learning_rate(self, optimizer):
        return [g['lr'] for g in optimizer.param_groups]

    def random_sample_coords(self, H, W, global_step):
        return random_sample_coords(H, W, global_step, self.options.TRAIN.precrop_iters)

    def sample_pixels(self, rays_hw6, rgbs, H, W, global_step, ray_chunk, sel=None):
        coords = self.random_sample_coords(H, W, global_step)
        if sel is None:
            sel = choice_without_replacement(np.random, coords.shape[0], ray_chunk)
        sc = _upload(coords[sel].long(), rays_hw6.device)
        return rays_hw6[sc[:, 0], sc[:, 1]], rgbs.view(H, W, -1)[sc[:, 0], sc[:, 1]]

    # ---- the chunk loop (trainer/basetrainer.py:264-309)
    def render_image(self, particle_pos, N_ray, ro, rays, focal_length, cw, iseval=False, camera=None):
        """rays=None + camera=(H, W, focal, c2w): every rank generates the rays of its own chunks (render_loop.render_image)."""
        rc = self.options.RENDERER.ray.ray_chunk
        dev_chunk = rc
        if N_ray > rc:   # full-image loops: larger fused calls, still multiples of the reference's chunk
            dev_chunk = max(rc, int(self.options.RENDERER.get('device_ray_chunk', rc)) // rc * rc)
        shard = iseval and self.world > 1      # reference chunks interleaved over the ranks, fused per rank (render_loop)
        return _render_image(self.renderer, particle_pos, N_ray, ro, rays, focal_length, cw, iseval=iseval, ray_chunk=rc,
                             device_chunk=dev_chunk, rank=self.rank if shard else 0, world=self.world if shard else 1, camera=camera)

    # ---- image dumps
    def vis_rgbs(self, rgbs, channel=3, test=False):
        node = self.options.TEST if test else self.options.TRAIN
        W, H = int(node.imgW // node.scale), int(node.imgH // node.scale)
        return rgbs.reshape(H, W, channel).detach().cpu().permute(2, 0, 1)

    def _write_png(self, chw, filename):
        from PIL import Image
        a = to8b(chw.permute(1, 2, 0).numpy())
        if a.shape[-1] == 1:
            a = a[..., 0]
        os.makedirs(osp.dirname(filename), exist_ok=True)
        Image.fromarray(a).save(filename)

    def visualization(self, pred_rgbs, gt_rgbs, step, mask=None, prefix=None):
        if self.rank != 0:
            return
        self._write_png(self.vis_rgbs(gt_rgbs), '{}/{}_{:05d}.png'.format(self.imgpath, prefix, step))
        self._write_png(self.vis_rgbs(pred_rgbs), '{}/{}_{:05d}_pred.png'.format(self.imgpath, prefix, step))
        if mask is not None:
            self._write_png(self.vis_rgbs(mask, channel=1), '{}/{}_{:05d}_mask.png'.format(self.imgpath, prefix, step))

    def _to_dev(self, data):
        """Host -> device copy of one dataset item.  The container (``box`` / ``box_normals``, ~39 k points, the same
        for every frame) is uploaded ONCE and the same device tensors are handed out afterwards, so the caches keyed
        on them (box grid, scene bounds) hit; a rotated container (ParticleDataset random_rot) differs
        per item and is uploaded as such."""
        out = {}
        for k, v in data.items():
            if k in ('box', 'box_normals') and isinstance(v, torch.Tensor):
                cache = self.__dict__.setdefault('_static_dev', {})
                host, dev_t = cache.get(k, (None, None))
                if host is None or host.shape != v.shape or not torch.equal(host, v):
                    host, dev_t = v.clone(), v.to(self.device)
                    cache[k] = (host, dev_t)
                out[k] = dev_t
            else:
                out[k] = v.to(self.device) if isinstance(v, torch.Tensor) else v
        return out

    def _frame_cache_budget(self):
        """Bytes the frame cache may pin: TRAIN.frame_cache_gb when the config names it, otherwise a quarter of the memory
        that is free on the device when the cache is created, capped at 16 GiB — the renderer's activation buffers (about
        0.8 GB per training pass + 25 % headroom) and other tenants of a shared device need the rest."""
        gb = None
        try:
            gb = self.options.TRAIN.frame_cache_gb
        except Exception:
            gb = None
        if gb is not None:
            return int(float(gb) * (1 << 30))
        if self.device.type == 'cuda':
            free, _total = torch.cuda.mem_get_info(self.device)
            return int(min(free // 4, 16 << 30))
        return 1 << 30

    def release_frame_cache(self):
        """Drop the cached frames (called at the end of train(); the cache is only useful inside the epoch loop)."""
        self.__dict__.pop('_frames_dev', None)

    def _frame_on_device(self, dataset, index, budget_bytes=None):
        """dataset[index] on the device, kept there: the end-to-end loop walks the same frames every epoch (their rays and
        images are 11.5 MB per view and frame: 0.5 ms of pageable upload per step, during which the GPU idles).  Datasets
        whose items are not deterministic (ParticleDataset random_rot) must not come through here."""
        cache = self.__dict__.setdefault('_frames_dev', {'ds': None, 'items': {}, 'bytes': 0, 'budget': None})
        if cache['ds'] is not dataset:
            cache.update(ds=dataset, items={}, bytes=0)
        if budget_bytes is None:
            if cache.get('budget') is None:
                cache['budget'] = self._frame_cache_budget()
            budget_bytes = cache['budget']
        item = cache['items'].get(index)
        if item is None:
            item = self._to_dev(dataset[index])
            size = sum(v.numel() * v.element_size() for v in item.values() if isinstance(v, torch.Tensor))
            if cache['bytes'] + size <= budget_bytes:
                cache['items'][index] = item
                cache['bytes'] += size
        return item

    def _dataset(self, node_key, views, split, imgnode):
        o = self.options
        return BlenderDataset(o[node_key].path, o, start_index=o[node_key].start_index, end_index=o[node_key].end_index,
                              imgW=imgnode.imgW, imgH=imgnode.imgH, imgscale=imgnode.scale, viewnames=views, split=split)


# ================================================================================================
class RendererTrainer(BaseTrainer):
    def init_fn(self):
        self.start_step = 0
        o = self.options
        self.train_view_names, self.test_viewnames = o['train'].views.warmup, o['test'].views
        self.dataset = self._dataset('train', self.train_view_names, 'train', o.TRAIN)
        self.test_dataset = self._dataset('test', self.test_viewnames, 'test', o.TEST)
        self.renderer = RenderNet(o.RENDERER, near=o.near, far=o.far).to(self.device)
        if o.TRAIN.pretained_renderer != '':
            self.load_pretained_renderer_model(o.TRAIN.pretained_renderer, partial_load=o.TRAIN.partial_load)
        self.optimizer = make_adam(self.renderer.parameters(), lr=o.TRAIN.LR.lr)
        self.lr_scheduler = ExponentialLR(self.optimizer, decay_epochs=o.TRAIN.LR.decay_epochs, gamma=0.1) \
            if o.TRAIN.LR.use_scheduler else None
        self.set_RGB_criterion()

    def resume(self, ckpt_file):
        ck = torch.load(ckpt_file, map_location=self.device)
        self.start_step = ck['step']
        self.renderer.load_state_dict(ck['renderer_state_dict'], strict=True)
        load_optimizer_state(self.optimizer, ck['optimizer_state_dict'])

    def save_checkpoint(self, global_step):
        if self.rank == 0:
            torch.save({'step': global_step, 'renderer_state_dict': self.renderer.state_dict(),
                        'optimizer_state_dict': portable_optimizer_state(self.optimizer)},
                       osp.join(self.exppath, 'models', f'{global_step}.pt'))

    def train(self, max_steps=None):
        o = self.options
        H, W = int(o.TRAIN.imgH // o.TRAIN.scale), int(o.TRAIN.imgW // o.TRAIN.scale)
        self.renderer.train()
        data = self._to_dev(self.dataset[0])              # always frame 0 (trainer_renderer.py:81)
        last = o.TRAIN.N_iters if max_steps is None else min(o.TRAIN.N_iters, self.start_step + max_steps)
        loss = None
        # the global np.random stream, drawn in the reference's order but one step ahead on a host thread
        self._sampler = PixelSampler(np.random, len(self.train_view_names), o.RENDERER.ray.ray_chunk,
                                     lambda s: self.random_sample_coords(H, W, s).shape[0], self.start_step)
        # Steady state: the whole step (gather -> forward -> loss -> backward -> Adam) replayed as ONE HIP graph
        # (train_step.GraphedRendererStep; TRAIN.renderer_graph: False keeps the eager step).  The first three steps of a run are eager:
        # they learn the passes' row capacities the capture needs.
        nv, rc = len(self.train_view_names), int(o.RENDERER.ray.ray_chunk)
        std_rgb = type(self.rgb_criterion) is torch.nn.MSELoss and self.rgb_criterion.reduction == 'mean'
        gstep = None
        if bool(getattr(o.TRAIN, 'renderer_graph', True)) and std_rgb and \
                GraphedRendererStep.eligible(self.renderer, self.optimizer, data['particles_pos'], self.world):
            gstep = GraphedRendererStep(self.renderer, self.optimizer, data['particles_pos'],
                                        [dict(rays=data['rays'][v], rgb=data['rgb'][v], cw=data['cw'][v]) for v in range(nv)], H, W, rc)
        self._graph_step = gstep
        try:
            for step_idx in range(self.start_step, last):
                if gstep is not None and step_idx - self.start_step >= 3 and gstep.ready():
                    loss = gstep.step(self.random_sample_coords(H, W, step_idx), self._sampler.next(step_idx))
                    if self.lr_scheduler is not None:
                        self.lr_scheduler.step()
                    if (step_idx + 1) % o.TRAIN.log_interval == 0:
                        gstep.verify()              # a truncated (and redone) step must not reach the logged values
                        out, rgbs = gstep.last_outputs()
                        with torch.no_grad():
                            for v in range(nv):
                                sl = slice(v * rc, (v + 1) * rc)
                                lv = self.rgb_criterion(out['rgb0'][sl], rgbs[sl])
                                if self.renderer.N_importance > 0:
                                    lv = lv + self.rgb_criterion(out['rgb1'][sl], rgbs[sl])
                                self.summary_writer.add_scalar(f'{self.train_view_names[v]}/rgbloss', lv.item(), step_idx)
                else:
                    loss = self.train_step(data, nv, H, W, step_idx)
                    self.update_step(loss)
                if (step_idx + 1) % o.TRAIN.save_interval == 0:
                    if gstep is not None:
                        gstep.verify()
                    self.eval(step_idx)
                    self.save_checkpoint(step_idx)
            if gstep is not None:
                gstep.verify()
        finally:                      # also on an exception: the worker thread must not outlive the loop
            self._sampler.close()
            self._sampler = None
        return loss

    def update_step(self, loss):
        self.optimizer.zero_grad()
        loss.backward()
        nfdist.allreduce_grads(list(self.renderer.parameters()), self.world)
        self.optimizer.step()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()

    def train_step(self, data, view_num, H, W, step_idx):
        rc = self.options.RENDERER.ray.ray_chunk
        coords = self.random_sample_coords(H, W, step_idx)
        sels = self._sampler.next(step_idx) if getattr(self, '_sampler', None) is not None else \
            [choice_without_replacement(np.random, coords.shape[0], rc) for _ in range(view_num)]
        # the views are rendered in ONE fused call (rays are independent; per-ray camera position), their pixels gathered
        # with one upload and one index_select per tensor
        rays, rgbs, ro = gather_view_pixels([data['rays'][v] for v in range(view_num)], [data['rgb'][v] for v in range(view_num)],
                                            [data['cw'][v] for v in range(view_num)], coords, sels, H, W)
        out = self.renderer(data['particles_pos'], ro, rays, None, None)
        fine = self.renderer.N_importance > 0
        std_rgb = type(self.rgb_criterion) is torch.nn.MSELoss and self.rgb_criterion.reduction == 'mean'
        if std_rgb and rgbs.is_cuda:
            return e2e_loss(out, rgbs, view_num, fine)          # the views' MSE sums and their gradients in one launch
        if std_rgb:
            total = summed_view_mse(out, rgbs, view_num, fine)
        else:
            total = 0.
            for v in range(view_num):
                sl = slice(v * rc, (v + 1) * rc)
                total = total + self.rgb_criterion(out['rgb0'][sl], rgbs[sl])
                if fine:
                    total = total + self.rgb_criterion(out['rgb1'][sl], rgbs[sl])
        if (step_idx + 1) % self.options.TRAIN.log_interval == 0:
            with torch.no_grad():
                for v in range(view_num):
                    sl = slice(v * rc, (v + 1) * rc)
                    lv = self.rgb_criterion(out['rgb0'][sl], rgbs[sl])
                    if fine:
                        lv = lv + self.rgb_criterion(out['rgb1'][sl], rgbs[sl])
                    self.summary_writer.add_scalar(f'{self.train_view_names[v]}/rgbloss', lv.item(), step_idx)
        return total

    def eval(self, step_idx):
        self.renderer.eval()
        res = {}
        with torch.no_grad():
            data = self._to_dev(self.test_dataset[0])
            for v, view in enumerate(self.test_viewnames):
                cw = data['cw'][v]
                rays = data['rays'][v].reshape(-1, 6)
                ret = self.render_image(data['particles_pos'], rays.shape[0], self.renderer.set_ro(cw), rays,
                                        data['focal'][v], cw, iseval=True)
                for lvl, key in ((0, 'pred_rgbs_0'), (1, 'pred_rgbs_1')):
                    if key in ret:
                        psnr = mse2psnr(img2mse(ret[key], data['rgb'][v])).item()
                        res[f'{view}/psnr_{lvl}'] = psnr
                        self.summary_writer.add_scalar(f'{view}/psnr_0_{lvl}', psnr, step_idx)
                        self.visualization(ret[key], data['rgb'][v], step_idx, mask=ret.get(f'mask_{lvl}'),
                                           prefix=f'{"coarse" if lvl == 0 else "fine"}_0_{view}')
        self.renderer.train()
        return res


# ================================================================================================
def _piecewise(boundaries, values):
    def fn(x):
        f = values[0]
        for b, v in zip(boundaries, values[1:]):
            if x > b:
                f = v
            else:
                break
        return f
    return fn


class E2ETrainer(BaseTrainer):
    def init_fn(self):
        o = self.options
        self.start_step, self.eval_count = 0, 0
        self.train_view_names, self.test_viewnames = o['train'].views.dynamic, o['test'].views
        self.dataset = self._dataset('train', self.train_view_names, 'train', o.TRAIN)
        self.test_dataset = self._dataset('test', self.test_viewnames, 'test', o.TEST)
        self.transition_model = ParticleNet(gravity=o.gravity).to(self.device)
        # truncated BPTT of length 1 (trainer_e2e.py:196-198): one forward per backward, so the transition step's launch
        # sequences can be replayed as HIP graphs (the step is bound by the host's launch rate)
        self.transition_model.training_graph = bool(getattr(o.TRAIN, 'transition_graph', True))
        self.renderer = RenderNet(o.RENDERER, near=o.near, far=o.far).to(self.device)
        if o.TRAIN.pretrained_transition_model != '':
            self.load_pretained_transition_model(o.TRAIN.pretrained_transition_model)
        if o.TRAIN.pretained_renderer != '':
            self.load_pretained_renderer_model(o.TRAIN.pretained_renderer, partial_load=o.TRAIN.partial_load)
        lr_r, lr_t = o.TRAIN.LR.renderer_lr, o.TRAIN.LR.trans_lr
        self.separate = o.TRAIN.seperate_render_transition
        if self.separate:
            self.optimizer = make_adam([{'params': self.renderer.parameters(), 'lr': lr_r}])
            self.transition_optimizer = make_adam([{'params': self.transition_model.parameters(), 'lr': lr_t}])
        else:
            self.optimizer = make_adam([{'params': self.renderer.parameters(), 'lr': lr_r},
                                               {'params': self.transition_model.parameters(), 'lr': lr_t}])
        self.schedulers = []
        if o.TRAIN.LR.use_scheduler:      # trainer_e2e.py:83-139
            self.schedulers.append(torch.optim.lr_scheduler.LambdaLR(
                self.optimizer, _piecewise([10000, 75000, 150000], [1.0, 0.5, 0.25, 0.125])))
            if self.separate:
                self.schedulers.append(torch.optim.lr_scheduler.LambdaLR(
                    self.transition_optimizer, _piecewise([10000, 30000, 50000, 100000, 300000],
                                                          [1.0, 0.5, 0.25, 0.125, 0.0625, 0.03125, 0.015625])))
        self.set_RGB_criterion()
        self.set_L1_criterion()

    def resume(self, ckpt_file):
        ck = torch.load(ckpt_file, map_location=self.device)
        self.start_step = ck['step']
        self.renderer.load_state_dict(ck['renderer_state_dict'], strict=True)
        self.transition_model.load_state_dict(ck['transition_model_state_dict'], strict=True)

    def save_checkpoint(self, global_step):
        if self.rank == 0:
            torch.save({'step': global_step, 'renderer_state_dict': self.renderer.state_dict(),
                        'transition_model_state_dict': self.transition_model.state_dict(),
                        'optimizer_state_dict': portable_optimizer_state(self.optimizer)},
                       osp.join(self.exppath, 'models', f'{global_step}.pt'))

    def train(self, max_steps=None):
        o = self.options
        H, W = int(o.TRAIN.imgH // o.TRAIN.scale), int(o.TRAIN.imgW // o.TRAIN.scale)
        global_step, done, loss = self.start_step, 0, None
        self.transition_model.train(); self.renderer.train()
        # Steady state: the WHOLE step replayed as one HIP graph (e2e_graph.GraphedE2EStep; TRAIN.e2e_graph: False keeps the eager step).
        # The first steps of a run are eager: they learn the pair / row capacities the capture needs and create the optimiser state.
        gstep = self.__dict__.get('_graph_step')
        if gstep is None and bool(getattr(o.TRAIN, 'e2e_graph', True)):
            from .e2e_graph import GraphedE2EStep
            if GraphedE2EStep.eligible(self):
                gstep = self._graph_step = GraphedE2EStep(self, H, W)

        # (The warm-up trainer draws its pixels one step ahead on a host thread.  Here that was measured a loss — 4.5 -> 6.9 ms
        # per step: this step is bound by the host's launch sequence, and a second Python thread costs it the GIL.)
        try:
            for _epoch in range(self.start_step, o.TRAIN.epochs):
                self.tmp_fluid_error = FluidErrors()
                for data_idx in range(len(self.dataset)):
                    data = self._frame_on_device(self.dataset, data_idx)
                    # everything a redone step must start from again: the carried state and EVERY random stream the step draws from
                    # (numpy: the pixel choice; torch CPU / device generators: noise_std / perturb draws of the renderer)
                    if gstep is not None and self.__dict__.get('_eager_steps', 0) >= 3 and gstep.ready(data):
                        loss = self._graph_train_step(gstep, data, data_idx, len(self.train_view_names), H, W, global_step)
                        global_step += 1; done += 1
                        if (global_step + 1) % o.TRAIN.save_interval == 0:
                            gstep.verify()
                            self.eval(global_step)
                            self.save_checkpoint(global_step)
                        if max_steps is not None and done >= max_steps:
                            gstep.verify()
                            return loss
                        continue
                    if gstep is not None:
                        gstep.verify()
                        gstep.invalidate_state()            # this step runs outside the graph: the carried state is the trainer's again
                    self._eager_steps = self.__dict__.get('_eager_steps', 0) + 1
                    saved = (getattr(self, 'pos_for_next_step', None), getattr(self, 'vel_for_next_step', None), np.random.get_state(),
                             torch.get_rng_state(), torch.cuda.get_rng_state(self.device) if torch.cuda.is_available() else None)
                    try:
                        loss = self.train_step(data, data_idx, len(self.train_view_names), H, W, global_step)
                        self.update_step(loss, global_step)
                    except PairCapacityExceeded:
                        # the graph-replayed transition step met more neighbour pairs than its graphs were captured for (the
                        # capacities have been raised): redo THIS step from the state and the random stream it started with
                        self.pos_for_next_step, self.vel_for_next_step = saved[0], saved[1]
                        np.random.set_state(saved[2])
                        torch.set_rng_state(saved[3])
                        if saved[4] is not None:
                            torch.cuda.set_rng_state(saved[4], self.device)
                        loss = self.train_step(data, data_idx, len(self.train_view_names), H, W, global_step)
                        self.update_step(loss, global_step)
                    global_step += 1; done += 1
                    if (global_step + 1) % o.TRAIN.save_interval == 0:
                        # the transition replicas are kept equal by determinism alone (update_step): verify before rank 0's copy
                        # becomes THE checkpoint, and re-seed the others from it if a replica drifted
                        if not nfdist.replicas_in_sync(self.transition_model.parameters(), self.world):
                            self.replica_resyncs = getattr(self, 'replica_resyncs', 0) + 1
                            if self.rank == 0:
                                print(f'[e2e] step {global_step}: transition-model replicas differed; re-broadcast from rank 0')
                        self.eval(global_step)
                        self.save_checkpoint(global_step)
                    if max_steps is not None and done >= max_steps:
                        return loss
        finally:
            if gstep is not None and sys.exc_info()[0] is None:
                gstep.verify()                  # every enqueued step settled (a redo corrects the returned loss tensor in place)
            if not getattr(self, 'keep_frame_cache', False):      # (a caller that calls train() block by block — bench.py — keeps it)
                self.release_frame_cache()      # the cached frames are only useful inside the epoch loop
        return loss

    def _graph_train_step(self, gstep, data, data_idx, view_num, H, W, global_step):
        """train_step + update_step through the replayed graph: the same random draws in the same order (pixel grid, then one selection per
        view), the schedulers stepped behind the launch, the logged distance read from the step's own prediction once it is verified."""
        rc = self.options.RENDERER.ray.ray_chunk
        coords = self.random_sample_coords(H, W, global_step)
        sels = [choice_without_replacement(np.random, coords.shape[0], rc) for _ in range(view_num)]
        loss = gstep.step(data, data_idx, coords, sels)
        for s in self.schedulers:
            s.step()
        if (global_step + 1) % self.options.TRAIN.log_interval == 0:
            gstep.verify()                      # a truncated (and redone) step must not reach the logged metric
            d = self.tmp_fluid_error.cal_errors(gstep.pred_pos().detach(), data['particles_pos_1'], data_idx + 1)
            self.summary_writer.add_scalar('Train/pred2gt_distance', d, global_step)
        return loss

    def trainsition_step_for_training(self, data, data_idx):
        if data_idx == 0:
            self.pos_for_next_step, self.vel_for_next_step = data['particles_pos'], data['particles_vel']
        pred_pos, pred_vel, _ = self.transition_model(self.pos_for_next_step, self.vel_for_next_step, data['box'],
                                                      data['box_normals'])
        # truncated BPTT of length 1: the carried state is detached (trainer_e2e.py:196-198)
        self.pos_for_next_step, self.vel_for_next_step = pred_pos.detach().clone(), pred_vel.detach().clone()
        return pred_pos

    def train_step(self, data, data_idx, view_num, H, W, global_step):
        pred_pos = self.trainsition_step_for_training(data, data_idx)
        if self.world > 1 and pred_pos.requires_grad:
            # data-parallel over rays: dL/d(pred_pos) is averaged over the ranks (59 KB) before the replicated transition backward
            pred_pos.register_hook(nfdist.mean_over_ranks_hook(self.world))
        log = (global_step + 1) % self.options.TRAIN.log_interval == 0
        if log:
            # a truncated step (graph-replayed forward beyond its pair capacities) must not reach the logged metric: verify NOW on log
            # steps (raises PairCapacityExceeded before any side effect; every other step is verified at the start of backward())
            self.transition_model.verify_training_step()
            d = self.tmp_fluid_error.cal_errors(pred_pos.detach(), data['particles_pos_1'], data_idx + 1)
            self.summary_writer.add_scalar('Train/pred2gt_distance', d, global_step)
        rc = self.options.RENDERER.ray.ray_chunk
        # frame t+1 supervises the particles predicted from frame t (:224-227).  The views are drawn in the reference's
        # order and rendered in ONE fused call (rays are independent; per-ray camera position), as in the warm-up trainer
        coords = self.random_sample_coords(H, W, global_step)
        sels = [choice_without_replacement(np.random, coords.shape[0], rc) for _ in range(view_num)]
        rays, rgbs, ro = gather_view_pixels([data['rays_1'][v] for v in range(view_num)],
                                            [data['rgb_1'][v] for v in range(view_num)],
                                            [data['cw_1'][v] for v in range(view_num)], coords, sels, H, W)
        out = self.renderer(pred_pos, ro, rays, None, None)
        fine = self.renderer.N_importance > 0
        wb = self.options.TRAIN.loss_weight['boundary_loss']
        std_rgb = type(self.rgb_criterion) is torch.nn.MSELoss and self.rgb_criterion.reduction == 'mean'
        std_l1 = type(self.L1_criterion) is torch.nn.L1Loss and self.L1_criterion.reduction == 'mean'
        if std_rgb and std_l1 and rgbs.is_cuda and pred_pos.dim() == 2 and pred_pos.dtype == torch.float32:
            # the whole loss and its gradients in one launch (train_step.e2e_loss); the criteria of the reference's configuration
            return e2e_loss(out, rgbs, view_num, fine, pred_pos, ((self.x_bound[1], self.y_bound[1], self.z_bound[1]),
                                                                 (self.x_bound[0], self.y_bound[0], self.z_bound[0])), wb)
        if std_rgb:
            total = summed_view_mse(out, rgbs, view_num, fine)
        else:
            total = 0.
            for v in range(view_num):
                sl = slice(v * rc, (v + 1) * rc)
                total = total + self.rgb_criterion(out['rgb0'][sl], rgbs[sl])
                if fine:
                    total = total + self.rgb_criterion(out['rgb1'][sl], rgbs[sl])
        if wb != 0.:
            total = total + self.cal_boundary_loss(pred_pos) * wb
        return total

    def update_step(self, loss, global_step):
        clip = self.options.TRAIN.grad_clip_value
        self.optimizer.zero_grad()
        if self.separate:
            self.transition_optimizer.zero_grad()
        loss.backward()
        # renderer gradients: one flat-bucket all-reduce (5.35 MB); the transition model's are already identical on every rank
        # (its only upstream gradient, dL/d pred_pos, was averaged by train_step's hook: SURVEY 8e).  REQUIREMENT: the transition
        # forward / backward must be bitwise deterministic on every rank (true of nf_cconv / nf_gemm: no float atomics, index-sorted
        # grid; capacity redos replay the same sums) — train() checks the replicas' bits at every checkpoint (dist.replicas_in_sync)
        nfdist.allreduce_grads(list(self.renderer.parameters()), self.world)
        if clip != 0:
            torch.nn.utils.clip_grad_norm_(self.renderer.parameters(), clip)
            torch.nn.utils.clip_grad_norm_(self.transition_model.parameters(), clip)
        self.optimizer.step()
        if self.separate:
            self.transition_optimizer.step()
        for s in self.schedulers:
            s.step()

    def eval(self, step_idx, render_frames=(0, 20, 30)):
        self.eval_count += 1
        self.transition_model.eval(); self.renderer.eval()
        dists, fe = [], FluidErrors()
        with torch.no_grad():
            for data_idx in range(len(self.test_dataset)):
                data = self._to_dev(self.test_dataset[data_idx])
                if data_idx == 0:
                    pos, vel = data['particles_pos'], data['particles_vel']
                pos, vel, _ = self.transition_model(pos, vel, data['box'], data['box_normals'])
                dists.append(fe.cal_errors(pos, data['particles_pos_1'], data_idx + 1))
                if data_idx in render_frames:
                    for v, view in enumerate(self.test_viewnames):
                        cw = data['cw_1'][v]
                        rays = data['rays_1'][v].reshape(-1, 6)
                        ret = self.render_image(pos, rays.shape[0], self.renderer.set_ro(cw), rays, data['focal'][v], cw, iseval=True)
                        key = 'pred_rgbs_1' if 'pred_rgbs_1' in ret else 'pred_rgbs_0'
                        self.summary_writer.add_scalar(f'{view}/psnr_{data_idx}', mse2psnr(img2mse(ret[key], data['rgb_1'][v])).item(), step_idx)
                        self.visualization(ret[key], data['rgb_1'][v], step_idx, prefix=f'fine_{data_idx}_{view}')
            self.summary_writer.add_scalar('avg_pred2gt_distance', float(np.mean(dists)), step_idx)
        self.transition_model.train(); self.renderer.train()
        return dists


def _same_tensor(a, b):
    return a is b or (a.shape == b.shape and a.dtype == b.dtype and bool(torch.equal(a, b)))


# ================================================================================================
class E2EEvaluator(BaseTrainer):
    """eval_e2e.py:24-160: roll the transition model over the test frames, render every frame from every test view."""

    def init_fn(self):
        o = self.options
        self.test_viewnames = o['test'].views
        self.test_dataset = self._dataset('test', self.test_viewnames, 'test', o.TEST)
        self.transition_model = ParticleNet(gravity=o.gravity).to(self.device)
        self.renderer = RenderNet(o.RENDERER, near=o.near, far=o.far).to(self.device)

    def resume(self, ckpt_file):
        ck = torch.load(ckpt_file, map_location=self.device)
        self.renderer.load_state_dict(ck['renderer_state_dict'], strict=True)
        self.transition_model.load_state_dict(ck['transition_model_state_dict'], strict=True)

    def eval(self, dump=True):
        self.transition_model.eval(); self.renderer.eval()
        dists, psnrs = [], []
        self.fluid_error = FluidErrors()
        from .rollout import CoupledRollout
        roll = None
        with torch.no_grad():
            n_frames = len(self.test_dataset)
            box_dev = None
            for data_idx in range(n_frames):
                data = self._to_dev(self.test_dataset[data_idx])       # (an unchanged container comes back as the SAME device tensors)
                if data_idx == 0:
                    # the rollout (eval_e2e.py:75-84: pos, vel = transition_model(pos, vel, box, box_normals) in front of every frame) with the step
                    # of frame t + 1 in flight on a side stream while frame t is measured, dumped and rendered (rollout.CoupledRollout: same
                    # states bit for bit).  The lookahead assumes the NEXT frame's container is this frame's (datasets: one box.pt per scene)
                    roll = CoupledRollout(self.transition_model, data['box'], data['box_normals'], device=self.device)
                    roll.start(data['particles_pos'], data['particles_vel'])
                    box_dev = (data['box'], data['box_normals'])
                elif not (_same_tensor(data['box'], box_dev[0]) and _same_tensor(data['box_normals'], box_dev[1])):
                    # a per-frame container (the reference passes data['box'] of EVERY frame, eval_e2e.py:72-77): the step in flight was
                    # computed against the previous frame's box — discard it and redo this frame's step with its own container
                    roll.drop()
                    roll = CoupledRollout(self.transition_model, data['box'], data['box_normals'], device=self.device)
                    roll.start(pos, vel)
                    box_dev = (data['box'], data['box_normals'])
                pos, vel, _ = roll.next_state(last=data_idx == n_frames - 1)
                dists.append(self.fluid_error.cal_errors(pos, data['particles_pos_1'], data_idx + 1))
                if dump and self.rank == 0:
                    for sub, p, col in (('Pred', pos, (255, 0, 0)), ('GT', data['particles_pos_1'], (3, 168, 158))):
                        os.makedirs(osp.join(self.particlepath, sub), exist_ok=True)
                        with open(osp.join(self.particlepath, sub, f'{data_idx + 1}.obj'), 'w') as fp:
                            record2obj(p, fp, color=col)
                for v, view in enumerate(self.test_viewnames):
                    cw = data['cw_1'][v]
                    rays = data['rays_1'][v].reshape(-1, 6)
                    ret = self.render_image(pos, rays.shape[0], self.renderer.set_ro(cw), rays, data['focal'][v], cw, iseval=True)
                    for lvl, key in (('coarse', 'pred_rgbs_0'), ('fine', 'pred_rgbs_1')):
                        if key in ret:
                            psnrs.append(mse2psnr(img2mse(ret[key], data['rgb_1'][v])).item())
                            if dump and self.rank == 0:
                                self._write_png(self.vis_rgbs(data['rgb_1'][v], test=True), f'{self.imgpath}/{lvl}/{view}/GT/{data_idx + 1:05d}.png')
                                self._write_png(self.vis_rgbs(ret[key], test=True), f'{self.imgpath}/{lvl}/{view}/Pred/{data_idx + 1:05d}.png')
        if roll is not None:
            roll.drop()
        if dump and self.rank == 0:
            import joblib
            joblib.dump({'dist': dists}, osp.join(self.exppath, 'pred2gt.pt'))
        return {'pred2gt': dists, 'psnr': psnrs}


class RendererEvaluation(BaseTrainer):
    """eval_renderer.py:46-148: render GT particle frames from one fixed camera with a warm-up checkpoint."""

    def init_fn(self):
        o = self.options
        self.renderer = RenderNet(o.RENDERER, near=o.TEST.near, far=o.TEST.far).to(self.device)
        files = sorted([f for f in os.listdir(o.TEST.data_path) if f.endswith('.npz')], key=lambda s: int(s[:-4]))
        self.files = [osp.join(o.TEST.data_path, f) for f in files][o.TEST.start_index:o.TEST.end_index]

    def resume(self, ckpt_file):
        sd = self.renderer.state_dict()
        sd.update(torch.load(ckpt_file, map_location=self.device)['renderer_state_dict'])
        self.renderer.load_state_dict(sd, strict=True)

    def pre_request(self, c2w=None):
        from . import ray_utils
        o = self.options
        W, H = o.TEST.imgW, o.TEST.imgH
        focal = .5 * W / np.tan(0.5 * o.TEST.camera_angle_x)
        if c2w is None:     # pose values of the reference's hard-coded evaluation camera (eval_renderer.py:67-92)
            c2w = torch.tensor([[0.3597943186759949, 0.09052024036645889, -0.18696719408035278, -4.842308521270752],
                                [-0.2077273577451706, 0.15678563714027405, -0.32383665442466736, -8.387124061584473],
                                [0.0, 0.37393447756767273, 0.181040421128273, 4.688809871673584]])
        c2w = c2w.to(self.device)
        if getattr(self, 'world', 1) > 1:
            # ray-tile sharding: no (H*W, 6) tensor per rank — each rank generates its own chunks' rays per frame
            return {'cw': c2w, 'focal': focal, 'rays': None, 'camera': (H, W, focal, c2w), 'n_ray': H * W}
        rays = ray_utils.get_rays_device(H, W, focal, c2w)
        return {'cw': c2w, 'focal': focal, 'rays': rays, 'camera': None, 'n_ray': rays.shape[0]}

    def eval(self, max_frames=53, dump=True):
        self.renderer.eval()
        rp = self.pre_request()
        out = []
        with torch.no_grad():
            for i, f in enumerate(self.files[:max_frames]):
                pos = torch.from_numpy(np.load(f)['pos']).float().to(self.device)
                ret = self.render_image(pos, rp['n_ray'], self.renderer.set_ro(rp['cw']), rp['rays'], rp['focal'], rp['cw'], iseval=True,
                                        camera=rp['camera'])
                out.append(ret)
                if dump and self.rank == 0:
                    name = osp.basename(f)[:-4]
                    for lvl, key in (('coarse', 'pred_rgbs_0'), ('fine', 'pred_rgbs_1')):
                        if key in ret:
                            self._write_png(self.vis_rgbs(ret[key], test=True), osp.join(self.exppath, 'render_GT', f'{lvl}_pred_{name}.png'))
        return out


class TransModelEvaluation:
    """eval_transmodel.py:19-154: roll out the transition model only; raw and box-clipped FluidErrors."""

    def __init__(self, options):
        self.options = options
        self.rank, self.world, local = nfdist.init_from_env()
        if not torch.cuda.is_available():
            raise RuntimeError('neurofluid_amd needs an MI355X: there is no CPU fallback for the hot path')
        self.device = torch.device('cuda', local)
        self.exppath = osp.join(options.expdir, options.expname)
        os.makedirs(osp.join(self.exppath, 'clip'), exist_ok=True)
        self.transition_model = ParticleNet(gravity=options.TEST.gravity).to(self.device)
        if options.resume_from:
            ck = torch.load(options.resume_from, map_location=self.device)
            ck = ck.get('transition_model_state_dict', ck.get('model_state_dict', ck))
            sd = self.transition_model.state_dict()
            sd.update({k: v for k, v in ck.items() if 'gravity' not in k})
            self.transition_model.load_state_dict(sd, strict=True)
        self.dataset = ParticleDataset(options.TEST.datapath, options.TEST.datatype, options.TEST.start_index,
                                       options.TEST.end_index, random_rot=False, window=2)
        self.fluid_erros, self.cliped_fluid_erros = FluidErrors(), FluidErrors()
        BaseTrainer.init_box_boundary(self)

    strict_clip_particles = BaseTrainer.strict_clip_particles

    def eval(self):
        d_all, dc_all = [], []
        with torch.no_grad():
            for i in range(len(self.dataset)):
                data = {k: v.to(self.device) for k, v in self.dataset[i].items()}
                if i == 0:
                    pos, vel = data['particles_pos_0'], data['particles_vel_0']
                pos, vel, _ = self.transition_model(pos, vel, data['box'], data['box_normals'])
                gt = data['particles_pos_1']
                d_all.append(self.fluid_erros.cal_errors(pos, gt, i + 1))
                dc_all.append(self.cliped_fluid_erros.cal_errors(self.strict_clip_particles(pos), self.strict_clip_particles(gt),
                                                                 i + 1))
                if self.options.TEST.save_obj and self.rank == 0:
                    with open(osp.join(self.exppath, f'pred_{i + 1}.obj'), 'w') as fp:
                        record2obj(pos, fp, color=(255, 0, 0))
        self.fluid_erros.save(osp.join(self.exppath, 'res.json'))
        self.cliped_fluid_erros.save(osp.join(self.exppath, 'clip', 'res.json'))
        return {'pred2gt': d_all, 'pred2gt_clipped': dc_all}


class TransModelTrainer(BaseTrainer):
    """trainer/trainer_transmodel.py:24-262: supervised fine-tuning of the transition model.  Per sample: a 2-step
    unroll (state NOT detached in between, :179-180), loss = 0.5*wmse_1 + 0.5*wmse_2 + boundary_1 + boundary_2
    (:182-189), optional clip_grad_norm_ (:198-199), Adam.  ``N_iters`` counts EPOCHS over the shuffled dataset
    (:167-168, DataLoader(shuffle=True) :124), checkpoint + rollout eval every ``save_interval`` epochs (:216-221);
    the checkpoint's ``step`` is the epoch index (:218) and resume() restores model and optimizer only (:111-115)."""

    def init_fn(self):
        o = self.options
        self.eval_count, self.start_step = 0, 0
        self.transition_model = ParticleNet(gravity=o.TRAIN.gravity).to(self.device)
        if o.TRAIN.pretrained:
            self.load_pretained_transition_model(o.TRAIN.pretrained)
        dp = o.TRAIN.datapath
        self.dataset = ParticleDataset(dp.train, dp.train_datatype, o.TRAIN.start_index, o.TRAIN.end_index,
                                       random_rot=True, window=3)
        self.test_dataset = ParticleDataset(dp.eval, dp.eval_datatype, o.TRAIN.start_index, o.TRAIN.end_index,
                                            random_rot=False, window=3)
        self.optimizer = make_adam(self.transition_model.parameters(), lr=o.TRAIN.lr)
        self.set_L1_criterion()

    def init_box_boundary(self):
        super().init_box_boundary(self.options.TRAIN.get('particle_radius', 0.025))

    def resume(self, ckpt_file):
        ck = torch.load(ckpt_file, map_location=self.device)
        self.transition_model.load_state_dict(ck['model_state_dict'], strict=True)
        load_optimizer_state(self.optimizer, ck['optimizer_state_dict'])

    def sample_loss(self, data):
        """The loss of one training sample (trainer_transmodel.py:170-189); returns (loss, parts)."""
        box, bn = data['box'], data['box_normals']
        p1, v1, n1 = self.transition_model(data['particles_pos_0'], data['particles_vel_0'], box, bn)
        p2, v2, n2 = self.transition_model(p1, v1, box, bn)
        loss1 = self.weighted_mse_loss(p1, data['particles_pos_1'], n1)
        loss2 = self.weighted_mse_loss(p2, data['particles_pos_2'], n2)
        b1, b2 = self.cal_boundary_loss(p1), self.cal_boundary_loss(p2)
        return 0.5 * loss1 + 0.5 * loss2 + b1 + b2, dict(loss1=loss1, loss2=loss2, bloss1=b1, bloss2=b2)

    def train(self, max_steps=None):
        o = self.options
        self.transition_model.train()
        global_step, loss = self.start_step, None
        clip = o.TRAIN.get('grad_clip_value', 0)
        for epoch_idx in range(self.start_step, o.TRAIN.N_iters):
            order = np.random.permutation(len(self.dataset))        # DataLoader(batch_size=1, shuffle=True)
            if self.world > 1:        # data parallel: an epoch is still len(dataset) samples, split over the ranks
                order = order[:max(len(order) // self.world, 1)]          # (each rank draws its own permutation)
            for i in order:
                data = self._to_dev(self.dataset[int(i)])
                loss, parts = self.sample_loss(data)
                self.optimizer.zero_grad()
                loss.backward()
                nfdist.allreduce_grads(list(self.transition_model.parameters()), self.world)
                if clip != 0:
                    torch.nn.utils.clip_grad_norm_(self.transition_model.parameters(), clip)
                self.optimizer.step()
                if (global_step + 1) % o.TRAIN.log_interval == 0:
                    for k, v in parts.items():
                        self.summary_writer.add_scalar(k, v.item(), global_step)
                    self.summary_writer.add_scalar('loss', loss.item(), global_step)
                    self.summary_writer.add_scalar('lr', self.get_learning_rate(self.optimizer)[0], global_step)
                global_step += 1
                if max_steps is not None and global_step - self.start_step >= max_steps:
                    return loss
            if (epoch_idx + 1) % o.TRAIN.save_interval == 0:
                if self.rank == 0:
                    torch.save({'step': epoch_idx, 'model_state_dict': self.transition_model.state_dict(),
                                'optimizer_state_dict': portable_optimizer_state(self.optimizer)},
                               osp.join(self.exppath, 'models', f'{global_step}.pt'))
                self.eval(global_step)
        return loss

    def eval(self, step_idx, dump=True):
        """Rollout over the evaluation windows (trainer_transmodel.py:224-262)."""
        self.transition_model.eval()
        self.eval_count += 1
        dists = []
        self.fluid_error = FluidErrors()
        with torch.no_grad():
            for data_idx in range(len(self.test_dataset)):
                data = self._to_dev(self.test_dataset[data_idx])
                if data_idx == 0:
                    pos, vel = data['particles_pos_0'], data['particles_vel_0']
                pos, vel, _ = self.transition_model(pos, vel, data['box'], data['box_normals'])
                d = self.fluid_error.cal_errors(pos, data['particles_pos_1'], data_idx + 1)
                dists.append(d)
                self.summary_writer.add_scalar('pred2gt_distance', d, self.eval_count * len(self.test_dataset) + data_idx + 1)
                if dump and self.rank == 0:
                    pdir = osp.join(self.particlepath, f'{step_idx}')
                    os.makedirs(pdir, exist_ok=True)
                    for name, p, col in ((f'pred_{data_idx + 1}.obj', pos, (255, 0, 0)),
                                         (f'gt_{data_idx + 1}.obj', data['particles_pos_1'], (3, 168, 158))):
                        with open(osp.join(pdir, name), 'w') as fp:
                            record2obj(p, fp, color=col)
            self.summary_writer.add_scalar('avg_pred2gt_distance', float(np.mean(dists)) if dists else 0.0, step_idx)
        self.transition_model.train()
        return dists
