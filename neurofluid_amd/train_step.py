"""One optimiser step of the warm-up trainer (/root/reference/trainer/trainer_renderer.py:75-143):
for each of the 4 warm-up views sample `ray_chunk` random pixels of frame 0 (centre crop for the first
`precrop_iters` steps, trainer/basetrainer.py:171-193), render them from the GT particles, loss = sum over views
of MSE(rgb0) + MSE(rgb1); zero_grad / backward / Adam / ExponentialLR (utils/lr_schedulers.py:3-12).
Used by bench.py --workload train and by the Trainer in neurofluid_amd/trainers.py."""
import numpy as np
import torch

from . import dist as nfdist


_COORDS = {}


def _upload(t, device):
    """Host -> device copy that does not synchronise the stream: pinned staging + non_blocking (a pageable .to(device)
    waits for everything enqueued before it, i.e. for the previous optimiser step)."""
    if device.type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def random_sample_coords(H, W, global_step, precrop_iters):
    """trainer/basetrainer.py:171-193 (CPU tensors, like the reference's device-less meshgrid).  The grid only
    has two variants (centre crop / full frame), so it is built once instead of once per view per step."""
    key = (H, W, global_step > precrop_iters)
    if key not in _COORDS:
        _COORDS[key] = _build_coords(H, W, global_step, precrop_iters)
    return _COORDS[key]


def _build_coords(H, W, global_step, precrop_iters):
    if global_step > precrop_iters:
        ys, xs = torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W)
    else:
        dH, dW = int(H // 2 * 0.5), int(W // 2 * 0.5)
        ys = torch.linspace(H // 2 - dH, H // 2 + dH - 1, 2 * dH)
        xs = torch.linspace(W // 2 - dW, W // 2 + dW - 1, 2 * dW)
    coords = torch.stack(torch.meshgrid(ys, xs, indexing="ij"), -1)
    return coords.reshape(-1, 2)


class PixelSampler:
    """Draws the per-view pixel selections `rng.choice(n, ray_chunk, replace=False)` (trainer_renderer.py:119) in the
    reference's order, one step ahead on a background thread: the draw is a full 160 000-element shuffle (1.4 ms per
    view on the host) and would otherwise sit between two GPU steps.  Same RNG stream, same indices.
    The read-ahead is invisible to the stream's other users: every draw records the generator state it started from,
    and close() joins the worker and rewinds `rng` to the state before the first selection nobody consumed, so after
    train() the stream is exactly where a sampler without read-ahead would have left it.  An exception in the worker
    (e.g. fewer pixels than ray_chunk) is re-raised by next()."""

    def __init__(self, rng, n_views, ray_chunk, n_pixels_of_step, first_step, depth=2):
        import queue
        import threading
        self.rng, self.n_views, self.ray_chunk, self.n_of = rng, n_views, ray_chunk, n_pixels_of_step
        self.q = queue.Queue(maxsize=depth)
        self.step = first_step
        self._stop = threading.Event()
        self._pending = None          # (step, sels, state_before) drawn but not yet queued when the stop flag was seen
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        import queue
        step = self.step
        try:
            while not self._stop.is_set():
                state = self.rng.get_state()
                n = self.n_of(step)
                sels = [self.rng.choice(n, size=[self.ray_chunk], replace=False) for _ in range(self.n_views)]
                item = (step, sels, state)
                while True:
                    if self._stop.is_set():
                        self._pending = item
                        return
                    try:
                        self.q.put(item, timeout=0.05)
                        break
                    except queue.Full:
                        continue
                step += 1
        except BaseException as e:          # surfaced by next(); never leaves the consumer blocked
            self.q.put(("error", e, None))

    def next(self, step):
        import queue
        while True:
            try:
                s, sels, _ = self.q.get(timeout=1.0)
                break
            except queue.Empty:
                if not self.t.is_alive() and self.q.empty():
                    raise RuntimeError("PixelSampler worker stopped without producing a selection")
        if s == "error":
            raise sels
        assert s == step, "PixelSampler is strictly sequential"
        return sels

    def close(self):
        self._stop.set()
        self.t.join(timeout=10.0)
        left = []
        try:
            while True:
                left.append(self.q.get_nowait())
        except Exception:
            pass
        if self._pending is not None:
            left.append(self._pending)
        states = [it[2] for it in left if it[0] != "error" and it[2] is not None]
        if states:                          # rewind to before the first unconsumed draw
            self.rng.set_state(states[0])


class ExponentialLR(torch.optim.lr_scheduler.LambdaLR):
    """lr = base_lr * gamma ** (epoch / decay_epochs)  (utils/lr_schedulers.py:3-12)."""

    def __init__(self, optimizer, decay_epochs, gamma=0.1, last_epoch=-1):
        super().__init__(optimizer, lambda e: gamma ** (e / decay_epochs), last_epoch)


def renderer_train_step(renderer, optimizer, scheduler, particles, views, H, W, step_idx, ray_chunk=1024,
                        precrop_iters=500, rng=np.random, rank=0, world=1, sampler=None):
    """views: list of dicts {cw (3,4), rays (H,W,6), rgb (H*W,3)} on the GPU.  Returns the loss tensor.
    The reference renders the views one after the other; rays are independent, so the views are batched into ONE
    renderer call (per-ray camera position) and the per-view MSEs are taken on slices — same loss, 4x fewer launches.
    The pixel RNG is drawn per view in the reference's order (np.random.choice, trainer_renderer.py:119)."""
    rays_l, rgbs_l, ro_l = [], [], []
    sels = sampler.next(step_idx) if sampler is not None else None
    for vi, v in enumerate(views):
        coords = random_sample_coords(H, W, step_idx, precrop_iters)
        sel = sels[vi] if sels is not None else rng.choice(coords.shape[0], size=[ray_chunk], replace=False)
        sc = _upload(coords[sel].long(), v["rays"].device)
        rays_l.append(v["rays"][sc[:, 0], sc[:, 1]])
        rgbs_l.append(v["rgb"].view(H, W, -1)[sc[:, 0], sc[:, 1]])
        ro_l.append(renderer.set_ro(v["cw"]).expand(ray_chunk, 3))
    out = renderer(particles, torch.cat(ro_l).contiguous(), torch.cat(rays_l), None, None)
    total = 0.
    for i, rgbs in enumerate(rgbs_l):
        sl = slice(i * ray_chunk, (i + 1) * ray_chunk)
        loss = torch.nn.functional.mse_loss(out["rgb0"][sl], rgbs)
        if renderer.N_importance > 0:
            loss = loss + torch.nn.functional.mse_loss(out["rgb1"][sl], rgbs)
        total = total + loss
    optimizer.zero_grad()
    total.backward()
    nfdist.allreduce_grads(list(renderer.parameters()), world)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return total.detach()


def make_train_step(net, scene, dev, rank=0, world=1, lr=5e-4, decay_epochs=10000, seed=10):
    """Synthetic warm-up workload for bench.py: 4 views (the synthetic camera), random target colours."""
    H = W = 400
    g = torch.Generator().manual_seed(seed + rank)
    rays = scene["rays"].view(H, W, 6).to(dev)
    cw = scene["c2w"].to(dev)
    views = [dict(cw=cw, rays=rays, rgb=torch.rand(H * W, 3, generator=g).to(dev)) for _ in range(4)]
    P = scene["P"].to(dev)
    for p in net.parameters():
        p.requires_grad_(True)
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    sched = ExponentialLR(opt, decay_epochs=decay_epochs, gamma=0.1)
    rng = np.random.RandomState(seed + rank)
    state = {"step": 1000}   # past precrop_iters: full-frame sampling (steady state of the 100k-step schedule)
    sampler = PixelSampler(rng, len(views), 1024, lambda s: random_sample_coords(H, W, s, 500).shape[0], state["step"])

    def step():
        loss = renderer_train_step(net, opt, sched, P, views, H, W, state["step"], 1024, 500, rng, rank, world, sampler)
        state["step"] += 1
        return loss

    return step
