"""The coupled per-frame body of the reference's e2e loop (/root/reference/eval_e2e.py:58-134: transition step, then render of the predicted
particles) with the transition step of frame t + 1 in flight on its own HIP stream while frame t renders.

The step depends on the previous STATE only, never on an image, and it is small (0.18 ms of kernels for 4 913 particles, replicated on every
rank of a multi-GPU run), while the renderer's two MLP launches are persistent grids whose last round leaves most of the chip idle (tile
quantisation: DESIGN section 5e).  Enqueued a frame ahead on a side stream, the step's kernels run in those tails instead of in front of the
next frame's first kernel, and the host's wait for the step's completion word (ParticleNet.step_async / AsyncStep.result) finds it long set.
Same kernels, same operands, same order per step: the states are bit-identical to the sequential loop's
(tests/test_gpu_trans.py::test_lookahead_rollout_bit_equal_to_sequential).

Under the lookahead the transition module's per-call side results (`pn._y3`, `pn.conv0_fluid.nns`, ...) alias scratch that the NEXT step is
already overwriting when next_state() returns: only the returned (pos, vel, num_neighbors) are valid; a caller that needs the neighbour
lists of a frame uses the sequential `pn(...)` call."""
import torch


class CoupledRollout:
    def __init__(self, transition_model, box, box_feats, device=None):
        self.pn, self.box, self.box_feats = transition_model, box, box_feats
        self.side = torch.cuda.Stream(device=device if device is not None else box.device)
        self._pending = None
        self._seen = {}

    def start(self, pos, vel):
        """(Re)start from a state: the step that produces the FIRST frame's particles is enqueued now."""
        self.drop()
        self._seen.clear()
        self._then_seen((pos, vel))
        self._pending = self.pn.step_async(pos, vel, self.box, self.box_feats, stream=self.side)

    def drop(self):
        """Forget the step in flight (it is consumed: the module's scratch and flag words must be at rest before the next one)."""
        if self._pending is not None:
            self._pending.result()
            self._pending = None

    def next_state(self, then=None, last=False):
        """(pos, vel, num_neighbors) of the next frame.  The step AFTER it is enqueued at once, from this state — or from `then` = (pos, vel)
        when the caller knows the rollout restarts there (a benchmark that returns to the initial cloud every few frames).
        last=True: this is the rollout's final frame — nothing is enqueued behind it (start() begins a new rollout)."""
        if self._pending is None:
            raise RuntimeError("CoupledRollout.next_state before start()")
        pos, vel, nn = self._pending.result()
        if last:
            self._pending = None
            return pos, vel, nn
        src = (pos, vel) if then is None else then
        # the next step's inputs were produced on the side stream itself (or, `then`, before this call): it must NOT wait for the frame the
        # caller's stream is still rendering — it is meant to run beside it
        self._pending = self.pn.step_async(src[0], src[1], self.box, self.box_feats, stream=self.side, wait_current=then is not None and not self._then_seen(then))
        return pos, vel, nn

    def _then_seen(self, then):
        """A restart state the rollout has been given before (the SAME tensor objects, unmodified) is complete on every stream by now; a
        NEW one is waited for once.  The seen states are held by strong reference: an address-based key alone could be recycled by the
        allocator for a fresh tensor (`P0.clone()` per cycle) that the caller's stream is still writing — it would then be taken for
        complete and read by the side stream too early.  At most 8 states are remembered (an unseen state only costs one event wait)."""
        key = (id(then[0]), then[0]._version, id(then[1]), then[1]._version)
        hit = self._seen.get(key)
        if hit is not None and hit[0] is then[0] and hit[1] is then[1]:
            return True
        if len(self._seen) >= 8:
            self._seen.clear()
        self._seen[key] = (then[0], then[1])
        return False
