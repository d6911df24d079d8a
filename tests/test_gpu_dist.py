"""RCCL on the box: the N > 1 data path of bench.py / the trainers is `backend="nccl"` (= RCCL); the multi-rank logic is
covered by the gloo tests (tests/test_dist_gloo.py, test_tile_sharded_eval_two_ranks_one_gpu), this one checks that the
collectives the path uses actually run through RCCL on an MI355X of this image (a 1-rank group: a second rank on the same
device is refused by RCCL, and the boxes have one GPU)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_single_rank_collectives():
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        from neurofluid_amd import dist as nfdist
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        assert dist.get_backend() == "nccl"
        x = torch.arange(12, dtype=torch.float32, device="cuda").view(4, 3)
        out = torch.empty(4, 3, device="cuda")
        dist.all_gather_into_tensor(out, x.contiguous())          # the RGB-tile gather of render_image
        assert torch.equal(out, x)
        flat = torch.ones(1 << 20, device="cuda")
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)                 # the flat gradient bucket of allreduce_grads
        assert float(flat.sum()) == float(1 << 20)
        t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                    # bench.py's max-over-ranks step time
        dist.barrier()
        torch.cuda.synchronize()
        # the same collectives through the package's helpers at world = 1 (early-outs) and a forced 1-rank gather
        assert torch.equal(nfdist.gather_chunks(x, 1, 4, 4, 0, 1), x)
        dist.destroy_process_group()
        print("rccl ok")
    """ % ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
