"""`sharding.py`'s collectives over the REAL backend (RCCL) with a one-rank process group on the test box's single GPU.
The world-size-2 / 8 tests run over gloo (two RCCL ranks cannot share one GPU); this closes the other half: the `nccl` code path
itself — communicator set-up on the MI355X, device-side gather / all-gather-into-tensor — executes, in a child process (a process
group is process-global state)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, socket
sys.path.insert(0, {root!r})
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from bsdf_diffusion_sampling_amd.sharding import gather_to_root, all_gather, pack_result, ShardedPlugin, shard_range
from bsdf_diffusion_sampling_amd.brdf_measured_disk import MyBSDF
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{{port}}", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
n = 100003
g = torch.Generator().manual_seed(1)
u = torch.rand(n, 2, generator=g)
r, a = 0.9 * torch.sqrt(u[:, 0]), 6.2831853 * u[:, 1]
wi = torch.stack([r * torch.cos(a), r * torch.sin(a), torch.sqrt(1 - r * r)], 1).float().to(dev)
plug = MyBSDF({{"filename": "chm_orange_rgb", "measured": False}})
sp = ShardedPlugin(plug)
assert sp.world == 1 and sp.local_range(n) == shard_range(n, 0, 1) == (0, n)
wo, pdf = sp.sample_local(wi, n, seed=5)
packed = pack_result(wo, pdf)
full = gather_to_root(packed, n)                       # dist.gather over RCCL, device tensors
assert torch.equal(full, packed)
assert torch.equal(all_gather(packed, n), packed)      # dist.all_gather_into_tensor over RCCL
t = torch.tensor([3.0], device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert t.item() == 3.0
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK", torch.isfinite(full).all().item())
"""


def test_sharding_collectives_over_rccl_with_one_rank():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK True" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
