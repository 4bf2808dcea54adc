"""world_size-2 gloo test of the multi-GPU path's host logic: batch sharding, parameter-gradient all-reduce and
result gathering reproduce the single-process answer (per-shard compute is the CPU oracle here -- the sharding
logic is what is under test; the kernels themselves are covered by the -m gpu tests)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle import cde, interp
    from helpers import LinearField, make_series
    from torchcde_amd.distributed import shard, shard_bounds, allreduce_gradients, gather_batch

    B, L, C, H = 11, 9, 3, 4                      # ragged split: 6 + 5
    x = make_series(B, L, C, torch.float64, seed=5)
    z0 = torch.randn(B, H, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    func = LinearField(H, C, torch.float64, scale=0.5, seed=5)
    lo, hi = shard_bounds(B)
    assert (hi - lo) == (6 if rank == 0 else 5)
    X = interp.CubicPath(interp.hermite_bdiff_coeffs(shard(x)))
    z = shard(z0).clone().requires_grad_(True)
    out = cde.cdeint(X, func, z, X.interval, adjoint=True, method="rk4", options=dict(step_size=1.0))
    out[:, -1].sum().backward()
    allreduce_gradients(list(func.parameters()))
    full = gather_batch(out.detach(), B)
    gz = gather_batch(z.grad, B)
    # shared step control of the adaptive solves: the default reducer is a sum all-reduce of the pending error sums
    from torchcde_amd.distributed import shared_step_control, step_control
    assert step_control() is None
    with shared_step_control(B):
        reduce, global_batch = step_control()
        sums = torch.tensor([1.0 + rank, 10.0 * (rank + 1)], dtype=torch.float64)
        reduce(sums)
        assert global_batch == B and torch.equal(sums, torch.tensor([3.0, 30.0], dtype=torch.float64))
    assert step_control() is None
    if rank == 0:
        torch.save(dict(out=full, gz=gz, gw=func.linear.weight.grad, gb=func.linear.bias.grad), tmp)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_solve_equals_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import cde, interp
    from helpers import LinearField, make_series
    tmp = str(tmp_path / "dist.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, tmp), nprocs=2, join=True)
    got = torch.load(tmp)
    B, L, C, H = 11, 9, 3, 4
    x = make_series(B, L, C, torch.float64, seed=5)
    z0 = torch.randn(B, H, dtype=torch.float64, generator=torch.Generator().manual_seed(5)).requires_grad_(True)
    func = LinearField(H, C, torch.float64, scale=0.5, seed=5)
    X = interp.CubicPath(interp.hermite_bdiff_coeffs(x))
    out = cde.cdeint(X, func, z0, X.interval, adjoint=True, method="rk4", options=dict(step_size=1.0))
    out[:, -1].sum().backward()
    assert torch.allclose(got["out"], out.detach(), rtol=1e-12, atol=1e-14)
    assert torch.allclose(got["gz"], z0.grad, rtol=1e-12, atol=1e-14)
    assert torch.allclose(got["gw"], func.linear.weight.grad, rtol=1e-10, atol=1e-13)
    assert torch.allclose(got["gb"], func.linear.bias.grad, rtol=1e-10, atol=1e-13)


def test_shard_bounds_cover_everything():
    from torchcde_amd.distributed import shard_bounds
    for n in (0, 1, 7, 32768, 262144 + 3):
        for world in (1, 2, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
