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


# ------------------------------------------------------------------------------------------ shared step controller
class _ProtocolLib:
    """A stand-in for libcde_mi355x.so that implements ONLY the control flow of the sharded adaptive entry points, on CPU
    memory (the pointers the host passes are real addresses of CPU tensors here), so that two gloo ranks can drive
    `shared_step_control` end to end -- forward (K4) and backward (K4a) -- without a GPU.  What it checks is what the
    kernels rely on: every launch is handed sums that are ALREADY the total over all ranks, in the right order
    (pending -> all-reduce -> launch; and for the backward launch -> pending -> all-reduce -> apply), the global batch
    size, one launch per all-reduce, and that both ranks stop after the same number of attempts."""

    N_ATTEMPTS = 7

    def __init__(self, rank, world):
        import ctypes
        self.ct, self.rank, self.world = ctypes, rank, world
        self.log = []

    # --- helpers
    def _doubles(self, p, n):
        return (self.ct.c_double * n).from_address(p.value if hasattr(p, "value") else p)

    def _status(self, ws, launched, stride, phase, n_accept):
        from torchcde_amd import _lib
        st = _lib.DopriStatus()
        st.phase, st.n_accept, st.n_reject = phase, n_accept, 0
        raw = bytes(st)
        base = ws.value + (launched & 1) * stride
        self.ct.memmove(base, raw, len(raw))

    # --- sizes
    def cde_dopri5_workspace_bytes(self, B, C, H, dt):
        return 4096

    def cde_dopri5_adjoint_workspace_bytes(self, B, C, H):
        return 8192

    def cde_dopri5_adjoint_status_stride(self):
        return 256

    def cde_dopri5_adjoint_reduced_count(self):
        return 8 + 2 * 16

    # --- forward: pending sums -> (all-reduce) -> one launch
    def cde_dopri5_pending_sums(self, ws, ws_bytes, B, C, H, dt, variant, act, launched, sums, stream):
        out = self._doubles(sums, 2)
        out[0], out[1] = (self.rank + 1) * (launched + 1), 10.0 * (self.rank + 1)
        self.log.append(("fwd_pending", launched))
        return 0

    def cde_dopri5_advance_sharded(self, *a):
        ws, launched, sums, global_batch = a[23], a[25], a[26], a[27]
        got = self._doubles(sums, 2)
        total_ranks = sum(range(1, self.world + 1))
        assert got[0] == total_ranks * (launched + 1) and got[1] == 10.0 * total_ranks, (self.rank, launched, got[0], got[1])
        assert global_batch == 11
        self.log.append(("fwd_launch", launched))
        from torchcde_amd import _lib
        size = self.ct.sizeof(_lib.DopriStatus)
        done = launched + 1 >= self.N_ATTEMPTS
        self._status(ws, launched + 1, size, 4 if done else 3, min(launched + 1, self.N_ATTEMPTS))   # (finished: the
        return 0                                               # real kernels keep copying the final block)

    # --- backward: one launch -> pending sums + images -> (all-reduce) -> apply
    def cde_dopri5_adjoint_advance(self, *a):
        ws, first, count, sums, global_batch = a[25], a[27], a[28], a[29], a[30]
        assert count == 1 and global_batch == 11
        if first == 0:
            assert sums is None or not getattr(sums, "value", sums), "the first launch of an interval has nothing pending"
        elif self.seminorm:
            assert self._doubles(sums, 8)[0] == sum(range(1, self.world + 1)) * first
        else:
            got = self._doubles(sums, 8 + 32)
            assert got[0] == sum(range(1, self.world + 1)) * first, (first, got[0])       # the PREVIOUS launch's sums
            assert got[8] == 100.0 * sum(range(1, self.world + 1)) and got[8 + 31] == got[8]  # ... and its images, reduced
        self.log.append(("bwd_launch", first))
        done = first + 1 >= self.N_ATTEMPTS
        self._status(ws, first + 1, 256, 4 if done else 3, min(first + 1, self.N_ATTEMPTS))
        return 0

    def cde_dopri5_adjoint_pending_sums(self, ws, ws_bytes, B, C, H, total, sums, stream):
        out = self._doubles(sums, 8 + 32)
        for i in range(8):
            out[i] = (self.rank + 1) * total
        for i in range(32):
            out[8 + i] = 100.0 * (self.rank + 1)
        self.log.append(("bwd_pending", total))
        return 0

    def cde_dopri5_adjoint_apply_reduced(self, ws, ws_bytes, B, C, H, rtol, atol, total, reduced, stream):
        got = self._doubles(reduced, 8 + 32)
        assert got[0] == sum(range(1, self.world + 1)) * total
        self.log.append(("bwd_apply", total))
        return 0

    def cde_dopri5_adjoint_finish(self, ws, ws_bytes, gw, gb, B, C, H, sharded, stream):
        assert sharded == (0 if self.seminorm else 1)       # "seminorm": the gradient images stayed local
        self.log.append(("finish", sharded))
        return 0

    # --- "seminorm": only the 8 state sums travel (both kernel families share the stand-in)
    seminorm = False

    def cde_dopri5_adjoint_state_sums(self, ws, ws_bytes, B, C, H, total, sums, stream):
        out = self._doubles(sums, 8)
        for i in range(8):
            out[i] = (self.rank + 1) * total
        self.log.append(("semi_pending", total))
        return 0

    def cde_dopri5_adjoint_apply_state_sums(self, ws, ws_bytes, B, C, H, total, reduced, stream):
        assert self._doubles(reduced, 8)[4] == sum(range(1, self.world + 1)) * total
        self.log.append(("semi_apply", total))
        return 0

    cde_dopri5_adjoint_mlp_state_sums = cde_dopri5_adjoint_state_sums
    cde_dopri5_adjoint_mlp_apply_state_sums = cde_dopri5_adjoint_apply_state_sums

    # --- the same protocol for the two-layer field (round 4): K4 / K4am under one controller
    MLP_REDUCED = 8 + 2 * 24

    def cde_dopri5_adjoint_mlp_workspace_bytes(self, B, C, H):
        return 4096 + 4 * (256 * 129 + 128 * 33) + 4096

    def cde_dopri5_adjoint_mlp_gradient_offset(self, B, C, H):
        return 4096

    def cde_dopri5_adjoint_mlp_gradient_upper_offset(self, B, C, H):
        return 0

    def cde_dopri5_adjoint_mlp_reduced_count(self):
        return self.MLP_REDUCED

    def cde_dopri5_pending_sums_mlp(self, ws, ws_bytes, B, C, H, dt, launched, sums, stream):
        out = self._doubles(sums, 2)
        out[0], out[1] = (self.rank + 1) * (launched + 1), 10.0 * (self.rank + 1)
        self.log.append(("mlp_fwd_pending", launched))
        return 0

    def cde_dopri5_advance_mlp_sharded(self, *a):
        width, ws, launched, sums, global_batch = a[6], a[25], a[27], a[28], a[29]
        got = self._doubles(sums, 2)
        total_ranks = sum(range(1, self.world + 1))
        assert width == 8 and global_batch == 11
        assert got[0] == total_ranks * (launched + 1) and got[1] == 10.0 * total_ranks, (self.rank, launched, got[0], got[1])
        self.log.append(("mlp_fwd_launch", launched))
        from torchcde_amd import _lib
        done = launched + 1 >= self.N_ATTEMPTS
        self._status(ws, launched + 1, self.ct.sizeof(_lib.DopriStatus), 4 if done else 3, min(launched + 1, self.N_ATTEMPTS))
        return 0

    def cde_dopri5_adjoint_mlp_advance_sharded(self, *a):
        width, ws, first, sums, global_batch = a[6], a[28], a[30], a[31], a[32]
        assert width == 8 and global_batch == 11
        if first == 0:
            assert sums is None or not getattr(sums, "value", sums), "the first launch of an interval has nothing pending"
        elif self.seminorm:
            assert self._doubles(sums, 8)[0] == sum(range(1, self.world + 1)) * first
        else:
            got = self._doubles(sums, self.MLP_REDUCED)
            assert got[0] == sum(range(1, self.world + 1)) * first, (first, got[0])
            assert got[8] == 100.0 * sum(range(1, self.world + 1)) and got[self.MLP_REDUCED - 1] == got[8]
        self.log.append(("mlp_bwd_launch", first))
        done = first + 1 >= self.N_ATTEMPTS
        self._status(ws, first + 1, 256, 4 if done else 3, min(first + 1, self.N_ATTEMPTS))
        return 0

    def cde_dopri5_adjoint_mlp_pending_sums(self, ws, ws_bytes, B, C, H, total, sums, stream):
        out = self._doubles(sums, self.MLP_REDUCED)
        for i in range(8):
            out[i] = (self.rank + 1) * total
        for i in range(8, self.MLP_REDUCED):
            out[i] = 100.0 * (self.rank + 1)
        self.log.append(("mlp_bwd_pending", total))
        return 0

    def cde_dopri5_adjoint_mlp_apply_reduced(self, ws, ws_bytes, B, C, H, rtol, atol, total, reduced, stream):
        got = self._doubles(reduced, self.MLP_REDUCED)
        assert got[0] == sum(range(1, self.world + 1)) * total
        self.log.append(("mlp_bwd_apply", total))
        return 0


def _shared_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import ctypes
    import importlib
    import types
    from torchcde_amd import _lib
    from torchcde_amd.distributed import shared_step_control, shard_bounds
    front = importlib.import_module("torchcde_amd.cdeint")
    fake = _ProtocolLib(rank, world)
    _lib.load = lambda: fake
    _lib.stream_ptr = lambda device: ctypes.c_void_p(0)
    lo, hi = shard_bounds(11)
    n = hi - lo
    path = types.SimpleNamespace(_native_inputs=lambda: (torch.zeros(n, 5, 3), torch.arange(5.), (n,)),
                                 _n_intervals=lambda: 4, _degree=_lib.PATH_LINEAR)
    field = types.SimpleNamespace(act=_lib.ACT_NONE, kind="affine")
    t = torch.tensor([0., 4.])
    with shared_step_control(11):
        plan = front._Dopri5Plan(path, field, (n,), 4, 3, t, 1e-4, 1e-6, dict(jump_t=torch.arange(5.)))
    w, b = torch.zeros(12, 4), torch.zeros(12)
    out = plan.run(torch.zeros(n, 4), w, b)
    assert front.last_dopri5_stats["n_accept"] == fake.N_ATTEMPTS
    plan.run_adjoint(out, torch.ones(n, 2, 4), w, b)
    assert front.last_dopri5_adjoint_stats["launches"] % front._DOPRI_CHUNK == 0
    # ... and the two-layer field (K4 / K4am): the same order of calls, with its own entry points
    hidden = types.SimpleNamespace(weight=torch.zeros(8, 4), bias=torch.zeros(8))
    field2 = types.SimpleNamespace(act=_lib.ACT_TANH, kind="mlp2", hidden=hidden)
    with shared_step_control(11):
        plan2 = front._Dopri5Plan(path, field2, (n,), 4, 3, t, 1e-4, 1e-6, None)
    w2, b2 = torch.zeros(12, 8), torch.zeros(12)
    out2 = plan2.run(torch.zeros(n, 4), w2, b2)
    assert front.last_dopri5_stats["n_accept"] == fake.N_ATTEMPTS
    grads = plan2.run_adjoint_mlp(out2, torch.ones(n, 2, 4), hidden.weight, hidden.bias, w2, b2)
    assert grads[1].shape == (8, 4) and grads[3].shape == (12, 8)
    assert front.last_dopri5_adjoint_stats["launches"] % front._DOPRI_CHUNK == 0
    # ... and both again with adjoint_options=dict(norm="seminorm"): 8 doubles per attempted step, images local
    fake.seminorm = True
    with shared_step_control(11):
        plan3 = front._Dopri5Plan(path, field, (n,), 4, 3, t, 1e-4, 1e-6, None, adjoint_options=dict(norm="seminorm"))
        plan4 = front._Dopri5Plan(path, field2, (n,), 4, 3, t, 1e-4, 1e-6, None, adjoint_options=dict(norm="seminorm"))
    plan3.run_adjoint(out, torch.ones(n, 2, 4), w, b)
    plan4.run_adjoint_mlp(out2, torch.ones(n, 2, 4), hidden.weight, hidden.bias, w2, b2)
    torch.save(fake.log, tmp + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_shared_step_control_protocol_on_two_gloo_ranks(tmp_path):
    """two gloo ranks drive `shared_step_control` END TO END through the host code that runs on
    the GPUs -- `_Dopri5Plan.run` (K4) and `.run_adjoint` (K4a, incl. the all-reduce of the gradient images the mixed
    adjoint norm needs) -- against a protocol stand-in for the shared library (the kernels themselves: -m gpu tests, where
    two shards run in lock step on one GPU).  The stand-in asserts inside every launch that the sums it is handed are the
    all-reduced ones; here: both ranks made the same calls in the same order."""
    tmp = str(tmp_path / "shared")
    port = 29500 + ((os.getpid() + 7) % 2000)
    mp.spawn(_shared_worker, args=(2, port, tmp), nprocs=2, join=True)
    logs = [torch.load(tmp + ".%d" % r) for r in range(2)]
    assert logs[0] == logs[1]
    kinds = [k for k, _ in logs[0]]
    n = _ProtocolLib.N_ATTEMPTS
    first_fwd = kinds.index("fwd_pending")
    assert kinds[first_fwd:first_fwd + 2 * n] == ["fwd_pending", "fwd_launch"] * n        # pending -> reduce -> launch
    first_bwd = kinds.index("bwd_launch")
    assert kinds[first_bwd:first_bwd + 3 * n] == ["bwd_launch", "bwd_pending", "bwd_apply"] * n
    assert "finish" in kinds
    # the two-layer field: pending -> reduce -> launch forward; launch -> pending -> reduce -> apply backward
    first_fwd = kinds.index("mlp_fwd_pending")
    assert kinds[first_fwd:first_fwd + 2 * n] == ["mlp_fwd_pending", "mlp_fwd_launch"] * n
    first_bwd = kinds.index("mlp_bwd_launch")
    assert kinds[first_bwd:first_bwd + 3 * n] == ["mlp_bwd_launch", "mlp_bwd_pending", "mlp_bwd_apply"] * n
    # "seminorm": launch -> 8 state sums -> reduce -> apply, for the one-layer and then the two-layer plan
    first = kinds.index("semi_pending") - 1
    assert kinds[first:first + 3 * n] == ["bwd_launch", "semi_pending", "semi_apply"] * n
    second = len(kinds) - 1 - kinds[::-1].index("mlp_bwd_launch") - 3 * (n - 1)
    assert kinds[second:second + 3 * n] == ["mlp_bwd_launch", "semi_pending", "semi_apply"] * n
