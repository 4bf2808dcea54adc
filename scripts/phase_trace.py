"""Where does an attempt's time go?  ("split the 39 us first".)

Runs the config-4 shard (B series, L = 128, C = 8, H = 32, dopri5, jump_t on the knots) with the INSTRUMENTED library
(CDE_PHASE_TRACE=1 -> libcde_mi355x_trace.so, csrc/cde_common.h "phase trace"), reads the stamp ring of the attempt kernel
and prints, for the steady-state attempts still in the ring, the time line of one attempt on the chip's common 100 MHz clock:

    gap      last workgroup of attempt n-1 done  ->  first workgroup of attempt n running   (the launch boundary)
    skew     first -> last workgroup start
    phases   per workgroup, median / max over workgroups
    drain    median workgroup done -> last workgroup done

    CDE_PHASE_TRACE=1 python scripts/phase_trace.py [k4|k4a] [batch]
"""
import ctypes
import os
import sys

os.environ["CDE_PHASE_TRACE"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchcde_amd as cde  # noqa: E402
from torchcde_amd import _lib  # noqa: E402
from helpers import LinearField, make_series  # noqa: E402

SLOTS, BLOCKS, RING = 40, 512, 32
NAMES = {
    "k4": ["entry", "ctrl loaded", "images+knots in LDS (barrier)", "pending sums reduced", "controller + scalars",
           "state in registers / outputs emitted", "stage 1", "stage 2", "stage 3", "stage 4", "stage 5", "stage 6",
           "error + stores issued", "workgroup sum"],
    "k4a": ["entry", "ctrl loaded", "pending sums reduced", "controller + scalars", "weights in registers / first state requested",
            "tile 1 (7 stages)", "tile 2", "tile 3", "tile 4", "tile 5", "tile 6", "tile 7", "tile 8", "helper images stored (join)",
            "workgroup sum"],
    "k4am": ["entry", "LDS images + pending sums reduced", "controller + scalars", "state / W1^T / first row in registers",
             "stage 1", "stage 2", "stage 3", "stage 4", "stage 5", "stage 6", "stage 7", "error sums + state stores",
             "workgroup sum"],
}


def read_ring(which):
    _lib.build()                                      # the instrumented library (a no-op when it is up to date)
    lib = _lib.load()
    fn = getattr(lib, "cde_debug_%s_phase_trace" % which)
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]
    buf = np.zeros((RING, BLOCKS, SLOTS), dtype=np.uint64)
    torch.cuda.synchronize()
    rc = fn(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
    return buf.astype(np.int64)


def report(which, ring, n_blocks):
    names = NAMES[which]
    n = len(names)
    ring = ring[:, :n_blocks, :n]
    live = [r for r in range(RING) if (ring[r, :, 0] > 0).all() and (ring[r, :, n - 1] > 0).all()]
    # order the ring entries by time, drop the first (may be a partially overwritten or warm-up attempt)
    live.sort(key=lambda r: ring[r, :, 0].min())
    rows = []
    for prev, cur in zip(live[:-1], live[1:]):
        a, b = ring[prev], ring[cur]
        start, end = b[:, 0], b[:, n - 1]
        if start.min() < a[:, n - 1].max():          # not consecutive launches of one stream: skip
            continue
        gap = start.min() - a[:, n - 1].max()
        skew = start.max() - start.min()
        phases = np.diff(b, axis=1)                   # (blocks, n - 1)
        rows.append(dict(gap=gap, skew=skew, med=np.median(phases, axis=0), mx=phases.max(axis=0),
                         total=end.max() - start.min(), drain=end.max() - np.median(end),
                         period=end.max() - a[:, n - 1].max(), body_med=np.median(end - start)))
    if not rows:
        print("no consecutive attempts in the ring")
        return
    us = 0.01                                         # 100 MHz ticks -> microseconds
    print("%s: %d consecutive attempts in the ring, %d workgroups; microseconds (100 MHz clock: +-0.01)" % (which, len(rows), n_blocks))
    print("  period (end of previous attempt's last workgroup -> end of this one's)  median %.2f  min %.2f  max %.2f"
          % tuple(us * f([r["period"] for r in rows]) for f in (np.median, np.min, np.max)))
    print("  launch gap                                   median %.2f" % (us * np.median([r["gap"] for r in rows])))
    print("  start skew over workgroups                   median %.2f" % (us * np.median([r["skew"] for r in rows])))
    print("  one workgroup, entry -> done                 median %.2f" % (us * np.median([r["body_med"] for r in rows])))
    print("  drain (median workgroup done -> last done)   median %.2f" % (us * np.median([r["drain"] for r in rows])))
    med = np.median(np.stack([r["med"] for r in rows]), axis=0)
    mx = np.median(np.stack([r["mx"] for r in rows]), axis=0)
    for i in range(n - 1):
        print("    -> %-40s median %6.2f   slowest workgroup %6.2f" % (names[i + 1], us * med[i], us * mx[i]))


def report_stage(ring, n_blocks):
    """K4a, second tile of each workgroup, stage 4 of 7: chain wave 0 (slots 15..19) and helper wave 0 (20..24)."""
    us = 0.01
    rows = [r for r in range(RING) if (ring[r, :n_blocks, 15] > 0).all() and (ring[r, :n_blocks, 24] > 0).all()]
    if not rows:
        return
    t = ring[rows][:, :n_blocks, :]                      # (attempts, blocks, slots)
    base = t[:, :, 15:16]
    def med(slot):
        return us * np.median(t[:, :, slot] - base[:, :, 0])
    print("  inside one stage (tile 2, stage 4), time since the chain wave left the previous stage's barrier; median over workgroups and attempts")
    for name, slot in (("chain : Y product complete", 16), ("chain : activation, contraction, next state and g tile published", 17),
                       ("chain : v product complete", 18), ("chain : through the stage barrier", 19),
                       ("helper: left the previous barrier", 20), ("helper: g tile and z in registers", 21),
                       ("helper: image MFMAs complete", 22), ("helper: at the stage barrier", 23), ("helper: through the stage barrier", 24)):
        print("    %-70s %6.2f" % (name, med(slot)))


def report_eval(ring, n_blocks):
    """K4am: inside the evaluation of stage 4 (slots 13..16 since slot 6 = end of stage 3)."""
    us = 0.01
    rows = [r for r in range(RING) if (ring[r, :n_blocks, 13] > 0).all() and (ring[r, :n_blocks, 16] > 0).all()]
    if not rows:
        return
    t = ring[rows][:, :n_blocks, :]
    print("  inside the evaluation of stage 4, wave 0 (median over workgroups and attempts):")
    for name, a, b in (("stage combination + control -> layer 1 done", 6, 13), ("layer 2 / activation / dL/dY2 / gu (this wave's groups)", 13, 14),
                       ("all-reduce of gu, f, kt over the four waves", 14, 15), ("dL/dY1, va = W1^T dL/dY1", 15, 16),
                       ("slopes stored, end of stage", 16, 7)):
        print("    %-62s %6.2f" % (name, us * np.median(t[:, :, b] - t[:, :, a])))


def main_k4am(B):
    from helpers import TwoLayerField
    dev = torch.device("cuda", 0)
    x = make_series(B, 128, 8, seed=0).to(dev)
    X = cde.CubicSpline(cde.hermite_cubic_coefficients_with_backward_differences(x))
    func = TwoLayerField(32, 8, 128, seed=0).to(dev)
    z = torch.randn(B, 32, generator=torch.Generator().manual_seed(0)).to(dev).requires_grad_(True)
    out = cde.cdeint(X, func, z, X.interval, adjoint_options=dict(norm="seminorm"))
    out[:, -1].sum().backward()
    torch.cuda.synchronize()
    n_blocks = min(BLOCKS, (B + 15) // 16)
    ring = read_ring("k4am")
    report("k4am", ring, n_blocks)
    report_eval(ring, n_blocks)


def main():
    _lib.build()
    which = sys.argv[1] if len(sys.argv) > 1 else "k4"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    if which == "k4am":
        return main_k4am(B)
    dev = torch.device("cuda", 0)
    x = make_series(B, 128, 8, seed=0).to(dev)
    X = cde.LinearInterpolation(cde.linear_interpolation_coeffs(x))
    func = LinearField(32, 8, scale=0.25, seed=0).to(dev)
    z = torch.randn(B, 32, generator=torch.Generator().manual_seed(0)).to(dev).requires_grad_(which == "k4a")
    extra = dict(adjoint_options=dict(norm="seminorm", jump_t=X.grid_points)) if which == "k4a" else {}
    for _ in range(2):
        out = cde.cdeint(X, func, z, X.interval, options=dict(jump_t=X.grid_points), **extra)
        if which == "k4a":
            out[:, -1].sum().backward()
    torch.cuda.synchronize()
    n_blocks = (B + 127) // 128 if which == "k4" else min(256, (B + 15) // 16)
    ring = read_ring(which)
    report(which, ring, min(n_blocks, BLOCKS))
    if which == "k4a":
        report_stage(ring, min(n_blocks, BLOCKS))


if __name__ == "__main__":
    main()
