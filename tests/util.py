"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

from morpheus_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL_FLOOR = 1e-3   # rel = |a-b| / max(|b|, REL_FLOOR)   (SURVEY 8d "state the floor")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def max_rel(a, b, floor=REL_FLOOR):
    a = torch.as_tensor(a).detach().double().reshape(-1).cpu()
    b = torch.as_tensor(b).detach().double().reshape(-1).cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(((a - b).abs() / b.abs().clamp(min=floor)).max())


def assert_close(a, b, tol, what="", floor=REL_FLOOR):
    r = max_rel(a, b, floor)
    assert r <= tol, f"{what}: max rel err {r:.3e} > {tol:.1e}"


def assert_close_counted(a, b, what="", tol=1e-4, floor=REL_FLOOR, max_tol=3e-4, max_frac=1.0 / 512.0, min_count=4):
    """The 1e-4 bar AT SURVEY 8(d)'s floor (rel = |a-b| / max(|b|, 1e-3)), stated as what fp32 can keep: every element within
    `max_tol` (3e-4), and at most max(min_count, ceil(n * max_frac)) elements above `tol` (default 0.2 %: measured 0..2 of 4 096
    strided SDF samples on the evaluation renders, worst 1.6e-4 -- profiles/r04_parity_f64.jsonl -- and 4 of 4 096 on the
    pose-optimising training render, worst 2.6e-4).  Used for the SDF only; the allowance is not fitted: assert_close_vs_f64
    below derives it from the reference's own fp32 error against its double run.  The elements above 1e-4 are reference
    values below ~1e-3 in magnitude (an SDF sample next to its zero crossing, the opacity / depth of a ray that only grazes
    the box) whose ABSOLUTE error is one or two fp32 ulps of the O(1) quantities that cancel to them;
    the gate counts them instead of moving the floor."""
    a = torch.as_tensor(a).detach().double().reshape(-1).cpu()
    b = torch.as_tensor(b).detach().double().reshape(-1).cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    rel = (a - b).abs() / b.abs().clamp(min=floor)
    n_bad = int((rel > tol).sum())
    allowed = max(min_count, int(np.ceil(a.numel() * max_frac)))
    worst = float(rel.max()) if rel.numel() else 0.0
    assert worst <= max_tol, f"{what}: max rel err {worst:.3e} > {max_tol:.1e} at floor {floor:g}"
    assert n_bad <= allowed, f"{what}: {n_bad} of {a.numel()} elements above {tol:.0e} at floor {floor:g} (allowed {allowed}; max {worst:.3e})"
    return worst, n_bad


def rel_err(a, b, floor=REL_FLOOR):
    a = torch.as_tensor(a).detach().double().reshape(-1).cpu()
    b = torch.as_tensor(b).detach().double().reshape(-1).cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return (a - b).abs() / b.abs().clamp(min=floor)


def assert_close_vs_f64(hip, ref32, f64, what="", tol=1e-4, floor=REL_FLOOR, factor=2.0, slack=2):
    """The 1e-4 bar at SURVEY 8(d)'s floor with the allowance DERIVED from the reference itself: `f64` is the imported
    reference run in double on the same inputs (tests/golden/round4.npz, oracle/make_golden.py:gen_round4), `ref32` the same
    reference in its own fp32.  Wherever the reference's fp32 value is itself further than `tol` from the double value (an SDF
    sample next to its zero crossing carries one or two fp32 ulps of the O(1) pre-activations that cancel to it), no fp32
    implementation can be held to `tol` against it; so the HIP result is held to the double value instead:
        * its worst element within `factor` x the reference's own worst (or `tol`, whichever is larger), and
        * at most `factor` x as many elements above `tol` as the reference's own fp32 result has (+ `slack`).
    -> (worst_hip, n_hip, worst_ref, n_ref)."""
    e_hip, e_ref = rel_err(hip, f64, floor), rel_err(ref32, f64, floor)
    worst_hip, worst_ref = float(e_hip.max()), float(e_ref.max())
    n_hip, n_ref = int((e_hip > tol).sum()), int((e_ref > tol).sum())
    assert worst_hip <= max(factor * worst_ref, tol), \
        f"{what}: max rel err vs float64 {worst_hip:.3e} > {factor:g} x the reference's own fp32 error {worst_ref:.3e} (floor {floor:g})"
    assert n_hip <= int(factor * n_ref) + slack, \
        f"{what}: {n_hip} elements above {tol:.0e} vs float64; the reference's own fp32 result has {n_ref} (floor {floor:g})"
    return worst_hip, n_hip, worst_ref, n_ref


def probe_points(n, stream=300, scale=1.15):
    return synth.hash_tensor((n, 3), stream, scale)


def grad_digest_check(named_grads, golden, prefix, tol, skip=()):
    """Compare per-tensor grad norm / sum / 64 strided samples with the golden digest."""
    checked = 0
    for k, g in named_grads.items():
        if any(s in k for s in skip):
            continue
        key = prefix + "|grad|" + k
        if key + "|norm" not in golden:
            continue
        g = g.detach().reshape(-1).double().cpu()
        gn = float(golden[key + "|norm"])
        scale = max(gn, 1e-12)
        assert abs(float(g.norm()) - gn) <= tol * max(gn, 1e-8) + 1e-12, f"{k} norm {float(g.norm())} vs {gn}"
        idx = torch.linspace(0, g.numel() - 1, min(64, g.numel())).long()
        smp = torch.from_numpy(golden[key + "|samples"]).double()
        # samples are compared relative to the tensor's RMS-per-element scale (atomics reorder sums)
        rms = scale / np.sqrt(g.numel())
        err = float((g[idx] - smp).abs().max())
        assert err <= tol * max(float(smp.abs().max()), rms) * 4 + 1e-12, f"{k} samples err {err}"
        checked += 1
    return checked


class DrawInjector:
    """Closed-form stand-ins for torch.rand / rand_like / randn_like (same class as oracle/make_golden.py): the k-th draw
    of shape `shape` is synth.hash_tensor(shape, 8000 + k), so the HIP path sees exactly the random perturbations the
    reference saw when the fixture was generated -- provided it draws in the same order with the same shapes."""

    def __init__(self, base=8000, remap=None):
        """remap: {draw number k: bool mask [n]} -- where the reference made draw k on a boolean-indexed SUBSET of n rows (shape
        [n_kept, ...]: get_normal_smoothness_loss drops the points outside the 1.1 sphere before it draws their angles,
        morpheus.py:543-549) and the HIP path keeps all n rows (the dropped ones leave through a zero weight), the reference's
        values go to the kept rows in their order and the others get 0."""
        self.k, self.base, self.remap = 0, base, dict(remap or {})

    def _next(self, shape, normal, device=None):
        self.k += 1
        mask = self.remap.get(self.k)
        if mask is not None:
            mask = torch.as_tensor(mask).reshape(-1).bool().cpu()
            assert int(shape[0]) == mask.numel(), (tuple(shape), mask.numel())
            sub = synth.hash_tensor((int(mask.sum()),) + tuple(shape[1:]), self.base + self.k, 0.5)
            v = torch.zeros(tuple(shape), dtype=sub.dtype)
            v[mask] = sub if normal else sub + 0.5
            v = v * 3.4 if normal else v
        else:
            v = synth.hash_tensor(tuple(shape), self.base + self.k, 0.5)
            v = v * 3.4 if normal else v + 0.5
        return v.to(device) if device is not None else v

    def __enter__(self):
        self._saved = (torch.rand, torch.rand_like, torch.randn_like)
        inj = self

        def rand(*size, device=None, dtype=None, **kw):
            size = size[0] if len(size) == 1 and isinstance(size[0], (list, tuple, torch.Size)) else size
            return inj._next(size, False, device)

        torch.rand = rand
        torch.rand_like = lambda t, **kw: inj._next(t.shape, False, t.device)
        torch.randn_like = lambda t, **kw: inj._next(t.shape, True, t.device)
        return self

    def __exit__(self, *a):
        torch.rand, torch.rand_like, torch.randn_like = self._saved
