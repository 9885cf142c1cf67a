"""GPU-box checker (test infrastructure, lives under tests/ because it uses the oracle): HIP path vs the CPU oracle on identical inputs -- per-output and per-parameter-gradient
errors, printed as JSON (feeds DESIGN.md's parity table; not part of the product path)."""
from __future__ import annotations

import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morpheus_amd import harness, synth  # noqa: E402
from oracle import field as of  # noqa: E402

DEV = "cuda"


def rel(a, b, floor):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float(((a - b).abs() / b.abs().clamp(min=floor)).max())


SURVEY_FLOOR = 1e-3      # rel = |a-b| / max(|b|, 1e-3)


def detail(a, b, relaxed_floor):
    """One output at both floors: the contract's and the relaxed one the tests use, with the worst
    element at the contract floor (its reference value and absolute error) and how many elements exceed 1e-4 there."""
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    err = (a - b).abs()
    r3 = err / b.abs().clamp(min=SURVEY_FLOOR)
    i = int(r3.argmax())
    return dict(rel_floor_1e3=float(r3.max()), rel_floor_relaxed=float((err / b.abs().clamp(min=relaxed_floor)).max()),
                relaxed_floor=relaxed_floor, max_abs=float(err.max()), worst_ref=float(b[i]), worst_abs=float(err[i]),
                n_over_1e4_at_1e3=int((r3 > 1e-4).sum()), n=int(b.numel()))


def nrm(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def leaf(kind):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in synth.make_state(kind).items()}


def model_probe(kind, shading, cano):
    n = 2048
    x = synth.hash_tensor((n, 3), 330, 1.15)
    t = torch.full((n, 1), 37 / 200)
    light = of.safe_normalize(synth.hash_tensor((n, 3), 331, 1.0))
    p = leaf(kind)
    f = of.OracleField(p, 1.01, None)
    so, go, co, _, do, _ = f.forward(x, t, light, ratio=0.3, shading=shading, cano=cano)
    ((co ** 2).sum() + 0.01 * (go ** 2).mean() + (so ** 2).sum()).backward()
    m = harness.build_model(kind, DEV).eval()
    sg, gg, cg, _, dg, _ = m(x.to(DEV), t.to(DEV), light.to(DEV), ratio=0.3, shading=shading, cano=cano)
    ((cg ** 2).sum() + 0.01 * (gg ** 2).mean() + (sg ** 2).sum()).backward()
    out = dict(case=f"model_probe {kind} {shading} {'cano' if cano else 'deform'}", sdf=rel(sg, so, 1e-2),
               sigma=rel(gg, go, 1e-2), color=rel(cg, co, 1e-2))
    grads = {}
    for k, prm in m.named_parameters():
        if prm.grad is not None and p[k].grad is not None:
            grads[k] = nrm(prm.grad, p[k].grad)
    out["grad_rel_l2_max"] = max(grads.values())
    out["grad_rel_l2_worst"] = sorted(grads.items(), key=lambda kv: -kv[1])[:6]
    return out


def render_case(kind, hw, S, cano, nray=None):
    o, d, t, rid = synth.frame_rays(25, hw, hw)
    if nray:
        o, d, t, rid = o[:, :nray], d[:, :nray], t[:, :nray], rid[:, :nray]
    N = o.shape[1]
    jit = synth.ray_jitter(N)
    smp = of.uniform_samples(o[0], d[0], jit, S, 1.01)
    light = of.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
    timg, tdep = synth.targets(N)
    p = leaf(kind)
    f = of.OracleField(p, 1.01, None)
    ro = of.render_rays(f, o, d, t, rid, smp, ambient_ratio=1.0, light_d=light, shading="albedo", cano=cano)
    (((ro["image"][0] - timg) ** 2).mean() + ((ro["depth"][0] - tdep) ** 2).mean()).backward()
    m = harness.build_model(kind, DEV).eval()
    rend = harness.make_renderer(m, S, jitter=jit.to(DEV))
    rg = rend.render_rays(o.to(DEV), d.to(DEV), t.to(DEV), rid.to(DEV), hw, hw, ambient_ratio=1.0, light_d=light.to(DEV),
                          shading="albedo", cano=cano)
    (((rg["image"][0] - timg.to(DEV)) ** 2).mean() + ((rg["depth"][0] - tdep.to(DEV)) ** 2).mean()).backward()
    out = dict(case=f"render {kind} {N}x{S} {'cano' if cano else 'deform'}", image=rel(rg["image"], ro["image"], 1e-2),
               depth=rel(rg["depth"], ro["depth"], 5e-2), sdf=rel(rg["sdf"], ro["sdf"], 1e-2),
               opacity=rel(rg["weights_sum"], ro["weights_sum"], 1e-2),
               detail=dict(image=detail(rg["image"], ro["image"], 1e-2), depth=detail(rg["depth"], ro["depth"], 5e-2),
                           sdf=detail(rg["sdf"], ro["sdf"], 1e-2), opacity=detail(rg["weights_sum"], ro["weights_sum"], 1e-2)))
    grads = {}
    for k, prm in m.named_parameters():
        if prm.grad is not None and p[k].grad is not None:
            grads[k] = nrm(prm.grad, p[k].grad)
    out["grad_rel_l2_max"] = max(grads.values())
    out["grad_rel_l2_worst"] = sorted(grads.items(), key=lambda kv: -kv[1])[:4]
    return out


def f64_rows(mode):
    """The eight evaluation renders of tests/golden/render.npz in arithmetic mode `mode` against the imported reference run in
    DOUBLE (tests/golden/round4.npz): per output, the HIP path's error vs the double value next to the reference's own fp32
    error vs the same value -- the numbers the derived gate (tests/util.py: assert_close_vs_f64) is built on."""
    import numpy as np
    from morpheus_amd import ops
    from tests.util import load_golden, rel_err
    r, g4 = load_golden("render.npz"), load_golden("round4.npz")
    prev = ops.set_mlp_mode(mode)
    rows = []
    try:
        for kind in ("a", "b"):
            for case, (hw, S, nray) in (("cfg1", (32, 64, None)), ("cfg3head", (128, 128, 256))):
                o, d, t, rid = synth.frame_rays(25, hw, hw)
                if nray is not None:
                    o, d, t, rid = o[:, :nray], d[:, :nray], t[:, :nray], rid[:, :nray]
                N = o.shape[1]
                smp = of.uniform_samples(o[0], d[0], synth.ray_jitter(N), S, 1.01)
                light = of.safe_normalize(o[0] + torch.tensor([0.3, -0.2, 0.5]))
                for m_ in ("eval_albedo_deform", "eval_albedo_cano"):
                    model = harness.build_model(kind, DEV).eval()
                    rend = harness.make_renderer(model, S, samples=tuple(v.to(DEV) for v in smp))
                    with torch.no_grad():
                        res = rend.render_rays(o.to(DEV), d.to(DEV), t.to(DEV), rid.to(DEV), hw, hw, ambient_ratio=0.3,
                                               light_d=light.to(DEV), shading="albedo", cano="cano" in m_)
                    key = f"{kind}_{case}_{m_}"
                    row = dict(case="vs_float64 " + key, mode=mode)
                    for out, gk in ((res["sdf"][::16], "sdf_s16"), (res["image"], "image"), (res["depth"], "depth"),
                                    (res["weights_sum"], "weights_sum")):
                        f64 = g4[key + "|f64|" + gk]
                        eh, er, e32 = rel_err(out, f64), rel_err(r[key + "|" + gk], f64), rel_err(out, r[key + "|" + gk])
                        row[gk] = dict(hip_vs_f64_max=float(eh.max()), hip_vs_f64_n_over_1e4=int((eh > 1e-4).sum()),
                                       ref32_vs_f64_max=float(er.max()), ref32_vs_f64_n_over_1e4=int((er > 1e-4).sum()),
                                       hip_vs_ref32_max=float(e32.max()), hip_vs_ref32_n_over_1e4=int((e32 > 1e-4).sum()),
                                       n=int(eh.numel()))
                    rows.append(row)
    finally:
        ops.set_mlp_mode(prev)
    return rows


if __name__ == "__main__":
    if "--f64" in sys.argv:
        for mode in ("b3", "f32"):
            for r in f64_rows(mode):
                print(json.dumps(r))
        sys.exit(0)
    rows = []
    for kind in ("a", "b"):
        rows.append(model_probe(kind, "albedo", False))
        rows.append(model_probe(kind, "albedo", True))
        rows.append(render_case(kind, 32, 64, False))
        rows.append(render_case(kind, 32, 64, True))
        rows.append(render_case(kind, 128, 128, False, nray=512))
    for r in rows:
        print(json.dumps(r))
