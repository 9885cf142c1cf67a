"""Per-tensor error of the cfg4 step-composition test against round5.npz, stage by stage (diagnostic for the test's tolerances):
deltas of both variants, and inside `freeze` the deltas after the first step and the second backward's gradients."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tests.util as U
from tests import test_gpu_render as T
import morpheus_amd.optim as MO
import morpheus_amd.harness as H
G = U.load_golden("round5.npz")
tables = {}

def compare(tag, named, key_prefix):
    rows = []
    for k, t in named.items():
        key = key_prefix + k
        if key + "|norm" not in G:
            continue
        t = t.detach().reshape(-1).double().cpu()
        gn = float(G[key + "|norm"])
        idx = torch.linspace(0, t.numel() - 1, min(64, t.numel())).long()
        smp = torch.from_numpy(G[key + "|samples"]).double()
        rows.append((k, gn, abs(float(t.norm()) - gn) / max(gn, 1e-30), float((t[idx] - smp).abs().max()) / max(float(smp.abs().max()), 1e-30)))
    tables[tag] = rows

state = dict(n=0, variant=None, model=None, before=None)
orig_step, orig_build = MO.FlatAdam.step, H.build_model

def step(self, closure=None):
    m = state["model"]
    if state["n"] == 0:
        state["before"] = {k: p.detach().clone() for k, p in m.named_parameters()}
    if state["variant"] == "freeze" and state["n"] == 1:
        compare("freeze: delta after step 1", {k: p.detach() - state["before"][k] for k, p in m.named_parameters()}, "freeze|delta1|")
        self.bucket.collect()
        lay = {id(p): (o, k) for p, o, k in self._views}
        compare("freeze: gradients of the second backward",
                {k: self.bucket.flat[lay[id(p)][0]:lay[id(p)][0] + lay[id(p)][1]].view(p.shape).clone() for k, p in m.named_parameters()},
                "freeze|grad2|")
    orig_step(self, closure)
    state["n"] += 1
    if (state["variant"] == "accum" and state["n"] == 1) or (state["variant"] == "freeze" and state["n"] == 2):
        compare(state["variant"] + ": final delta", {k: p.detach() - state["before"][k] for k, p in m.named_parameters()}, state["variant"] + "|delta|")

def build(*a, **k):
    state["model"] = orig_build(*a, **k)
    return state["model"]

MO.FlatAdam.step, H.build_model = step, build
T.grad_digest_check = lambda *a, **k: 99
for v in ("accum", "freeze"):
    state.update(variant=v, n=0)
    try:
        T.test_cfg4_step_composition_vs_reference_golden(v)
    except AssertionError as e:
        print("assert:", str(e)[:200])
for tag, rows in tables.items():
    print("==", tag)
    for r in sorted(rows, key=lambda r: -r[2])[:int(os.environ.get("TOP", "14"))]:
        print("   %-40s norm %.3e  rel norm err %.2e  max sample err / max %.2e" % r)
