"""Host logic of the flat optimiser (no GPU): bucket layout, torch.optim.Adam-format state, EMA rule, loud failure."""
import pytest
import torch

from morpheus_amd._lib import MorpheusHipError
from morpheus_amd.dist import GradBucket
from morpheus_amd.optim import FlatAdam, FlatEMA


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g))
    return mk(3, 5), mk(7), mk(2, 2)


def _groups(a, b, c):
    return [{"name": "decoder_sdf", "params": [a, b], "lr": 1e-2}, {"name": "pose", "params": [c], "lr": 1e-3}]


def test_layout_views_and_groups():
    a, b, c = _params()
    a0, c0 = a.detach().clone(), c.detach().clone()
    opt = FlatAdam(_groups(a, b, c), betas=(0.9, 0.99), eps=1e-15)
    assert opt._seg_end == [24, 28] and opt.n == 28            # groups start on 4-element boundaries
    assert torch.equal(a, a0) and torch.equal(c, c0)           # values preserved
    assert a.data_ptr() == opt.flat_p.data_ptr() and c.data_ptr() == opt.flat_p[24:].data_ptr()
    assert a.grad.data_ptr() == opt.bucket.flat.data_ptr() and c.grad.shape == c.shape
    assert [g["name"] for g in opt.param_groups] == ["decoder_sdf", "pose"]     # the keys update_learning_rate uses
    opt.zero_grad(set_to_none=True)                             # detaches p.grad: backward hands over fresh tensors
    assert a.grad is None
    (a.sum() * 2 + c.sum()).backward()
    assert a.grad.data_ptr() != opt.bucket.flat.data_ptr() and b.grad is None
    opt.bucket.allreduce_mean()                                 # single process: collect only, no collective
    assert a.grad.data_ptr() == opt.bucket.flat.data_ptr() and b.grad.data_ptr() == opt.bucket.flat[15:].data_ptr()
    assert float(opt.bucket.flat[:15].min()) == 2.0 and float(opt.bucket.flat[24:28].max()) == 1.0
    assert float(opt.bucket.flat[15:22].abs().max()) == 0.0     # b had no gradient: the zeros of zero()
    (a.sum() * 3).backward()                                    # a second backward accumulates into the bound view
    assert float(opt.bucket.flat[:15].max()) == 5.0
    opt.zero_grad()
    assert float(opt.bucket.flat.abs().max()) == 0.0


def test_state_dict_interchanges_with_torch_adam():
    a, b, c = _params()
    ra, rb, rc = (torch.nn.Parameter(t.detach().clone()) for t in (a, b, c))
    ref = torch.optim.Adam(_groups(ra, rb, rc), betas=(0.9, 0.99), eps=1e-15)
    for _ in range(3):
        for p in (ra, rb, rc):
            p.grad = torch.randn_like(p)
        ref.step()
    opt = FlatAdam(_groups(a, b, c), betas=(0.9, 0.99), eps=1e-15)
    opt.load_state_dict(ref.state_dict())
    assert opt._steps == [3, 3, 3]
    assert torch.equal(opt.exp_avg[:15].view(3, 5), ref.state[ra]["exp_avg"])
    assert torch.equal(opt.exp_avg_sq[24:28].view(2, 2), ref.state[rc]["exp_avg_sq"])
    sd = opt.state_dict()
    assert float(sd["state"][1]["step"]) == 3.0 and sd["param_groups"][1]["name"] == "pose"
    back = torch.optim.Adam(_groups(*(torch.nn.Parameter(t.detach().clone()) for t in (a, b, c))), betas=(0.9, 0.99),
                            eps=1e-15)
    back.load_state_dict(sd)                                    # and torch's Adam accepts ours
    assert torch.equal(back.state[back.param_groups[0]["params"][1]]["exp_avg"], ref.state[rb]["exp_avg"])


def test_no_cpu_path_and_unsupported_configs():
    a, b, c = _params()
    opt = FlatAdam(_groups(a, b, c))
    with pytest.raises(MorpheusHipError):
        opt.step()
    with pytest.raises(NotImplementedError):
        FlatAdam([{"params": [torch.nn.Parameter(torch.zeros(2))], "betas": (0.5, 0.9)},
                  {"params": [torch.nn.Parameter(torch.zeros(2))]}])
    with pytest.raises(NotImplementedError):
        FlatAdam([torch.nn.Parameter(torch.zeros(2, dtype=torch.float64))])


def test_ema_rule_and_store_restore():
    a, b, c = _params()
    opt = FlatAdam(_groups(a, b, c))
    ema = FlatEMA(opt, decay=0.95)
    shadow0 = a.detach().clone()
    with torch.no_grad():
        a.add_(1.0)
    ema.update()                                                # n=1: decay = min(0.95, 2/11)
    d = min(0.95, 2.0 / 11.0)
    want = shadow0 - (1.0 - d) * (shadow0 - a.detach())
    assert torch.allclose(ema.shadow[:15].view(3, 5), want, atol=1e-7)
    live = a.detach().clone()
    ema.store(); ema.copy_to()
    assert torch.allclose(a.detach(), want, atol=1e-7)
    ema.restore()
    assert torch.equal(a.detach(), live)
    sd = ema.state_dict()
    assert sd["num_updates"] == 1 and len(sd["shadow_params"]) == 3 and sd["shadow_params"][2].shape == (2, 2)
    ema2 = FlatEMA(opt, decay=0.5)
    ema2.load_state_dict(sd)
    assert ema2.decay == 0.95 and torch.equal(ema2.shadow, ema.shadow)


def test_grad_bucket_custom_layout():
    a, b, c = _params()
    bk = GradBucket.from_layout([(a, 4, 15), (c, 20, 4)], 24, a.device)
    assert a.grad.data_ptr() == bk.flat[4:].data_ptr() and bk.nbytes == 96 and b.grad is None


def test_per_parameter_steps_survive_a_freeze_lr_checkpoint():
    """A reference run with freeze_lr steps the pose group only on real-view iterations (morpheus.py:1399-1424): its
    torch.optim.Adam checkpoint holds DIFFERENT step counts per parameter.  FlatAdam keeps them per parameter."""
    a, b, c = _params()
    ra, rb, rc = (torch.nn.Parameter(t.detach().clone()) for t in (a, b, c))
    ref = torch.optim.Adam(_groups(ra, rb, rc), betas=(0.9, 0.99), eps=1e-15)
    for it in range(4):
        ra.grad, rb.grad = torch.randn_like(ra), torch.randn_like(rb)
        rc.grad = torch.randn_like(rc) if it % 2 == 0 else None          # the 'pose' group has no gradient every other step
        ref.step()
    assert float(ref.state[ra]["step"]) == 4 and float(ref.state[rc]["step"]) == 2
    opt = FlatAdam(_groups(a, b, c), betas=(0.9, 0.99), eps=1e-15)
    opt.load_state_dict(ref.state_dict())
    assert opt._steps == [4, 4, 2]
    sd = opt.state_dict()
    assert [float(sd["state"][i]["step"]) for i in range(3)] == [4.0, 4.0, 2.0]
    assert opt._kseg_end == [15, 22, 24, 28] and opt._kseg_param == [0, 1, -1, 2]   # parameter segments + the alignment pad


def test_ema_follows_model_parameter_order_and_validates():
    """torch_ema's shadow_params follow the order of the iterable it was built from (the reference: model.parameters(),
    registration order), not the optimiser's group order."""
    a, b, c = _params()
    opt = FlatAdam(_groups(a, b, c))
    ema = FlatEMA(opt, decay=0.9, parameters=[c, a, b])                    # a different order than the groups' (a, b, c)
    sd = ema.state_dict()
    assert [tuple(t.shape) for t in sd["shadow_params"]] == [(2, 2), (3, 5), (7,)]
    assert torch.equal(sd["shadow_params"][0], c.detach()) and torch.equal(sd["shadow_params"][1], a.detach())
    ema_wrong = FlatEMA(opt, decay=0.9)                                     # optimiser order: shapes do not line up
    with pytest.raises(ValueError):
        ema_wrong.load_state_dict(sd)
    ema_ok = FlatEMA(opt, decay=0.5, parameters=[c, a, b])
    ema_ok.load_state_dict(sd)
    assert torch.equal(ema_ok.shadow, ema.shadow)
    with pytest.raises(ValueError):
        FlatEMA(opt, decay=0.9, parameters=[a, b])
