"""Optimizer state of the fused optimizers is saved in torch.optim's layout (`ckpt.pth` carried an empty
optimizer state, so the reference's [model, optimizer, step] checkpoints could not be resumed) and the data-parallel
gradient hooks of a finished optimizer are removed.  Host logic only — no kernel is launched."""
import torch
import torch.nn as nn

from unlearn_saliency_amd.flat import FlatArena
from unlearn_saliency_amd.optim import FusedMaskedAdam, FusedMaskedSGD


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4), nn.Flatten(), nn.Linear(4 * 6 * 6, 5))


def test_sgd_state_roundtrips_through_torch_optim_sgd():
    m = _model()
    arena = FlatArena.from_module(m)
    opt = FusedMaskedSGD(arena, 0.013, momentum=0.9, weight_decay=5e-4)
    opt.momentum_buffer.copy_(torch.randn(arena.n))
    opt.steps, opt._first_step = 3, False
    sd = opt.state_dict()
    assert len(sd["state"]) == len(arena.names)
    ref = torch.optim.SGD(m.parameters(), 0.5, momentum=0.1)
    ref.load_state_dict(sd)  # torch's own loader accepts the layout
    assert ref.param_groups[0]["lr"] == 0.013 and ref.param_groups[0]["momentum"] == 0.9
    for p, o, k in zip(m.parameters(), arena.offsets, arena.numels):
        assert torch.equal(ref.state[p]["momentum_buffer"].reshape(-1), opt.momentum_buffer[o:o + k])
    m2 = _model()
    a2 = FlatArena.from_module(m2)
    opt2 = FusedMaskedSGD(a2, 0.1, momentum=0.9)
    opt2.load_state_dict(ref.state_dict())  # and back from a genuine torch.optim.SGD state
    assert torch.equal(opt2.momentum_buffer, opt.momentum_buffer) and opt2.param_groups[0]["lr"] == 0.013
    opt2.load_state_dict(sd)
    assert opt2.steps == 3 and opt2._first_step is False


def test_adam_state_roundtrips_through_torch_optim_adam():
    m = _model()
    arena = FlatArena.from_module(m)
    opt = FusedMaskedAdam(arena, lr=1e-4)
    opt.exp_avg.copy_(torch.randn(arena.n))
    opt.exp_avg_sq.copy_(torch.rand(arena.n))
    opt.steps = 7
    sd = opt.state_dict()
    ref = torch.optim.Adam(m.parameters(), lr=1.0)
    ref.load_state_dict(sd)
    p0 = next(m.parameters())
    assert float(ref.state[p0]["step"]) == 7.0 and ref.param_groups[0]["lr"] == 1e-4
    assert torch.equal(ref.state[p0]["exp_avg_sq"].reshape(-1), opt.exp_avg_sq[:p0.numel()])
    opt2 = FusedMaskedAdam(FlatArena.from_module(_model()), lr=1.0)
    opt2.load_state_dict(ref.state_dict())
    assert opt2.steps == 7 and torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)


def test_unstarted_optimizer_has_empty_state_like_torch():
    opt = FusedMaskedSGD(FlatArena.from_module(_model()), 0.1, momentum=0.9)
    assert opt.state_dict()["state"] == {}


def test_close_removes_reducer_hooks():
    class FakeReducer:
        removed = 0

        def remove(self):
            FakeReducer.removed += 1

    opt = FusedMaskedSGD(FlatArena.from_module(_model()), 0.1)
    opt._reducer = FakeReducer()
    opt.close()
    opt.close()
    assert FakeReducer.removed == 1 and opt._reducer is None
