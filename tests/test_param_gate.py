"""Host logic of ``tvqaplus_amd.groups.gate``: the parameters of a module that is applied to several streams (model/stage.py:226-269)
receive ONE gradient per step -- the sum of what every group call put into the gate's sink plus whatever reached the alias through
ordinary autograd.  Runs on CPU (no kernel involved)."""
import torch

from tvqaplus_amd import groups


class _Use(torch.autograd.Function):
    """Stand-in for a group call: its parameter gradients are constants k, delivered the way the real groups deliver them."""

    @staticmethod
    def forward(ctx, x, k, *params):
        ctx.sinks = groups._sinks(params)
        ctx.k = k
        ctx.shapes = [p.shape for p in params]
        return x * 1.0

    @staticmethod
    def backward(ctx, g):
        grads = [torch.full(s, ctx.k) for s in ctx.shapes]
        return (g, None) + groups._deliver(ctx.sinks, grads)


def test_gate_hands_every_parameter_the_sum_of_its_uses():
    w = torch.randn(4, requires_grad=True)
    b = torch.randn(3, requires_grad=True)
    x = torch.ones(2, requires_grad=True)
    gw, gb = groups.gate([w, b])
    y = _Use.apply(x, 1.0, gw, gb) + _Use.apply(x, 2.0, gw, gb) + _Use.apply(x, 4.0, gw)   # the third use touches w only
    (y.sum() + (gw * 3.0).sum()).backward()                                                # + an ordinary use of the alias
    assert torch.equal(w.grad, torch.full((4,), 1.0 + 2.0 + 4.0 + 3.0))
    assert torch.equal(b.grad, torch.full((3,), 3.0))
    assert torch.equal(x.grad, torch.full((2,), 3.0))


def test_ungated_parameters_take_the_ordinary_path():
    w = torch.randn(4, requires_grad=True)
    x = torch.ones(2, requires_grad=True)
    (_Use.apply(x, 1.0, w) + _Use.apply(x, 2.0, w)).sum().backward()
    assert torch.equal(w.grad, torch.full((4,), 3.0))


def test_unused_gate_leaves_no_gradient():
    w = torch.randn(4, requires_grad=True)
    x = torch.ones(2, requires_grad=True)
    groups.gate([w])
    (x * 2).sum().backward()
    assert w.grad is None
