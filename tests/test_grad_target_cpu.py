"""Host logic of ops._grad_target (no GPU, no kernel call): the in-place accumulation of a parameter's gradient across
the backward nodes of one pass keeps only a weak reference to the first node's tensor, so it can never write into a
tensor autograd has dropped (torch.autograd.grad(inputs=[activation]) and foreign contributions)."""
import torch

from de_i2i_gan_amd import ops


class _Mul(torch.autograd.Function):
    log = []

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.param = w
        return x * w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        if not ops._wants_grad(ctx, 1):
            _Mul.log.append("skipped")
            return g * w, None
        dw, ptr, acc = ops._grad_target(ctx.param, w.shape, w.device)
        _Mul.log.append("acc" if acc else "first")
        if acc:
            first = ops._grad_slots[ctx.param.data_ptr()][1]()
            assert first is not None and first.data_ptr() == ptr
            first.add_(g * x)
        else:
            dw.copy_(g * x)
        return g * w, dw


def test_second_node_accumulates_into_first_nodes_tensor():
    _Mul.log.clear()
    w = torch.nn.Parameter(torch.randn(5))
    x = torch.randn(5, requires_grad=True)
    _Mul.apply(_Mul.apply(x, w), w).sum().backward()
    assert _Mul.log == ["first", "acc"]
    assert torch.allclose(w.grad, (2 * x * w).detach())


def test_activation_only_gradient_never_touches_a_dropped_tensor():
    _Mul.log.clear()
    w = torch.nn.Parameter(torch.randn(5))
    x = torch.randn(5, requires_grad=True)
    (gx,) = torch.autograd.grad(_Mul.apply(_Mul.apply(x, w), w).sum(), [x])
    assert _Mul.log == ["skipped", "skipped"] and w.grad is None
    assert torch.allclose(gx, (w * w).detach())


def test_dead_first_tensor_gives_a_fresh_one():
    """Simulates the engine dropping the first node's gradient: the slot's weak reference is dead -> no accumulate."""
    w = torch.nn.Parameter(torch.randn(3))
    seen = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            t, _, acc = ops._grad_target(w, w.shape, w.device)
            seen.append(acc)
            del t                                    # nothing keeps the first tensor alive
            t2, _, acc2 = ops._grad_target(w, w.shape, w.device)
            seen.append(acc2)
            return g

    x = torch.randn(3, requires_grad=True)
    Probe.apply(x).sum().backward()
    assert seen == [0, 0]


def test_foreign_contribution_to_the_same_parameter():
    _Mul.log.clear()
    w = torch.nn.Parameter(torch.randn(5))
    x = torch.randn(5, requires_grad=True)
    (_Mul.apply(_Mul.apply(x, w), w).sum() + (w * 3).sum()).backward()
    assert torch.allclose(w.grad, (2 * x * w).detach() + 3)


def test_leaf_captured_by_autograd_grad_is_wanted():
    _Mul.log.clear()
    w = torch.nn.Parameter(torch.randn(5))
    x = torch.randn(5, requires_grad=True)
    (gw,) = torch.autograd.grad(_Mul.apply(_Mul.apply(x, w), w).sum(), [w])
    assert "skipped" not in _Mul.log
    assert torch.allclose(gw, (2 * x * w).detach())
