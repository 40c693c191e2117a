"""CPU suite: host-side rules that need no GPU -- seeds of the fused dropout layer, the optimizer's folded l2 term, and when
the concat buffer's gradient may be accumulated into in place."""
import torch


def test_dropout_dense_default_seed_depends_on_the_torch_seed_only():
    """Two models built one after the other under the same torch.manual_seed draw the same mask streams (the default salt is
    drawn from torch's generator, not from a per-process construction counter); two layers of one model differ."""
    from h2gcn_amd.layers import DropoutDense

    def build():
        torch.manual_seed(5)
        return [DropoutDense(8, 4, True, 0.5).seed for _ in range(3)]

    first, second = build(), build()
    assert first == second and len(set(first)) == 3
    torch.manual_seed(6)
    assert DropoutDense(8, 4, True, 0.5).seed != first[0]
    assert DropoutDense(8, 4, True, 0.5, seed=77).seed == 77


def test_keras_adam_steps_an_l2_registered_kernel_that_got_no_data_gradient():
    """`KerasAdam.set_l2` folds 2*l2*w into the step.  A registered kernel without a data gradient (it took no part in the loss)
    still owes that term -- exactly what autograd would have produced had the penalty been part of the loss."""
    from h2gcn_amd.optim import KerasAdam

    torch.manual_seed(0)
    w_used, w_idle = torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5, 2))
    ref_used, ref_idle = (torch.nn.Parameter(p.detach().clone()) for p in (w_used, w_idle))
    l2 = 0.05
    folded = KerasAdam([w_used, w_idle], lr=0.01)
    folded.set_l2([w_used, w_idle], l2)
    in_graph = KerasAdam([ref_used, ref_idle], lr=0.01)
    x = torch.randn(6, 4)
    for _ in range(3):
        folded.zero_grad(set_to_none=True)
        (x @ w_used).square().sum().backward()              # w_idle: no gradient at all
        folded.step()
        in_graph.zero_grad(set_to_none=True)
        ((x @ ref_used).square().sum() + l2 * (ref_used ** 2).sum() + l2 * (ref_idle ** 2).sum()).backward()
        in_graph.step()
    assert torch.allclose(w_used, ref_used, atol=1e-6) and torch.allclose(w_idle, ref_idle, atol=1e-6)
    assert not torch.equal(w_idle.detach(), torch.nn.Parameter(w_idle.detach().clone()).detach() * 0 + ref_idle.detach() * 0 + w_idle.detach() * 0)  # (moved)


def test_buffer_gradient_is_private_only_behind_a_layer_that_allocates_it():
    """`H2GCN._buffer_grad_is_private`: the propagation's backward may accumulate into the gradient of its concat buffer in
    place only when that gradient is a fresh tensor -- behind DropoutDense / Dense, or an ACTIVE nn.Dropout.  nn.Dropout in eval
    mode or with rate 0 hands its input through without an autograd node: the rule looks through it like through nn.Identity."""
    from h2gcn_amd import layers as L
    from h2gcn_amd.models.H2GCN import H2GCN

    class Stub:
        tags = {}
        _buffer_grad_is_private = H2GCN._buffer_grad_is_private

    def private(*layers, tags=None):
        s = Stub()
        s.layer_objs, s.tags = [torch.nn.Identity()] + list(layers), (tags or {})
        return s._buffer_grad_is_private(1)

    dense = L.DropoutDense(8, 4, True, 0.5)
    active, off, zero = torch.nn.Dropout(0.5).train(), torch.nn.Dropout(0.5).eval(), torch.nn.Dropout(0.0).train()
    assert private(dense) and private(active) and private(torch.nn.Identity(), dense)
    assert private(off, dense) and private(zero, dense)            # looked through: the dense layer's fresh gradient
    assert not private(off) and not private(zero)                   # ... nothing behind it: the caller's own tensor
    assert not private(off, torch.nn.ReLU()) and not private()
    assert not private(off, dense, tags={1: "a"})                   # a tagged pass-through exposes the tensor
    assert not private(dense, tags={0: "a"})                        # the block's last layer is tagged
