"""GPU tests: the training step captured in a HIP graph (kws_amd.GraphedStep) reproduces the eager step bit for bit --
the headline cell, the low-rank cell and the reference's default two-layer model with its fused head -- and follows new
data copied into the captured input tensors."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from kws_amd import FastGRNNCUDA, GraphedStep, RNNClassifierModel
DEV = "cuda:0"


@pytest.mark.parametrize("F,H,rank,B", [(32, 128, None, 48), (32, 256, 16, 37), (32, 256, None, 32)])
def test_captured_cell_step_equals_eager_and_follows_new_batches(F, H, rank, B):
    T = 25
    torch.manual_seed(3)
    m = FastGRNNCUDA(F, H, wRank=rank, uRank=rank, device=DEV)
    params = list(m.parameters())
    x = torch.randn(T, B, F, device=DEV)
    G = torch.randn(T, B, H, device=DEV)

    def step():
        for p in params:
            p.grad = None
        hs = m(x)
        hs.backward(G)
        return hs

    def eager(xv, Gv):
        x.copy_(xv); G.copy_(Gv)
        hs = step()
        torch.cuda.synchronize()
        return hs.detach().clone(), [p.grad.clone() for p in params]

    batches = [(torch.randn(T, B, F, device=DEV), torch.randn(T, B, H, device=DEV)) for _ in range(3)]
    want = [eager(*b) for b in batches]
    g = GraphedStep(step)
    for (xv, Gv), (hs_e, gr_e) in zip(batches, want):
        x.copy_(xv); G.copy_(Gv)
        hs = g()
        torch.cuda.synchronize()
        assert torch.equal(hs, hs_e)
        for p, ge in zip(params, gr_e):
            assert torch.equal(p.grad, ge)


def test_captured_two_layer_model_step_equals_eager():
    T, B, F, C = 31, 40, 32, 12
    torch.manual_seed(5)
    m = RNNClassifierModel("FastGRNNCUDA", F, 2, [256, 128], [None, None], [None, None], [1.0, 1.0], [1.0, 1.0],
                           "sigmoid", "tanh", num_classes=C, device=DEV)
    params = list(m.parameters())
    x = torch.randn(T, B, F, device=DEV)
    y = torch.randint(0, C, (B,), device=DEV)

    def step():
        for p in params:
            p.grad = None
        m.init_hidden()
        loss = m.loss(x, y)
        loss.backward()
        return loss

    loss_e = step().detach().clone()
    torch.cuda.synchronize()
    gr_e = [p.grad.clone() for p in params]
    g = GraphedStep(step)
    for _ in range(2):
        loss = g()
    torch.cuda.synchronize()
    assert torch.equal(loss.detach(), loss_e)
    for p, ge in zip(params, gr_e):
        assert torch.equal(p.grad, ge)
