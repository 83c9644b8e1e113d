"""GPU suite: fused BatchNorm(+residual)(+ReLU) kernels (csrc/pointwise.hip) and the split-K Linear against torch."""
import pytest
import torch

from helpers import max_rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,c", [(5000, 32), (3001, 128), (777, 512), (40000, 64)])
@pytest.mark.parametrize("res", [False, True])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("train", [True, False])
def test_bn_act_matches_torch(n, c, res, relu, train):
    from pointcloudpdf_amd.dense import bn_act

    g = torch.Generator(device="cuda").manual_seed(n + c)
    x0 = torch.randn(n, c, device="cuda", generator=g) * 2 + 0.5
    r0 = torch.randn(n, c, device="cuda", generator=g) if res else None
    go = torch.randn(n, c, device="cuda", generator=g)
    outs = []
    for fused in (True, False):
        bn = torch.nn.BatchNorm1d(c).cuda()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, c)); bn.bias.copy_(torch.linspace(-0.3, 0.3, c))
            bn.running_mean.copy_(torch.linspace(-0.1, 0.1, c)); bn.running_var.copy_(torch.linspace(0.8, 1.2, c))
        bn.train(train)
        x = x0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if res else None
        if fused:
            y = bn_act(bn, x, r, relu)
        else:
            y = bn(x)
            if res:
                y = y + r
            if relu:
                y = torch.relu(y)
        y.backward(go)
        outs.append(dict(y=y.detach(), gx=x.grad, gr=r.grad if res else None, gw=bn.weight.grad, gb=bn.bias.grad,
                         rm=bn.running_mean.clone(), rv=bn.running_var.clone(), nb=bn.num_batches_tracked.clone()))
    a, b = outs
    for k in a:
        if a[k] is None:
            continue
        assert max_rel(a[k].float().cpu().numpy(), b[k].float().cpu().numpy()) < 2e-5, k


def test_split_k_linear_matches_torch():
    from pointcloudpdf_amd.dense import linear

    g = torch.Generator(device="cuda").manual_seed(1)
    x0 = torch.randn(50001, 32, device="cuda", generator=g)
    lin = torch.nn.Linear(32, 64).cuda()
    go = torch.randn(50001, 64, device="cuda", generator=g)
    res = []
    for f in (lambda x: linear(lin, x), lambda x: lin(x)):
        lin.zero_grad()
        x = x0.clone().requires_grad_(True)
        y = f(x)
        y.backward(go)
        res.append((y.detach(), x.grad, lin.weight.grad.clone(), lin.bias.grad.clone()))
    for a, b in zip(*res):
        assert max_rel(a.cpu().numpy(), b.cpu().numpy()) < 2e-5
