"""optim.HipAdam (csrc/optim.hip) against torch.optim.Adam: the optimiser of the reference's inversion loop
(/root/reference/exp/cips3d/models/projector_v10.py:279-390, 1210-1216)."""
import pytest
import torch

from cips_3dplusplus_amd.optim import HipAdam

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_hip_adam_tracks_torch_adam():
    """Twelve steps over tensors of awkward sizes (1, 3, 4097 elements, a 7 M-element one is not needed: the kernel is
    element-wise), two groups with their own learning rate / betas, a learning rate that is ramped and set to 0 for some steps
    (the pose phase freezes the decoder this way: the moments still move), a parameter without a gradient on some steps."""
    g = torch.Generator(device=DEV).manual_seed(0)
    shapes = [(1,), (3,), (1, 3, 1, 1), (512, 512), (4097,), (2, 18, 512), (33, 7)] + [(64, 64)] * 60     # > 48 tensors per group
    ref_p = [torch.randn(*s, device=DEV, generator=g).requires_grad_(True) for s in shapes]
    hip_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    mk = lambda ps: [{"params": ps[:5], "lr": 0.02, "betas": (0.9, 0.999)}, {"params": ps[5:], "lr": 0.005, "betas": (0.8, 0.99)}]  # noqa: E731
    o_ref, o_hip = torch.optim.Adam(mk(ref_p)), HipAdam(mk(hip_p))
    for step in range(12):
        for grp_r, grp_h, lr0 in zip(o_ref.param_groups, o_hip.param_groups, (0.02, 0.005)):
            lr = 0.0 if (step in (4, 5) and lr0 == 0.005) else lr0 * (0.5 + 0.05 * step)
            grp_r["lr"] = grp_h["lr"] = lr
        for i, (a, b) in enumerate(zip(ref_p, hip_p)):
            if i == 2 and step % 3 == 0:                   # no gradient this step: torch skips the parameter, so does HipAdam
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, device=DEV, generator=g) * (1e-4 if i % 2 else 1.0)
            a.grad, b.grad = gr.clone(), gr.clone()
        o_ref.step(); o_hip.step()
        for i, (a, b) in enumerate(zip(ref_p, hip_p)):
            assert float((a - b).detach().abs().max()) <= 2e-6 * max(1.0, float(a.detach().abs().max())), (step, i)
    st_r, st_h = o_ref.state[ref_p[3]], o_hip.state[hip_p[3]]
    assert int(st_r["step"]) == st_h["step"] == 12
    assert float((st_r["exp_avg"] - st_h["exp_avg"]).abs().max()) < 1e-6
    assert float((st_r["exp_avg_sq"] - st_h["exp_avg_sq"]).abs().max()) < 1e-6
    assert o_hip.state[hip_p[2]]["step"] == 8              # skipped on steps 0, 3, 6, 9
    opt = HipAdam([torch.zeros(3, requires_grad=True)])    # construction is fine ...
    opt.param_groups[0]["params"][0].grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="fp32 CUDA"):
        opt.step()                                          # ... the step refuses CPU tensors (no CPU fallback)


@pytest.mark.parametrize("n0,n1", [(2 * 3 * 256 * 256, 2 * 3 * 64 * 64), (1001, 7), (4099, 0)])
def test_fused_squared_difference_pair_matches_torch(n0, n1):
    """autograd.SqDiffPairFn (cips3d_sqdiff_pair / _bwd): the inversion loss' two squared-difference terms
    (projector_v10.py:1173-1178) against the torch expression -- value, both gradients, bit-identical repeats."""
    from cips_3dplusplus_amd import autograd as AG
    g = torch.Generator().manual_seed(n0 + n1)
    a0, b0 = torch.randn(n0, generator=g).cuda(), torch.randn(n0, generator=g).cuda()
    a1 = torch.randn(n1, generator=g).cuda() if n1 else None
    b1 = torch.randn(n1, generator=g).cuda() if n1 else None
    c0, c1 = 1.0 / n0, (50.0 / n1 if n1 else 0.0)
    x0 = a0.clone().requires_grad_(True)
    x1 = a1.clone().requires_grad_(True) if n1 else None
    ref = c0 * ((x0.double() - b0.double()) ** 2).sum()
    if n1:
        ref = ref + c1 * ((x1.double() - b1.double()) ** 2).sum()
    (3.0 * ref).backward()
    y0 = a0.clone().requires_grad_(True)
    y1 = a1.clone().requires_grad_(True) if n1 else None
    out = AG.SqDiffPairFn.apply(y0, b0, c0, y1, b1, c1)
    assert out.shape == () and abs(float(out) - float(ref)) < 2e-6 * abs(float(ref))
    (3.0 * out).backward()
    assert float((y0.grad - x0.grad).abs().max()) <= 1e-6 * float(x0.grad.abs().max())
    if n1:
        assert float((y1.grad - x1.grad).abs().max()) <= 1e-6 * float(x1.grad.abs().max())
    again = AG.SqDiffPairFn.apply(a0, b0, c0, a1, b1, c1)
    assert torch.equal(again, out.detach())


def test_surrogate_loss_takes_the_fused_node_on_the_gpu():
    from cips_3dplusplus_amd.projector import surrogate_loss
    g = torch.Generator().manual_seed(5)
    t_rgb, t_th = torch.randn(2, 3, 32, 32, generator=g).cuda(), torch.randn(2, 3, 8, 8, generator=g).cuda()
    rgb = torch.randn(2, 3, 32, 32, generator=g).cuda().requires_grad_(True)
    th = torch.randn(2, 3, 8, 8, generator=g).cuda().requires_grad_(True)
    loss = surrogate_loss(t_rgb, t_th)(rgb, th)
    assert "SqDiffPairFn" in type(loss.grad_fn).__name__
    ref = ((rgb.detach() - t_rgb) ** 2).mean() + 50.0 * ((th.detach() - t_th) ** 2).mean()
    assert abs(float(loss) - float(ref)) < 1e-5 * float(ref)
    loss.backward()
    assert float((rgb.grad - 2 * (rgb.detach() - t_rgb) / rgb.numel()).abs().max()) < 1e-9
    # CPU tensors (the oracle's loops) keep the torch expression
    l2 = surrogate_loss(t_rgb.cpu(), t_th.cpu())(rgb.detach().cpu().requires_grad_(True), th.detach().cpu())
    assert "SqDiffPairFn" not in type(l2.grad_fn).__name__
