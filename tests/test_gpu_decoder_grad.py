"""GPU tests of the one-call decoder backward's building blocks and of the call itself (BASELINE config 5;
the reference gets this backward from one `loss.backward()`, /root/reference/exp/cips3d/models/projector_v10.py:1203-1209).
Every kernel is compared with an fp64 evaluation of the same formula; bars are stated relative to what the fp32 evaluation of
that formula itself misses against fp64."""
import pytest
import torch

import cips_3dplusplus_amd as pkg
from cips_3dplusplus_amd import hip

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, ref):
    return float((a.double().cpu() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("B,M,K,P", [(2, 512, 512, 4096), (1, 64, 32, 65536), (2, 32, 32, 1024), (3, 96, 160, 2048)])
@pytest.mark.parametrize("sa,sb", [(1.0, 1.0), (2.0 ** -30, 2.0 ** 17), (2.0 ** 20, 2.0 ** -24)])
def test_split_wgrad_is_as_accurate_as_fp32_at_any_magnitude(B, M, K, P, sa, sb):
    """dwm[b] = dy[b] x[b]^T over the pixels on split-fp16 products: gradients of any magnitude (the scale comes from the
    measured maxima) within 1.5x of what the fp32-MFMA kernel misses against fp64, no inf / nan; accumulate mode adds."""
    g = torch.Generator(device=DEV).manual_seed(M + K + B)
    dy = torch.randn(B, M, P, device=DEV, generator=g) * sa
    x = torch.randn(B, K, P, device=DEV, generator=g) * sb
    x[:, :, ::7] *= 1e-3                                   # a wide spread inside one tensor
    ref = torch.einsum("bmp,bkp->bmk", dy.double().cpu(), x.double().cpu())
    f32 = hip.gemm_wgrad(dy, x)
    sp = hip.gemm_wgrad_split(dy, x, hip.absmax(dy), hip.absmax(x))
    assert bool(torch.isfinite(sp).all())
    e32, esp = _rel(f32, ref), _rel(sp, ref)
    assert esp <= 1.5 * e32 + 1e-7, (esp, e32)
    acc = sp.clone()
    hip.gemm_wgrad_split(dy, x, hip.absmax(dy), hip.absmax(x), out=acc)
    assert _rel(acc, 2 * ref) <= 1.5 * e32 + 1e-7
