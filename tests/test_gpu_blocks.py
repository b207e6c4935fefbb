"""Stand-alone DoubleConv / Down / Up / OutConv modules (networks/unet_parts.py of this build, running through the
operator-level C ABI) against the reference's block-level goldens G1: train-mode output, loss, input and parameter
gradients, running statistics after one and two calls, eval-mode output.  The goldens were produced by the
reference's own modules (tools/gen_goldens.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _build(name):
    from networks.unet_parts import DoubleConv, Down, OutConv, Up
    if name == "g1_doubleconv_3_8":
        return DoubleConv(3, 8)
    if name == "g1_doubleconv_8_8_mid4":
        return DoubleConv(8, 8, 4)
    if name.startswith("g1_down_8_16"):
        return Down(8, 16)
    if name.startswith("g1_up_16_8_convT"):
        return Up(16, 8, bilinear=False)
    if name == "g1_up_16_8_bilinear":
        return Up(16, 8, bilinear=True)
    if name == "g1_outconv_8_2":
        return OutConv(8, 2)
    raise KeyError(name)


def close(a, b, rtol=2e-4, atol=2e-5):
    np.testing.assert_allclose(a.detach().float().cpu().numpy(), np.asarray(b), rtol=rtol, atol=atol)


NAMES = ["g1_doubleconv_3_8", "g1_doubleconv_8_8_mid4", "g1_down_8_16", "g1_down_8_16_odd", "g1_up_16_8_convT",
         "g1_up_16_8_convT_odd", "g1_up_16_8_bilinear", "g1_outconv_8_2"]


@pytest.mark.parametrize("name", NAMES)
def test_block_module_matches_reference_golden(name):
    g = load_golden(name)
    m = _build(name)
    sd0 = {k[4:]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith("sd0.")}
    m.load_state_dict(sd0)
    m = m.cuda().train()
    ins = [torch.from_numpy(np.asarray(g[f"in{i}"])).cuda().requires_grad_(True) for i in range(2) if f"in{i}" in g.files]
    out = m(*ins)
    close(out, g["out_train"])
    loss = out.square().mean()
    loss.backward()
    close(loss, g["loss"], rtol=1e-4, atol=1e-7)
    for i, x in enumerate(ins):
        close(x.grad, g[f"gin{i}"], rtol=2e-3, atol=2e-6)
    for k, p in m.named_parameters():
        close(p.grad, g["g." + k], rtol=2e-3, atol=5e-6)
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            close(v, g["sd1." + k], rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        m(*[x.detach() for x in ins])
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            close(v, g["sd2." + k], rtol=1e-5, atol=1e-6)
    m.eval()
    with torch.no_grad():
        close(m(*[x.detach() for x in ins]), g["out_eval"])


def test_bilinear_upsample_is_the_adjoint_pair():
    """ustrun_upsample2x_fwd == F.interpolate(align_corners=True); _bwd == its autograd, odd and 1-pixel extents too."""
    import torch.nn.functional as TF
    from ustrun.blocks import _BilinearFn
    g = torch.Generator().manual_seed(4)
    for n, c, h, w in ((2, 8, 5, 7), (1, 4, 1, 3), (3, 12, 16, 16)):
        x = torch.randn(n, c, h, w, generator=g)
        xr = x.clone().requires_grad_(True)
        yr = TF.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        xg = x.cuda().requires_grad_(True)
        yg = _BilinearFn.apply(xg)                            # NHWC
        close(yg.permute(0, 3, 1, 2), yr.detach().numpy(), rtol=1e-4, atol=1e-5)
        yg.backward(dy.permute(0, 2, 3, 1).contiguous().cuda())
        close(xg.grad, xr.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_unet_bilinear_matches_oracle():
    """UNet(bilinear=True) (unet_model.py:17-22 channel plan) composed from the block modules vs the CPU oracle."""
    from networks.unet_model import UNet
    from oracle import unet_ref as U
    torch.manual_seed(11)
    sd = U.make_state_dict(3, 2, bilinear=True, base=8)
    x = torch.randn(2, 3, 32, 32)
    ref_sd = U.clone_sd(sd, requires_grad=True)
    ref = U.unet_forward(x, ref_sd, train=True, bilinear=True)
    ref.square().mean().backward()
    m = UNet(3, 2, bilinear=True, base_channels=8)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict({k: v.detach().clone() for k, v in sd.items()})
    m = m.cuda().train()
    out = m(x.cuda())
    close(out, ref.detach().numpy(), rtol=1e-3, atol=1e-5)
    out.square().mean().backward()
    for k, p in m.named_parameters():
        a, b = p.grad.double().cpu().flatten(), ref_sd[k].grad.double().flatten()
        assert float((a - b).norm() / (b.norm() + 1e-30)) < 2e-3, k
    with pytest.raises(RuntimeError, match="f32"):
        UNet(3, 2, bilinear=True, base_channels=8, dtype="bf16").cuda()(x.cuda())
