"""GPU parity of the validation path (ustrun.evaluate.validate) against the oracle's restatement of the reference's
`test()` (train.py:253-395) on the same seeded loaders and weights."""
import numpy as np
import pytest
import torch

from oracle import eval_ref as E
from oracle import unet_ref as U

pytestmark = pytest.mark.gpu


def _weights(c, k, base, seed):
    torch.manual_seed(seed)
    sd = U.make_state_dict(c, k, base=base)
    g = torch.Generator().manual_seed(seed + 1)
    for key in sd:                  # running statistics and affine away from their defaults: eval must use them
        if key.endswith("running_mean"):
            sd[key] = 0.1 * torch.randn(sd[key].shape, generator=g)
        if key.endswith("running_var"):
            sd[key] = 0.5 + torch.rand(sd[key].shape, generator=g)
        if key.endswith(("1.weight", "4.weight")) and sd[key].dim() == 1:
            sd[key] = 1 + 0.3 * torch.randn(sd[key].shape, generator=g)
    return sd


@pytest.mark.parametrize("dataset,c,k", [("fundus", 3, 2), ("prostate", 1, 2), ("MNMS", 1, 4)])
def test_validate_matches_oracle(dataset, c, k):
    from networks.unet_model import UNet
    from ustrun import synthetic
    from ustrun.evaluate import validate
    H, base = 48, 8
    sd = _weights(c, k, base, seed=21)
    loaders = synthetic.test_loaders(dataset, 2, 3, 2, c, H, seed=5)
    want, want_dom = E.validate(dataset, sd, loaders)
    model = UNet(n_channels=c, n_classes=k, base_channels=base)
    model.load_state_dict({kk: v.detach().clone() for kk, v in sd.items()})
    model = model.cuda()
    got, got_dom = validate(dataset, model, loaders, epoch=3, log=None)
    one, one_dom = validate(dataset, model, loaders, epoch=3, log=None, coalesce=1)     # a forward per loader batch
    assert one == got and one_dom == got_dom                # coalescing batches does not change a bit
    assert model.training                                   # left in train mode, as the reference does
    # predictions are thresholded logits: a pixel within rounding of the boundary may flip between summation orders
    np.testing.assert_allclose(got, want, atol=2e-3)
    np.testing.assert_allclose(got_dom, want_dom, atol=4e-3)
    # the running statistics did not move (eval mode) and num_batches_tracked is untouched
    for kk, v in model.state_dict().items():
        if "running" in kk or "num_batches" in kk:
            assert torch.equal(v.cpu(), sd[kk]), kk


def test_predict_and_batch_dice_are_the_reference_rules():
    """first-index arg-max on ties, sigmoid(x) >= .5 per channel, Dice from device counts == utils.metrics on the host"""
    from ustrun.evaluate import batch_dice, predict
    from utils import metrics
    g = torch.Generator().manual_seed(3)
    lg = torch.randn(3, 4, 16, 16, generator=g)
    lg[:, 1] = lg[:, 2]                                      # ties between classes 1 and 2
    pred = predict("MNMS", lg.cuda())
    assert torch.equal(pred.cpu(), E.predict("MNMS", lg))
    tgt = torch.randint(0, 4, (3, 16, 16), generator=g)
    np.testing.assert_allclose(batch_dice("MNMS", pred, tgt.cuda()), metrics.dice_coeff_3label(pred.cpu().numpy(), tgt), rtol=1e-12)
    lf = torch.randn(3, 2, 16, 16, generator=g)
    lf[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1e-9, -1e-9])
    pf = predict("fundus", lf.cuda())
    assert torch.equal(pf.cpu().bool(), E.predict("fundus", lf))
    tf = (torch.rand(3, 2, 16, 16, generator=g) > 0.5).float()
    np.testing.assert_allclose(batch_dice("fundus", pf, tf.cuda()), metrics.dice_coeff_2label(pf.cpu().numpy(), tf), rtol=1e-12)
