"""GPU suite: PerturbationsModel (global_attacks/perturb_model.py:148-757) -- FGSM, I-FGSM (graph and
eager), targeted I-FGSM and MI-FGSM -- against the reference's own outputs on FlowNetC."""
from argparse import Namespace

import pytest
import torch

from conftest import load_golden, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("tag,method,targeted", [("fgsm", "fgsm", False), ("ifgsm", "ifgsm", False),
                                                 ("ifgsm_targeted", "ifgsm", True), ("mifgsm", "mifgsm", False)])
def test_perturbations_model_matches_reference(tag, method, targeted):
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    from understanding_flow_robustness_amd.perturb_model import PerturbationsModel
    z = load_golden("perturb_model_flownetc")
    args = Namespace(flownet="FlowNetC", flow_loss="l2")
    net = fetch_model(args, synthetic_seed=0).to(DEV)
    model = PerturbationsModel(perturb_method=method, perturb_mode="both", output_norm=0.02, n_step=3,
                               learning_rate=0.008, momentum=0.47, probability_diverse_input=0.0, disparity=False,
                               targeted=targeted, args=args)
    n0, n1, a0, a1 = model.forward(net, t(z["img0"], DEV), t(z["img1"], DEV), t(z["gt"], DEV))
    for got, key in ((n0, f"{tag}_noise0"), (n1, f"{tag}_noise1"), (a0, f"{tag}_adv0")):
        diff = (got.cpu() - t(z[key])).abs()
        # the outputs are step multiples of sign(gradient): a gradient within rounding of zero may flip
        # on another platform; everything else must agree to float rounding
        frac = float((diff > 1e-6).float().mean())
        assert frac < 5e-3, f"{key}: {frac:.2e} of the pixels differ"
        assert float(diff.max()) <= 2 * 0.02 + 1e-6


def test_adversarial_training_batch_composes_the_pinned_pieces():
    """training/train.py:171-222: attack in eval mode, EPE of the attacked prediction, clean + adversarial
    batch; parameters trainable and the module back in train mode afterwards; the captured step is reused."""
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model, predict_flow
    from understanding_flow_robustness_amd.losses import compute_epe
    from understanding_flow_robustness_amd.perturb_model import PerturbationsModel, adversarial_training_batch
    args = Namespace(flownet="FlowNetC", flow_loss="l2", perturb_method="ifgsm", perturb_mode="both", output_norm=0.02,
                     perturb_n_step=3, perturb_learning_rate=0.005, perturb_momentum=0.47, probability_diverse_input=0.0)
    net = fetch_model(args, synthetic_seed=0).to(DEV)
    for p in net.parameters():
        p.requires_grad_(True)
    net.train()
    g = torch.Generator().manual_seed(3)
    i1, i2 = torch.rand(2, 3, 64, 128, generator=g).to(DEV), torch.rand(2, 3, 64, 128, generator=g).to(DEV)
    flow = (3 * torch.randn(2, 2, 64, 128, generator=g)).to(DEV)
    valid = (torch.rand(2, 64, 128, generator=g) > 0.2).float().to(DEV)
    gt_full = torch.cat((flow, valid[:, None]), 1)
    for _ in range(2):
        b1, b2, bf, bv, epe = adversarial_training_batch(net, i1, i2, flow, valid, gt_full, args)
    assert b1.shape == (4, 3, 64, 128) and bf.shape == (4, 2, 64, 128) and bv.shape == (4, 64, 128)
    assert torch.equal(b1[:2], i1) and torch.equal(bf[2:], flow)
    assert net.training and all(p.requires_grad for p in net.parameters())
    assert len(net.__dict__["_ufr_perturb_steps"]) == 1
    net.eval()
    pm = PerturbationsModel("ifgsm", "both", 0.02, 3, 0.005, 0.47, 0.0, args=args)
    _, _, a1, a2 = pm.forward(net, i1, i2, gt_full)
    assert float((a1 - b1[2:]).abs().max()) <= 1e-6 and float((a2 - b2[2:]).abs().max()) <= 1e-6
    assert float((b1[2:] - i1).abs().max()) <= 0.02 + 1e-6                      # inside the epsilon ball
    with torch.no_grad():
        want = compute_epe(gt=gt_full, pred=predict_flow(net, None, a1, a2, args))
    assert abs(float(epe) - float(want)) <= 1e-4 * abs(float(want))
    loss = net(b1, b2)
    (loss[0] if isinstance(loss, tuple) else loss).mean().backward()             # the fine-tuning step still differentiates
    assert net.conv1[0].weight.grad is not None


@pytest.mark.parametrize("mode", ["both", "left", "right"])
def test_uniform_noise_follows_the_reference_np_random_stream(mode):
    """perturb_model.py:332-382: two host draws whatever the mode; bit-exact under the same seed."""
    import numpy as np
    from understanding_flow_robustness_amd.perturb_model import PerturbationsModel
    z = load_golden("perturb_noise_uniform")
    pm = PerturbationsModel(perturb_method="uniform", perturb_mode=mode, output_norm=0.05,
                            args=Namespace(flownet="FlowNetC", flow_loss="l2"))
    np.random.seed(int(z["seed"]))
    n0, n1, a0, a1 = pm.forward(None, t(z["img0"], DEV), t(z["img1"], DEV), None)
    assert np.float32(np.random.uniform()) == np.float32(z[f"{mode}_next_draw"]), "np.random consumed differently"
    for got, key in ((n0, "noise0"), (n1, "noise1"), (a0, "adv0"), (a1, "adv1")):
        assert np.array_equal(got.cpu().numpy(), z[f"{mode}_{key}"]), key


def test_gaussian_noise_statistics_and_clipping():
    """perturb_model.py:274-330 (skimage random_noise restated): sigma = eps/4, result clipped to [0,1]."""
    import numpy as np
    from understanding_flow_robustness_amd.perturb_model import PerturbationsModel
    pm = PerturbationsModel(perturb_method="gaussian", perturb_mode="left", output_norm=0.08,
                            args=Namespace(flownet="FlowNetC", flow_loss="l2"))
    img = torch.full((1, 3, 64, 96), 0.5, device=DEV)
    np.random.seed(3)
    n0, n1, a0, a1 = pm.forward(None, img, img.clone(), None)
    assert float(n1.abs().max()) == 0.0 and torch.equal(a1, img)
    assert abs(float(n0.std()) - 0.02) < 1e-3 and abs(float(n0.mean())) < 1e-3
    edge = torch.zeros(1, 3, 64, 96, device=DEV)
    _, _, a0, _ = pm.forward(None, edge, edge.clone(), None)
    assert float(a0.min()) == 0.0 and float(a0.max()) > 0.0
