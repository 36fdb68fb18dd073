"""CPU suite, part 3: the model-level oracle (oracle/flow_oracle.py) reproduces the golden vectors
captured from the reference's own Python modules (FlowNetC forward/gradient, attack() traces,
EPE / cosine metrics); the host-side pieces of the product that need no GPU are checked too."""
import pytest
import torch

from conftest import assert_close, load_golden, t

ATTACK_CASES = [("cos_lr1000", False, 1000.0), ("l2_lr1000", True, 1000.0), ("cos_lr5", False, 5.0),
                ("l2_lr1", True, 1.0), ("cos_lr1e6", False, 1.0e6)]


@pytest.fixture(scope="module")
def flownetc_sd():
    from understanding_flow_robustness_amd.flownets.flownetc import FlowNetC
    from understanding_flow_robustness_amd.flownets.weights import state_dict_digest, synthetic_state_dict
    sd = synthetic_state_dict(FlowNetC().state_dict(), seed=0)
    z = load_golden("flownetc_fwd_64x128")
    assert state_dict_digest(sd) == float(z["weight_digest"]), "synthetic weight generator drifted"
    return sd


@pytest.mark.parametrize("case", ["flownetc_fwd_64x128", "flownetc_fwd_128x192"])
def test_flownetc_oracle_matches_reference(oracle, flownetc_sd, case):
    from oracle import flow_oracle as fo
    z = load_golden(case)
    x1, x2 = t(z["x1"]).requires_grad_(True), t(z["x2"]).requires_grad_(True)
    flow = fo.flownetc_forward(flownetc_sd, x1, x2)
    assert_close(flow, t(z["flow"]), rtol=1e-6, atol_scale=1e-7, what="flow")
    loss = fo.flow_loss(flow, t(z["target"]))
    assert abs(float(loss) - float(z["loss"])) < 1e-6
    g1, g2 = torch.autograd.grad(loss, (x1, x2))
    assert_close(g1, t(z["g1"]), rtol=1e-5, atol_scale=1e-6, what="d loss / d frame 1")
    assert_close(g2, t(z["g2"]), rtol=1e-5, atol_scale=1e-6, what="d loss / d frame 2")


@pytest.mark.parametrize("name,l2,lr", ATTACK_CASES)
def test_attack_oracle_matches_reference_trace(oracle, flownetc_sd, name, l2, lr):
    from oracle import flow_oracle as fo
    z = load_golden("attack_flownetc_64x128")
    predict = lambda a, b: fo.flownetc_forward(flownetc_sd, a, b)
    for iters in (1, 2):
        patch = t(z["patch0"]).clone()
        a_t, a_r, patch, n, _ = fo.patch_attack(predict, t(z["tgt"]), t(z["ref"]), patch, t(z["mask"]),
                                                t(z["patch0"]), t(z["target"]), lr=lr, max_count=iters, l2=l2)
        assert n == iters
        assert_close(patch, t(z[f"{name}_it{iters}_patch"]), rtol=1e-5, atol_scale=1e-6, what="patch")
        assert_close(a_t[:, :, 20:45, 50:75], t(z[f"{name}_it{iters}_adv_tgt"]), rtol=1e-5, atol_scale=1e-6)
        assert_close(a_r[:, :, 20:45, 50:75], t(z[f"{name}_it{iters}_adv_ref"]), rtol=1e-5, atol_scale=1e-6)


def test_losses_match_reference_golden():
    """patch_attacks/losses.py:8-50 -- product implementation and oracle restatement."""
    from oracle import flow_oracle as fo
    from understanding_flow_robustness_amd import losses
    z = load_golden("losses_epe_cossim")
    pred, gt2, gt3 = t(z["pred"]), t(z["gt2"]), t(z["gt3"])
    for mod in (losses, fo):
        assert abs(mod.compute_epe(gt2, pred) - float(z["epe2"])) <= 1e-6 * float(z["epe2"])
        assert abs(mod.compute_epe(gt3, pred) - float(z["epe3"])) <= 1e-6 * float(z["epe3"])
        assert abs(mod.compute_cossim(gt2, pred) - float(z["cos2"])) <= 1e-6
        assert abs(mod.compute_cossim(gt3, pred) - float(z["cos3"])) <= 1e-6


def test_registry_mirrors_reference_choices():
    from argparse import Namespace
    from understanding_flow_robustness_amd.flownets import utils_model as um
    choices = um.get_flownet_choices()
    assert choices[:3] == ["FlowNetS", "FlowNetC", "FlowNet2"] and "RAFT" in choices and len(choices) == 12
    net = um.fetch_model(Namespace(flownet="FlowNetC"), synthetic_seed=0)
    assert not net.training and sum(p.numel() for p in net.parameters()) == 39175298   # FlowNetC.py:8
    with pytest.raises(ValueError):
        um.fetch_model(Namespace(flownet="nope"))
    with pytest.raises(NotImplementedError):
        um.fetch_model(Namespace(flownet="SpyNet"))


def test_batched_shared_patch_equals_sum_of_sample_gradients(oracle, flownetc_sd):
    """The batch extension's definition, on the oracle: B pairs, one patch, loss = batch mean ->
    the update uses the SUM over samples of d(loss)/d(adv_b)."""
    from oracle import flow_oracle as fo
    g = torch.Generator().manual_seed(7)
    tgt, ref = torch.rand(2, 3, 64, 128, generator=g), torch.rand(2, 3, 64, 128, generator=g)
    mask = torch.zeros(2, 3, 64, 128)
    mask[0, :, 10:30, 20:40] = 1
    mask[1, :, 30:50, 80:100] = 1
    patch0 = torch.rand(1, 3, 64, 128, generator=g)
    target = torch.randn(2, 2, 64, 128, generator=g)
    predict = lambda a, b: fo.flownetc_forward(flownetc_sd, a, b)
    trace = []
    p = patch0.clone()
    fo.patch_attack(predict, tgt, ref, p, mask, patch0, target, lr=5e4, max_count=1, trace=trace)
    gsum = ((trace[0]["g_tgt"] + trace[0]["g_ref"]) * (mask != 0).float()).sum(0, keepdim=True)
    assert_close(p, patch0 - torch.clamp(0.5 * 5e4 * gsum, -2, 2), rtol=1e-6, atol_scale=1e-7)
