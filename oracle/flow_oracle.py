"""CPU restatement of the model-level hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional (state_dict in, tensors out) torch-fp32 restatements of the reference's flow networks and
attack loops, with the native operators routed to the C oracle (oracle/oracle_ops.py).  Pinned by
golden fixtures generated from the reference's own Python modules (tests/golden/make_golden_models.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this file.

Each function cites the reference file:line it follows.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import oracle_ops as oo

_RGB_MEAN = torch.tensor([0.40066648, 0.39482617, 0.3784785], dtype=torch.float64).view(1, 3, 1, 1)


def _lrelu(x):
    return F.leaky_relu(x, 0.1)


def _conv(sd, name, x, stride=1, act=True):
    w = sd[name + ".weight"]
    y = F.conv2d(x, w, sd.get(name + ".bias"), stride=stride, padding=(w.shape[-1] - 1) // 2)
    return _lrelu(y) if act else y


def _deconv(sd, name, x, act=True):
    y = F.conv_transpose2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=2, padding=1)
    return _lrelu(y) if act else y


def correlate(a, b, patch=21, dil_patch=2):
    """models/submodules.py:124-138."""
    out = oo.spatial_correlation_sample(a.contiguous(), b.contiguous(), kernel_size=1, patch_size=patch,
                                        stride=1, padding=0, dilation_patch=dil_patch)
    bsz, ph, pw, h, w = out.shape
    return out.view(bsz, ph * pw, h, w) / a.size(1)


def flownetc_forward(sd, x1, x2, div_flow=20.0):
    """models/FlowNetC.py:81-197 (eval branch)."""
    x1 = (x1.double() - _RGB_MEAN).float()                      # :73-79,:93-94
    x2 = (x2.double() - _RGB_MEAN).float()
    c1a = _conv(sd, "conv1.0", x1, 2); c2a = _conv(sd, "conv2.0", c1a, 2); c3a = _conv(sd, "conv3.0", c2a, 2)
    c1b = _conv(sd, "conv1.0", x2, 2); c2b = _conv(sd, "conv2.0", c1b, 2); c3b = _conv(sd, "conv3.0", c2b, 2)
    corr = _lrelu(correlate(c3a, c3b))                          # :134,:139
    redir = _conv(sd, "conv_redir.0", c3a)                      # :142
    c3_1 = _conv(sd, "conv3_1.0", torch.cat((redir, corr), 1))
    c4 = _conv(sd, "conv4_1.0", _conv(sd, "conv4.0", c3_1, 2))
    c5 = _conv(sd, "conv5_1.0", _conv(sd, "conv5.0", c4, 2))
    c6 = _conv(sd, "conv6_1.0", _conv(sd, "conv6.0", c5, 2))
    flow6 = _conv(sd, "predict_flow6", c6, act=False)
    cat5 = torch.cat((c5, _deconv(sd, "deconv5.0", c6), _deconv(sd, "upsampled_flow6_to_5", flow6, act=False)), 1)
    flow5 = _conv(sd, "predict_flow5", cat5, act=False)
    cat4 = torch.cat((c4, _deconv(sd, "deconv4.0", cat5), _deconv(sd, "upsampled_flow5_to_4", flow5, act=False)), 1)
    flow4 = _conv(sd, "predict_flow4", cat4, act=False)
    cat3 = torch.cat((c3_1, _deconv(sd, "deconv3.0", cat4), _deconv(sd, "upsampled_flow4_to_3", flow4, act=False)), 1)
    flow3 = _conv(sd, "predict_flow3", cat3, act=False)
    cat2 = torch.cat((c2a, _deconv(sd, "deconv2.0", cat3), _deconv(sd, "upsampled_flow3_to_2", flow3, act=False)), 1)
    flow2 = _conv(sd, "predict_flow2", cat2, act=False)
    return F.interpolate(flow2 * div_flow, scale_factor=4, mode="bilinear", align_corners=False)  # :194-197


def _pwc_warp(x, flo):
    """models/PWCNet.py:164-204."""
    B, C, H, W = x.shape
    xx = torch.arange(0, W).view(1, -1).repeat(H, 1).view(1, 1, H, W).repeat(B, 1, 1, 1)
    yy = torch.arange(0, H).view(-1, 1).repeat(1, W).view(1, 1, H, W).repeat(B, 1, 1, 1)
    vgrid = torch.cat((xx, yy), 1).float() + flo
    vx = 2.0 * vgrid[:, 0] / max(W - 1, 1) - 1.0
    vy = 2.0 * vgrid[:, 1] / max(H - 1, 1) - 1.0
    vgrid = torch.stack((vx, vy), dim=3)
    output = F.grid_sample(x, vgrid, align_corners=False)
    mask = F.grid_sample(torch.ones_like(x), vgrid, align_corners=False)
    return output * (mask >= 0.0001).float()


def pwcnet_forward(sd, im1, im2):
    """models/PWCNet.py:225-367 (eval branch)."""
    def c(name, x, stride=1, dil=1):
        w = sd[name + ".0.weight"]
        return _lrelu(F.conv2d(x, w, sd[name + ".0.bias"], stride=stride, padding=dil, dilation=dil))

    def pyramid(im):
        x = im.flip(1)                                               # :230-231 RGB -> BGR
        out = []
        for a, b, cc in (("1a", "1aa", "1b"), ("2a", "2aa", "2b"), ("3a", "3aa", "3b"), ("4a", "4aa", "4b"),
                         ("5a", "5aa", "5b"), ("6aa", "6a", "6b")):
            x = c("conv" + cc, c("conv" + b, c("conv" + a, x, 2)))
            out.append(x)
        return out

    f1, f2 = pyramid(im1), pyramid(im2)

    def decode(lvl, x):
        for i in range(5):
            x = torch.cat((c(f"conv{lvl}_{i}", x), x), 1)
        flow = F.conv2d(x, sd[f"predict_flow{lvl}.weight"], sd[f"predict_flow{lvl}.bias"], padding=1)
        return x, flow

    def dec(name, x):
        return F.conv_transpose2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=2, padding=1)

    corr = _lrelu(correlate(f1[5], f2[5], 9, 1))
    x, flow = decode(6, corr)
    scale = {5: 0.625, 4: 1.25, 3: 2.5, 2: 5.0}
    for lvl in (5, 4, 3, 2):
        up_flow, up_feat = dec(f"deconv{lvl + 1}", flow), dec(f"upfeat{lvl + 1}", x)
        warped = _pwc_warp(f2[lvl - 1], up_flow * scale[lvl])
        corr = _lrelu(correlate(f1[lvl - 1], warped, 9, 1))
        x, flow = decode(lvl, torch.cat((corr, f1[lvl - 1], up_flow, up_feat), 1))
    y = c("dc_conv4", c("dc_conv3", c("dc_conv2", c("dc_conv1", x), dil=2), dil=4), dil=8)
    y = c("dc_conv6", c("dc_conv5", y, dil=16))
    flow2 = flow + F.conv2d(y, sd["dc_conv7.weight"], sd["dc_conv7.bias"], padding=1)
    return 20 * F.interpolate(flow2, scale_factor=4, mode="bilinear", align_corners=False)


# ------------------------------------------------------------------------------------------- FlowNet2
def _fn2_refine(sd, p, c6, skips, inter=False):
    flow, x = _conv(sd, p + "predict_flow6", c6, act=False), c6
    for lvl, skip in zip((5, 4, 3, 2), skips):
        up = _deconv(sd, f"{p}upsampled_flow{lvl + 1}_to_{lvl}", flow, act=False)
        x = torch.cat((skip, _deconv(sd, f"{p}deconv{lvl}.0", x), up), 1)
        feat = _conv(sd, f"{p}inter_conv{lvl}.0", x, act=False) if inter else x
        flow = _conv(sd, f"{p}predict_flow{lvl}", feat, act=False)
    return flow


def _fn2_flownetc(sd, p, x):
    """models/flownet2/FlowNetC.py:69-131."""
    def tower(im):
        c2 = _conv(sd, p + "conv2.0", _conv(sd, p + "conv1.0", im, 2), 2)
        return c2, _conv(sd, p + "conv3.0", c2, 2)
    c2a, c3a = tower(x[:, 0:3])
    _, c3b = tower(x[:, 3:])
    corr = _lrelu(correlate(c3a, c3b))
    c3_1 = _conv(sd, p + "conv3_1.0", torch.cat((_conv(sd, p + "conv_redir.0", c3a), corr), 1))
    c4 = _conv(sd, p + "conv4_1.0", _conv(sd, p + "conv4.0", c3_1, 2))
    c5 = _conv(sd, p + "conv5_1.0", _conv(sd, p + "conv5.0", c4, 2))
    c6 = _conv(sd, p + "conv6_1.0", _conv(sd, p + "conv6.0", c5, 2))
    return _fn2_refine(sd, p, c6, (c5, c4, c3_1, c2a))


def _fn2_flownets(sd, p, x):
    """models/flownet2/FlowNetS.py:58-104."""
    c2 = _conv(sd, p + "conv2.0", _conv(sd, p + "conv1.0", x, 2), 2)
    c3 = _conv(sd, p + "conv3_1.0", _conv(sd, p + "conv3.0", c2, 2))
    c4 = _conv(sd, p + "conv4_1.0", _conv(sd, p + "conv4.0", c3, 2))
    c5 = _conv(sd, p + "conv5_1.0", _conv(sd, p + "conv5.0", c4, 2))
    c6 = _conv(sd, p + "conv6_1.0", _conv(sd, p + "conv6.0", c5, 2))
    return _fn2_refine(sd, p, c6, (c5, c4, c3, c2))


def _fn2_flownetsd(sd, p, x):
    """models/flownet2/FlowNetSD.py:67-126."""
    c0 = _conv(sd, p + "conv0.0", x)
    c1 = _conv(sd, p + "conv1_1.0", _conv(sd, p + "conv1.0", c0, 2))
    c2 = _conv(sd, p + "conv2_1.0", _conv(sd, p + "conv2.0", c1, 2))
    c3 = _conv(sd, p + "conv3_1.0", _conv(sd, p + "conv3.0", c2, 2))
    c4 = _conv(sd, p + "conv4_1.0", _conv(sd, p + "conv4.0", c3, 2))
    c5 = _conv(sd, p + "conv5_1.0", _conv(sd, p + "conv5.0", c4, 2))
    c6 = _conv(sd, p + "conv6_1.0", _conv(sd, p + "conv6.0", c5, 2))
    return _fn2_refine(sd, p, c6, (c5, c4, c3, c2), inter=True)


def _fn2_fusion(sd, p, x):
    """models/flownet2/FlowNetFusion.py:48-71."""
    c0 = _conv(sd, p + "conv0.0", x)
    c1 = _conv(sd, p + "conv1_1.0", _conv(sd, p + "conv1.0", c0, 2))
    c2 = _conv(sd, p + "conv2_1.0", _conv(sd, p + "conv2.0", c1, 2))
    flow2 = _conv(sd, p + "predict_flow2", c2, act=False)
    cat1 = torch.cat((c1, _deconv(sd, p + "deconv1.0", c2), _deconv(sd, p + "upsampled_flow2_to_1", flow2, act=False)), 1)
    flow1 = _conv(sd, p + "predict_flow1", _conv(sd, p + "inter_conv1.0", cat1, act=False), act=False)
    cat0 = torch.cat((c0, _deconv(sd, p + "deconv0.0", cat1), _deconv(sd, p + "upsampled_flow1_to_0", flow1, act=False)), 1)
    return _conv(sd, p + "predict_flow0", _conv(sd, p + "inter_conv0.0", cat0, act=False), act=False)


def flownet2_forward(sd, x1, x2, div_flow=20.0):
    """models/flownet2_models.py:122-205, native ops on the C oracle."""
    resample = lambda img, flow: oo.Resample2dFunction.apply(img.contiguous(), flow.contiguous(), 1, True)
    cnorm = lambda t: oo.ChannelNormFunction.apply(t.contiguous(), 2)
    x1 = (x1.double() - _RGB_MEAN).float()
    x2 = (x2.double() - _RGB_MEAN).float()
    x = torch.cat((x1, x2), dim=1)
    up_bl = lambda f: F.interpolate(f, scale_factor=4, mode="bilinear", align_corners=False)
    up_nn = lambda f: F.interpolate(f, scale_factor=4, mode="nearest")

    def stage(flow):
        res = resample(x[:, 3:], flow)
        return torch.cat((x, res, flow / div_flow, cnorm(x[:, :3] - res)), dim=1)

    flow_c = up_bl(_fn2_flownetc(sd, "flownetc.", x) * div_flow)
    flow_s1 = up_bl(_fn2_flownets(sd, "flownets_1.", stage(flow_c)) * div_flow)
    flow_s2 = up_nn(_fn2_flownets(sd, "flownets_2.", stage(flow_s1)) * div_flow)
    norm_s2 = cnorm(flow_s2)
    err_s2 = cnorm(x[:, :3] - resample(x[:, 3:], flow_s2))
    flow_sd = up_nn(_fn2_flownetsd(sd, "flownets_d.", x) / div_flow)
    norm_sd = cnorm(flow_sd)
    err_sd = cnorm(x[:, :3] - resample(x[:, 3:], flow_sd))
    return _fn2_fusion(sd, "flownetfusion.", torch.cat((x[:, :3], flow_sd, flow_s2, norm_sd, norm_s2, err_sd, err_s2), 1))


def flownet2s_forward(sd, x1, x2):
    """models/FlowNet2S.py:62-108 (the registry's `FlowNetS`, models/__init__.py:2): own RGB mean in float64, the
    FlowNetS trunk on cat(x1, x2), eval output `upsample1(flow2 * 20)`."""
    mean = torch.tensor((0.4114511, 0.43205959, 0.45015125), dtype=torch.float64).view(1, 3, 1, 1)
    x = torch.cat(((x1.double() - mean).float(), (x2.double() - mean).float()), dim=1)
    flow2 = _fn2_flownets(sd, "", x)
    return F.interpolate(flow2 * 20, scale_factor=4, mode="bilinear", align_corners=False)


# ------------------------------------------------------------------------------------------- RAFT
def _raft_norm(sd, prefix, x, kind):
    if kind == "instance":                                           # nn.InstanceNorm2d: no affine, no stats
        return F.instance_norm(x)
    if kind == "batch":                                              # eval mode: running statistics
        return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                            sd[prefix + ".weight"], sd[prefix + ".bias"], training=False)
    return x


def _raft_encoder(sd, p, x, kind):
    """models/raft/extractor.py:142-215 (BasicEncoder) with ResidualBlock :5-78."""
    def conv(name, t, stride=1):
        w = sd[f"{p}.{name}.weight"]
        return F.conv2d(t, w, sd[f"{p}.{name}.bias"], stride=stride, padding=(w.shape[-1] - 1) // 2)

    x = F.relu(_raft_norm(sd, f"{p}.norm1", conv("conv1", x, 2), kind))
    for layer, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        for blk, s in ((0, stride), (1, 1)):
            q = f"{layer}.{blk}"
            y = F.relu(_raft_norm(sd, f"{p}.{q}.norm1", conv(q + ".conv1", x, s), kind))
            y = F.relu(_raft_norm(sd, f"{p}.{q}.norm2", conv(q + ".conv2", y), kind))
            if s != 1:
                # norm3 is registered twice (extractor.py:66-68); a state_dict load leaves the values
                # of the later key, downsample.1.*, in the shared module
                x = _raft_norm(sd, f"{p}.{q}.downsample.1", conv(q + ".downsample.0", x, s), kind)
            x = F.relu(x + y)
    return conv("conv2", x)


def _raft_lookup(pyramid, coords, r=4):
    """models/raft/corr.py:72-96 + utils/utils.py:62-77 (grid_sample, align_corners=True)."""
    coords = coords.permute(0, 2, 3, 1)
    B, H, W, _ = coords.shape
    out = []
    for i, corr in enumerate(pyramid):
        d = torch.linspace(-r, r, 2 * r + 1, dtype=coords.dtype, device=coords.device)
        delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1)
        cl = coords.reshape(B * H * W, 1, 1, 2) / 2 ** i + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
        hh, ww = corr.shape[-2:]
        xg = 2 * cl[..., 0:1] / (ww - 1) - 1
        yg = 2 * cl[..., 1:2] / (hh - 1) - 1
        s = F.grid_sample(corr, torch.cat([xg, yg], dim=-1), align_corners=True)
        out.append(s.view(B, H, W, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().to(coords.dtype)     # `.float()` in the reference


def raft_forward(sd, image1, image2, iters=12, levels=4, radius=4, alternate_corr=False):
    """models/raft/raft.py:124-233, test_mode=True; returns (flow_low, flow_up).  Runs in the dtype and on the device
    of its inputs: float32 on the CPU is the reference's arithmetic operation for operation (the `.float()` casts of
    the reference are no-ops there); float64 serves as the conditioning-free truth of the RAFT gradient tests, and
    HIP tensors give the pure-torch spelling (grid_sample lookup, torch GRU, unfold upsampling) on the same device
    as the product's kernels."""
    dt, dev = image1.dtype, image1.device
    def conv(name, t, pad=None):
        w = sd[name + ".weight"]
        pad = ((w.shape[-2] - 1) // 2, (w.shape[-1] - 1) // 2) if pad is None else pad
        return F.conv2d(t, w, sd[name + ".bias"], padding=pad)

    image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
    image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
    f = _raft_encoder(sd, "fnet", torch.cat([image1, image2], 0), "instance")
    B = image1.shape[0]
    fmap1, fmap2 = f[:B].to(dt), f[B:].to(dt)
    _, C, H, W = fmap1.shape
    if alternate_corr:
        f2_levels = [fmap2]
        for _ in range(levels - 1):
            f2_levels.append(F.avg_pool2d(f2_levels[-1], 2, stride=2))
    else:
        corr = torch.matmul(fmap1.view(B, C, H * W).transpose(1, 2), fmap2.view(B, C, H * W))
        corr = (corr.view(B, H, W, 1, H, W) / torch.sqrt(torch.tensor(C, device=dev).to(dt))).reshape(B * H * W, 1, H, W)
        pyramid = [corr]
        for _ in range(levels - 1):
            pyramid.append(F.avg_pool2d(pyramid[-1], 2, stride=2))
    cnet = _raft_encoder(sd, "cnet", image1, "batch")
    net, inp = torch.tanh(cnet[:, :128]), torch.relu(cnet[:, 128:])
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    coords0 = torch.stack([xs, ys], dim=0).to(dt)[None].repeat(B, 1, 1, 1)
    coords1 = coords0.clone()
    flow_up = None
    ub = "update_block."
    for _ in range(iters):
        coords1 = coords1.detach()
        if alternate_corr:                                           # corr.py:117-137 on the C oracle
            outs = []
            for i in range(levels):
                c_i = (coords1.permute(0, 2, 3, 1) / 2 ** i).reshape(B, 1, H, W, 2).contiguous()
                (o,) = oo.altcorr_forward(fmap1.detach().permute(0, 2, 3, 1).contiguous(),
                                          f2_levels[i].detach().permute(0, 2, 3, 1).contiguous(), c_i, radius)
                outs.append(o.squeeze(1))
            corr_feat = torch.stack(outs, dim=1).reshape(B, -1, H, W) / torch.sqrt(torch.tensor(C, device=dev).to(dt))
        else:
            corr_feat = _raft_lookup(pyramid, coords1, radius)
        flow = coords1 - coords0
        cor = F.relu(conv(ub + "encoder.convc1", corr_feat))
        cor = F.relu(conv(ub + "encoder.convc2", cor))
        flo = F.relu(conv(ub + "encoder.convf2", F.relu(conv(ub + "encoder.convf1", flow))))
        mf = torch.cat([F.relu(conv(ub + "encoder.conv", torch.cat([cor, flo], 1))), flow], 1)
        x = torch.cat([inp, mf], 1)
        for tag in ("1", "2"):                                       # update.py:35-73
            hx = torch.cat([net, x], 1)
            z = torch.sigmoid(conv(ub + "gru.convz" + tag, hx))
            r = torch.sigmoid(conv(ub + "gru.convr" + tag, hx))
            q = torch.tanh(conv(ub + "gru.convq" + tag, torch.cat([r * net, x], 1)))
            net = (1 - z) * net + z * q
        delta = conv(ub + "flow_head.conv2", F.relu(conv(ub + "flow_head.conv1", net)))
        mask = 0.25 * conv(ub + "mask.2", F.relu(conv(ub + "mask.0", net)))
        coords1 = coords1 + delta
        fl = coords1 - coords0                                       # raft.py:111-122
        m = torch.softmax(mask.view(B, 1, 9, 8, 8, H, W), dim=2)
        up = F.unfold(8 * fl, [3, 3], padding=1).view(B, 2, 9, 1, 1, H, W)
        flow_up = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(B, 2, 8 * H, 8 * W)
    return coords1 - coords0, flow_up


# ------------------------------------------------------------------------------------------- attack
def flow_loss(flow, target, l2=False):
    """patch_attacks/main.py:557-566."""
    if l2:
        return (torch.sum((flow - target) ** 2, dim=1) + 1e-8).sqrt().mean()
    return (1 - F.cosine_similarity(flow, target)).mean()


def patch_attack(predict, tgt, ref, patch, mask, patch_init, target, lr=1e3, alpha=0.0, max_count=2,
                 l2=False, clamp=(0.0, 1.0), trace=None):
    """patch_attacks/main.py:523-613 for a `predict(adv_tgt, adv_ref) -> flow` callable.

    `patch` is updated in place like the reference's patch_var; returns
    (adv_tgt, adv_ref, patch, executed_iterations, last_loss).  B = 1 is the reference's arithmetic
    exactly (it adds the image gradient outside the mask too, where nothing ever reads the patch); with
    per-sample canvas patches ([B,3,H,W]) every sample is its own reference attack.  (A [1,3,H,W] canvas patch
    with B > 1 sums masked canvas gradients: kept for the CPU tests of round 1; ONE patch behind several
    pairs is `patch_attack_placed`, in patch coordinates.)
    """
    adv_tgt = (1 - mask) * tgt + mask * patch                   # :537-542
    adv_ref = (1 - mask) * ref + mask * patch
    count, loss_scalar = 0, 1.0
    while loss_scalar > 0.1:                                    # :546
        count += 1
        adv_tgt = adv_tgt.detach().requires_grad_(True)
        adv_ref = adv_ref.detach().requires_grad_(True)
        flow = predict(adv_tgt, adv_ref)
        loss_data = flow_loss(flow, target, l2)
        loss_reg = F.l1_loss(mask * patch, mask * patch_init)   # :568-570 (no gradient path to the update)
        loss = (1 - alpha) * loss_data + alpha * loss_reg
        g_tgt, g_ref = torch.autograd.grad(loss, (adv_tgt, adv_ref))
        g = g_tgt + g_ref
        if patch.shape[0] == 1 and g.shape[0] > 1:
            g = (g * (mask != 0).float()).sum(0, keepdim=True)
        patch -= torch.clamp(0.5 * lr * g, -2, 2)               # :581-583
        adv_tgt = torch.clamp((1 - mask) * tgt + mask * patch, *clamp)   # :585-600
        adv_ref = torch.clamp((1 - mask) * ref + mask * patch, *clamp)
        loss_scalar = float(loss.detach())                      # :605
        if trace is not None:
            trace.append(dict(loss=loss_scalar, patch=patch.clone(), adv_tgt=adv_tgt.detach().clone(),
                              adv_ref=adv_ref.detach().clone(), g_tgt=g_tgt.clone(), g_ref=g_ref.clone()))
        if count > max_count - 1:                               # :610-611
            break
    return adv_tgt.detach(), adv_ref.detach(), patch, count, loss_scalar


def place(patch_p, origins, H, W):
    """[1,3,ph,pw] at per-pair (row, column) origins -> canvases [B,3,H,W], zero elsewhere."""
    ph, pw = patch_p.shape[-2:]
    out = patch_p.new_zeros(len(origins), patch_p.shape[1], H, W)
    for b, (oy, ox) in enumerate(origins):
        out[b, :, oy:oy + ph, ox:ox + pw] = patch_p[0]
    return out


def patch_attack_placed(predict, tgt, ref, patch_p, mask_p, origins, target, lr=1e3, max_count=2, l2=False,
                        clamp=(0.0, 1.0), groups=1, trace=None):
    """The batch extension of patch_attacks/main.py:523-613: ONE patch in PATCH coordinates --
    what the reference carries between samples is the canvas cropped at the placement (ry, rx) back to
    `patch_shape`, main.py:396-424 -- behind B pairs that show it at `origins[b] = (row, column)`.
    Per iteration: loss = mean over all B*H*W pixels; every pair's pre-clamp gradient (g_tgt + g_ref) is cropped at
    its placement, masked by [mask_p != 0], and summed -- pairs of a group in ascending order, then the groups in
    ascending order (`groups` = the ranks of a sharded run: same summation tree, bit for bit) -- before
    `P -= clamp(0.5*lr*G, +-2)`.  `patch_p` [1,3,ph,pw] is updated in place.
    Returns (adv_tgt, adv_ref, patch_p, executed_iterations, last_loss)."""
    B, _, H, W = tgt.shape
    ph, pw = patch_p.shape[-2:]
    mask = place(mask_p, origins, H, W)
    shown = (mask_p != 0).float()
    adv_tgt = (1 - mask) * tgt + mask * place(patch_p, origins, H, W)        # :537-542
    adv_ref = (1 - mask) * ref + mask * place(patch_p, origins, H, W)
    count, loss_scalar = 0, 1.0
    per = B // groups
    while loss_scalar > 0.1:                                               # :546
        count += 1
        adv_tgt = adv_tgt.detach().requires_grad_(True)
        adv_ref = adv_ref.detach().requires_grad_(True)
        loss = flow_loss(predict(adv_tgt, adv_ref), target, l2)
        g_tgt, g_ref = torch.autograd.grad(loss, (adv_tgt, adv_ref))
        G = torch.zeros_like(patch_p)
        for gi in range(groups):
            row = torch.zeros_like(patch_p)
            for b in range(gi * per, (gi + 1) * per):
                oy, ox = origins[b]
                row = row + (g_tgt[b:b + 1, :, oy:oy + ph, ox:ox + pw] + g_ref[b:b + 1, :, oy:oy + ph, ox:ox + pw])
            G = G + row * shown
        patch_p -= torch.clamp(0.5 * lr * G, -2, 2)                        # :581-583
        canvas = place(patch_p, origins, H, W)
        adv_tgt = torch.clamp((1 - mask) * tgt + mask * canvas, *clamp)    # :585-600
        adv_ref = torch.clamp((1 - mask) * ref + mask * canvas, *clamp)
        loss_scalar = float(loss.detach())                                 # :605
        if trace is not None:
            trace.append(dict(loss=loss_scalar, patch=patch_p.clone(), G=G.clone(), g_tgt=g_tgt.clone(), g_ref=g_ref.clone()))
        if count > max_count - 1:                                          # :610-611
            break
    return adv_tgt.detach(), adv_ref.detach(), patch_p, count, loss_scalar


def compute_flow_loss(flow_output, ground_truth, flow_loss="cossim"):
    """global_attacks/perturb_model.py:128-145 (the part after predict_flow)."""
    if flow_loss == "cossim":
        loss = 1 - F.cosine_similarity(flow_output, ground_truth[:, :2, ...])
    elif flow_loss == "l2":
        loss = (torch.sum((flow_output - ground_truth[:, :2, ...]) ** 2, dim=1) + 10e-8).sqrt()
    elif flow_loss == "l1":
        loss = (flow_output - ground_truth[:, :2, ...]).abs()
    else:
        raise NotImplementedError
    if ground_truth.shape[1] == 3:
        valid = ground_truth[:, 2, ...]
        return (loss * valid).sum() / (valid.sum() + 1e-8)
    return loss.mean()


def universal_attack(predict, img0, img1, delta, target, n_step=10, lr=2e-3, eps=0.02, flow_loss="cossim",
                     method="ifgsm", mode="both", ascent=False, shared=False):
    """global_attacks/universal_perturbation.py:452-530.  `shared=True` is the build's batch
    extension (one [2,3,H,W] perturbation, direction from the summed gradient)."""
    if shared:
        d = delta.reshape(-1, 2, 3, *img0.shape[-2:])[0].clone()
        adv0, adv1 = torch.clamp(img0 + d[0], 0, 1), torch.clamp(img1 + d[1], 0, 1)
    else:
        adv0 = torch.clamp(img0 + delta[0, 0], 0.0, 1.0)            # :667-675
        adv1 = torch.clamp(img1 + delta[:, 1], 0.0, 1.0)
    sgn = -1.0 if not ascent else 1.0
    n0 = n1 = None
    for _ in range(n_step):
        adv0 = adv0.detach().requires_grad_(True)
        adv1 = adv1.detach().requires_grad_(True)
        loss = compute_flow_loss(predict(adv0, adv1), target, flow_loss)
        g0, g1 = torch.autograd.grad(loss, (adv0, adv1), allow_unused=True)
        g1 = torch.zeros_like(img1) if g1 is None else g1
        if shared:
            g0, g1 = g0.sum(0), g1.sum(0)
        d0 = torch.sign(g0) if "ifgsm" in method else g0
        d1 = torch.sign(g1) if "ifgsm" in method else g1
        s0 = lr * d0 if mode in ("both", "left") else torch.zeros_like(d0)
        s1 = lr * d1 if mode in ("both", "right") else torch.zeros_like(d1)
        if shared:
            d[0] = torch.clamp(d[0] + sgn * s0, -eps, eps)
            d[1] = torch.clamp(d[1] + sgn * s1, -eps, eps)
            adv0, adv1 = torch.clamp(img0 + d[0], 0, 1), torch.clamp(img1 + d[1], 0, 1)
            continue
        adv0 = torch.clamp(adv0.detach() + sgn * s0, 0.0, 1.0)
        adv1 = torch.clamp(adv1.detach() + sgn * s1, 0.0, 1.0)
        n0 = torch.clamp(adv0 - img0, -eps, eps)
        n1 = torch.clamp(adv1 - img1, -eps, eps)
        adv0, adv1 = img0 + n0, img1 + n1
    if shared:
        return adv0.detach(), adv1.detach(), d
    return adv0.detach(), adv1.detach(), torch.stack([n0, n1], dim=1)


# ------------------------------------------------------------------------------------------- metrics
def compute_epe(gt, pred):
    """patch_attacks/losses.py:8-28."""
    _, _, h_pred, w_pred = pred.size()
    bs, nc, h_gt, w_gt = gt.size()
    pred = F.interpolate(pred, size=(h_gt, w_gt), mode="bilinear", align_corners=False)
    u = pred[:, 0] * (w_gt / w_pred)
    v = pred[:, 1] * (h_gt / h_pred)
    epe = torch.sqrt((gt[:, 0] - u) ** 2 + (gt[:, 1] - v) ** 2)
    if nc == 3:
        valid = gt[:, 2]
        return float((epe * valid).sum() / (valid.sum() + 1e-8))
    return float(epe.sum() / (bs * h_gt * w_gt))


def compute_cossim(gt, pred):
    """patch_attacks/losses.py:31-50."""
    bs, nc, h_gt, w_gt = gt.size()
    pred = F.interpolate(pred, size=(h_gt, w_gt), mode="bilinear", align_corners=False)
    sim = F.cosine_similarity(gt[:, :2], pred)
    if nc == 3:
        valid = gt[:, 2]
        return float((sim * valid).sum() / (valid.sum() + 1e-8))
    return float(sim.sum() / (bs * h_gt * w_gt))
