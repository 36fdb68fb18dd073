"""Registry level of the drop-in boundary: `get_flownet_choices`, `fetch_model`, `predict_flow`
with the reference's names, argument meaning and string matching (models/utils_model.py:10-157,
:627-681).  Feature-map plumbing (return_feat_maps / overwrite_feat_maps, :160-626) belongs to the
paper's analysis scripts and is out of the hot path's scope.
"""
from __future__ import annotations

import os
import re

import torch

from .flownetc import FlowNetC
from .weights import synthetic_state_dict

_IMPLEMENTED = ("FlowNetC", "PWCNet", "PWCNet_adv_ifgsm_l2_002", "RAFT", "RAFT_adv_kitti2012_ifgsm_l2_002",
                "FlowNet2", "FlowNetS")


def get_flownet_choices():
    """utils_model.py:10-24 (verbatim list: it is the CLI's `choices=`)."""
    return ["FlowNetS", "FlowNetC", "FlowNet2", "FlowNetCFlexLarger_k3_reps3",
            "FlowNetCFlexLarger_k3_reps3_adv_ifgsm_l2_002", "FlowNetCFlexLarger_k5_reps0", "SpyNet",
            "PWCNet", "PWCNet_adv_ifgsm_l2_002", "RAFT", "RAFT_FlowNetCEncoder_WoContext",
            "RAFT_adv_kitti2012_ifgsm_l2_002"]


_CHECKPOINTS = {  # utils_model.py:100-155
    "FlowNetC": ("FlowNet2-C_checkpoint.pth.tar", "state_dict"),
    "FlowNetS": ("FlowNet2-S_checkpoint.pth.tar", "state_dict"),
    "FlowNet2": ("FlowNet2_checkpoint.pth.tar", "state_dict"),
    "PWCNet": ("pwc_net_chairs.pth.tar", None),
    "PWCNet_adv_ifgsm_l2_002": ("adv_kitti2012_pwcnet_ifgsm_l2_0.02.pth", None),
    "RAFT": ("raft-things.pth", None),
    "RAFT_adv_kitti2012_ifgsm_l2_002": ("adv_kitti2012_raft_ifgsm_l2_0.02.pth", None),
}


def _build(args, return_feat_maps):
    name = args.flownet
    if name == "FlowNetC":
        return FlowNetC(return_feat_maps=return_feat_maps)
    if name in ("PWCNet", "PWCNet_adv_ifgsm_l2_002"):
        from .pwcnet import PWCDCNet
        return PWCDCNet()
    if re.findall("^RAFT", name) and "FlowNetCEncoder" not in name:
        from .raft import RAFT
        # utils_model.py:49-69: the Namespace is mutated, callers read these fields back
        args.small = False
        args.mixed_precision = "adv" not in name
        args.alternate_corr = getattr(args, "alternate_corr", False)
        args.fnorm, args.cnorm = "instance", "batch"
        args.no_separate_context = False
        args.corr_levels, args.iters = 4, 12
        args.flowNetCEnc = False
        args.update_no_motion_downsampling = False      # reference bug: never set
        return RAFT(args)
    if name == "FlowNet2":
        from .flownet2 import FlowNet2
        return FlowNet2()
    if name == "FlowNetS":
        from .flownet2 import FlowNet2S                 # models/__init__.py:2: FlowNet2S as FlowNetS
        return FlowNet2S(return_feat_maps=return_feat_maps)
    raise NotImplementedError(
        f"{name!r} is in the reference registry but outside this build's hot-path scope "
        f"(implemented: {', '.join(_IMPLEMENTED)})")


def fetch_model(args, pretrained_path: str = "pretrained_models", return_feat_maps: bool = False,
                synthetic_seed: int | None = None) -> torch.nn.Module:
    """utils_model.py:27-157.  Loads the reference's checkpoint file; a missing file raises FileNotFoundError
    like the reference's `torch.load`.  Seeded synthetic weights (there are no checkpoints offline) are used
    only when `synthetic_seed` is passed explicitly."""
    if args.flownet not in get_flownet_choices():
        raise ValueError(f"unknown flownet {args.flownet!r}")
    net = _build(args, return_feat_maps)
    fname, key = _CHECKPOINTS.get(args.flownet, (None, None))
    path = os.path.join(str(pretrained_path), fname) if fname else None
    if synthetic_seed is None:
        if not (path and os.path.exists(path)):
            raise FileNotFoundError(
                f"fetch_model({args.flownet!r}): checkpoint {path!r} not found (pass synthetic_seed=<int> for "
                f"seeded synthetic weights)")
        weights = torch.load(path, map_location="cpu")
        if isinstance(weights, dict) and "state_dict" in weights:   # wrapped checkpoints, whatever `key` says
            weights = weights["state_dict"]
        try:
            net.load_state_dict(weights)
        except RuntimeError:
            # utils_model.py:132-142: positional copy for checkpoints saved under other key names
            own = net.state_dict()
            for (k, _), (_, v) in zip(list(own.items()), list(weights.items())):
                own[k] = v
            net.load_state_dict(own)
    else:
        net.load_state_dict(synthetic_state_dict(net.state_dict(), seed=synthetic_seed))
    return net.eval()


def predict_flow(flow_net, ref_past_img, tgt_img, ref_future_img, args, return_feat_maps=False,
                 overwrite_feat_maps=None, feat_maps_to_numpy: bool = True):
    """utils_model.py:627-681: uniform call -- RAFT takes [0,255] images and returns (low, up)."""
    if return_feat_maps or overwrite_feat_maps:
        raise NotImplementedError("feature-map capture/overwrite is analysis-only (out of scope)")
    if "RAFT" in args.flownet:
        _, flow_pred = flow_net(image1=tgt_img * 255.0, image2=ref_future_img * 255.0, test_mode=True)
        return flow_pred
    return flow_net(tgt_img, ref_future_img)
