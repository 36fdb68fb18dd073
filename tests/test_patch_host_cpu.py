"""CPU suite, part 6: host-side patch initialisation / placement (patch_attacks/utils_patch.py:236-358)
and the per-sample glue of patch_attacks/main.py::train (:363-461) against goldens captured from
the reference under fixed np.random seeds.  Placement and shapes are index outputs: bit-exact."""
import numpy as np
import torch

from conftest import load_golden, t


def test_init_patch_circle_and_circle_transform_match_reference():
    from understanding_flow_robustness_amd import utils_patch as up
    z = load_golden("patch_host_transform")
    np.random.seed(1234)
    patch, mask, shape = up.init_patch_circle(384, 0.1329)
    assert tuple(shape) == tuple(z["init_shape"]) == (1, 3, 51, 51)
    assert np.array_equal(patch, z["init_patch"]) and np.array_equal(mask, z["init_mask"])
    assert mask.dtype == np.float32 and int(mask[0, 0].sum()) == int(z["init_mask"][0, 0].sum())
    for tag, seed, dshape in (("t0", 7, (1, 3, 256, 256)), ("t1", 8, (1, 3, 384, 1280))):
        np.random.seed(seed)
        x, xm, xp, rx, ry, pshape = up.circle_transform(patch.copy(), mask.copy(), patch.copy(), dshape, shape, True)
        assert [rx, ry] == list(z[f"{tag}_loc"]) and tuple(pshape) == tuple(z[f"{tag}_shape"])
        ys, xs = slice(ry, ry + pshape[-2]), slice(rx, rx + pshape[-1])
        assert np.array_equal(x[:, :, ys, xs], z[f"{tag}_patch"])
        assert np.array_equal(xm[:, :, ys, xs], z[f"{tag}_mask"])
        assert np.array_equal(xp[:, :, ys, xs], z[f"{tag}_init"])
        assert np.allclose([x.sum(), xm.sum(), xp.sum()], z[f"{tag}_sum"], rtol=0, atol=0)
        assert x.shape == dshape


def test_train_glue_matches_reference_on_cpu(oracle):
    """circle_transform -> attack (CPU oracle) -> mask*patch -> crop -> zoom back, one loader item."""
    from oracle import flow_oracle as fo
    from understanding_flow_robustness_amd import utils_patch as up
    from understanding_flow_robustness_amd.flownets.flownetc import FlowNetC
    from understanding_flow_robustness_amd.flownets.weights import state_dict_digest, synthetic_state_dict
    z = load_golden("patch_host_transform")
    sd = synthetic_state_dict(FlowNetC().state_dict(), seed=0)
    assert state_dict_digest(sd) == float(z["weight_digest"])
    tgt, ref = t(z["train_tgt"]), t(z["train_ref"])
    np.random.seed(99)
    p0, m0, sh0 = up.init_patch_circle(128, 0.2)
    assert np.array_equal(p0, z["train_patch0"]) and np.array_equal(m0, z["train_mask0"])
    predict = lambda a, b: fo.flownetc_forward(sd, a, b)
    np.random.seed(5)
    with torch.no_grad():
        flow = predict(tgt, ref)
    patch, mask, init, rx, ry, pshape = up.circle_transform(p0.copy(), m0.copy(), p0.copy(), tuple(tgt.shape), sh0, True)
    patch_t, mask_t, init_t = torch.FloatTensor(patch), torch.FloatTensor(mask), torch.FloatTensor(init)
    fo.patch_attack(predict, tgt, ref, patch_t, mask_t, init_t, -flow, lr=1e5, max_count=2)
    p1, m1, i1, sh1 = up.crop_and_restore(torch.mul(mask_t, patch_t).numpy(), mask_t.numpy(), init_t.numpy(), rx, ry,
                                          pshape, sh0)
    assert tuple(sh1) == tuple(z["train_shape1"])
    assert np.array_equal(m1, z["train_mask1"]) and np.array_equal(i1, z["train_init1"])
    assert np.array_equal(p1, z["train_patch1"])


def test_square_transform_matches_reference():
    """`--patch_type square` (utils_patch.py:781-846): quarter turns applied in place, corner, and the position of the
    `np.random` stream afterwards -- all exact."""
    from understanding_flow_robustness_amd import utils_patch as up
    z = load_golden("patch_square")
    np.random.seed(321)
    patch, shape = up.init_patch_square(384, 0.1329)
    assert tuple(shape) == tuple(z["init_shape"]) == (1, 3, 51, 51) and np.array_equal(patch, z["init_patch"])
    mask = np.ones(shape)
    for tag, seed, dshape, norot in (("t0", 7, (1, 3, 256, 256), False), ("t1", 8, (1, 3, 384, 1280), False),
                                     ("t2", 9, (1, 3, 384, 1280), True)):
        np.random.seed(seed)
        p, m, pi = patch.copy(), mask.copy(), patch.copy()
        x, xm, xp, rx, ry = up.square_transform(p, m, pi, dshape, shape, norotate=norot)
        assert [rx, ry] == list(z[f"{tag}_loc"]) and x.shape == dshape
        assert np.array_equal(p, z[f"{tag}_rotated_in_place"]) and np.array_equal(pi, p)
        assert np.array_equal(x[:, :, ry:ry + 51, rx:rx + 51], z[f"{tag}_patch"])
        assert np.array_equal([x.sum(), xm.sum(), xp.sum()], z[f"{tag}_sum"])
        assert np.random.random() == float(z[f"{tag}_next_draw"])


def test_square_train_glue_matches_reference_on_cpu(oracle):
    """main.py:383-461 with `patch_type="square"`: square_transform -> attack (CPU oracle) -> mask * patch -> crop (no zoom)."""
    from oracle import flow_oracle as fo
    from understanding_flow_robustness_amd import utils_patch as up
    from understanding_flow_robustness_amd.flownets.flownetc import FlowNetC
    from understanding_flow_robustness_amd.flownets.weights import state_dict_digest, synthetic_state_dict
    z = load_golden("patch_square")
    sd = synthetic_state_dict(FlowNetC().state_dict(), seed=0)
    assert state_dict_digest(sd) == float(z["weight_digest"])
    tgt, ref = t(z["train_tgt"]), t(z["train_ref"])
    np.random.seed(77)
    p0, sh0 = up.init_patch_square(128, 0.2)
    assert np.array_equal(p0, z["train_patch0"])
    predict = lambda a, b: fo.flownetc_forward(sd, a, b)
    np.random.seed(6)
    with torch.no_grad():
        flow = predict(tgt, ref)
    patch, mask, init, rx, ry = up.square_transform(p0.copy(), np.ones(sh0), p0.copy(), tuple(tgt.shape), sh0)
    patch_t, mask_t, init_t = torch.FloatTensor(patch), torch.FloatTensor(mask), torch.FloatTensor(init)
    fo.patch_attack(predict, tgt, ref, patch_t, mask_t, init_t, -flow, lr=1e5, max_count=2)
    p1, m1, i1, sh1 = up.crop_and_restore(torch.mul(mask_t, patch_t).numpy(), mask_t.numpy(), init_t.numpy(), rx, ry, sh0, sh0)
    assert tuple(sh1) == tuple(z["train_shape1"])
    assert np.array_equal(m1, z["train_mask1"]) and np.array_equal(i1, z["train_init1"])
    assert np.array_equal(p1, z["train_patch1"])
