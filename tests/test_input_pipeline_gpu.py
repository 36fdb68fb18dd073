"""GPU suite for the loaders' transforms on the device (csrc/imresize.hip, input_pipeline.py, kitti_io.py;
SURVEY.md 8 f4): byte-exact against Pillow (what the reference's `imresize` calls) at the KITTI sizes,
against the reference's own custom_transforms classes (golden) for the train and validation pipelines,
and the KITTI flow decode against the host formula."""
import random

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("src,dst", [((375, 1242), (384, 1280)), ((370, 1226), (384, 1280)), ((375, 1242), (256, 832)),
                                     ((64, 80), (73, 91)), ((64, 80), (64, 91)), ((64, 80), (70, 80)),
                                     ((97, 131), (40, 55)), ((33, 47), (33, 47)), ((120, 90), (256, 256))])
def test_imresize_is_pillow_bilinear_bit_for_bit(src, dst):
    from understanding_flow_robustness_amd import input_pipeline as ip
    rng = np.random.default_rng(src[0] + dst[1])
    img = rng.integers(0, 256, size=src + (3,), dtype=np.uint8)
    img[: src[0] // 3] = (img[: src[0] // 3] // 128) * 255
    want = np.array(Image.fromarray(img).resize((dst[1], dst[0]), resample=Image.BILINEAR))
    got = ip.imresize(torch.from_numpy(img).to(DEV), dst).cpu().numpy()
    assert got.dtype == np.uint8 and np.array_equal(got, want), f"{int((got != want).sum())} bytes differ"
    # flip-then-resize (RandomHorizontalFlip precedes the resize) through the kernel's mirrored read
    want_f = np.array(Image.fromarray(np.ascontiguousarray(img[:, ::-1])).resize((dst[1], dst[0]), resample=Image.BILINEAR))
    got_f = ip.imresize(torch.from_numpy(img).to(DEV), dst, flip=True).cpu().numpy()
    assert np.array_equal(got_f, want_f)


def test_single_channel_and_rgba():
    from understanding_flow_robustness_amd import input_pipeline as ip
    rng = np.random.default_rng(2)
    g = rng.integers(0, 256, size=(50, 70), dtype=np.uint8)
    want = np.array(Image.fromarray(g).resize((91, 64), resample=Image.BILINEAR))
    got = ip.imresize(torch.from_numpy(g[:, :, None].copy()).to(DEV), (64, 91)).cpu().numpy()[:, :, 0]
    assert np.array_equal(got, want)


def test_to_tensor_and_crop():
    from understanding_flow_robustness_amd import input_pipeline as ip
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, size=(40, 60, 3), dtype=np.uint8)
    want = torch.from_numpy(np.transpose(img.astype(np.float32), (2, 0, 1))).float() / 255       # ArrayToTensor
    got = ip.to_tensor(torch.from_numpy(img).to(DEV)).cpu()
    assert torch.equal(got, want)
    got_c = ip.to_tensor(torch.from_numpy(img).to(DEV), crop=(5, 7, 20, 31)).cpu()
    assert torch.equal(got_c, want[:, 5:25, 7:38])
    with pytest.raises(RuntimeError):
        ip.to_tensor(torch.from_numpy(img).to(DEV), crop=(30, 7, 20, 31))


def test_transform_pipelines_match_reference_classes():
    """custom_transforms.Compose([...]) of patch_attacks/main.py:199-216 with the reference's RNG streams."""
    from understanding_flow_robustness_amd import input_pipeline as ip
    z = load_golden("input_pipeline")
    imgs = [torch.from_numpy(z["imgs"][i]).to(DEV) for i in range(3)]
    for seed in (1, 2, 3, 4):
        random.seed(seed); np.random.seed(seed + 10)
        tr = ip.Compose([ip.RandomHorizontalFlip(), ip.RandomScaleCrop(h=64, w=64), ip.ArrayToTensor()])
        got = torch.stack(tr(list(imgs))).cpu()
        assert torch.equal(got, torch.from_numpy(z[f"train_seed{seed}"])), f"train pipeline, seed {seed}"
    va = ip.Compose([ip.Scale(h=96, w=160), ip.ArrayToTensor()])
    assert torch.equal(torch.stack(va(list(imgs))).cpu(), torch.from_numpy(z["valid"]))


def test_kitti_flow_decode_on_device(tmp_path):
    from understanding_flow_robustness_amd import kitti_io
    rng = np.random.default_rng(9)
    u = rng.integers(-30000, 30000, size=(37, 124)) / 64.0
    v = rng.integers(-30000, 30000, size=(37, 124)) / 64.0
    valid = (rng.random((37, 124)) > 0.4).astype(np.uint8)
    p = tmp_path / "000000_10.png"
    kitti_io.flow_write_png(str(p), u, v, valid)
    hu, hv, hvalid = kitti_io.flow_read_png(str(p))
    want = torch.FloatTensor(np.dstack((hu, hv, hvalid)).transpose(2, 0, 1))       # validation_flow.py:192-194
    got = kitti_io.flow_read_png_device(str(p), DEV).cpu()
    assert torch.equal(got, want)
