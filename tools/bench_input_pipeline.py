"""Validation-loader transform of one KITTI frame (375x1242 -> 384x1280, /255, HWC->CHW): the reference's
host path (Pillow resize + numpy/torch conversion + H2D of the float32 tensor) against the device path
(H2D of the uint8 bytes + csrc/imresize.hip).   python tools/bench_input_pipeline.py"""
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(n=50):
    from understanding_flow_robustness_amd import input_pipeline as ip
    dev = "cuda:0"
    img = np.random.default_rng(0).integers(0, 256, size=(375, 1242, 3), dtype=np.uint8)
    f32 = img.astype(np.float32)                                   # what load_as_float hands to the transforms

    def host():
        r = np.array(Image.fromarray(f32.astype("uint8")).resize((1280, 384), resample=Image.BILINEAR))
        return (torch.from_numpy(np.transpose(r, (2, 0, 1))).float() / 255).to(dev)

    def device():
        return ip.to_tensor(ip.imresize(torch.from_numpy(img).to(dev), (384, 1280)))
    assert torch.equal(host(), device())
    for name, fn in (("host (reference)", host), ("device", device)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        print(f"{name:18s}: {(time.perf_counter() - t0) / n * 1e3:6.3f} ms per frame", flush=True)


if __name__ == "__main__":
    main()
