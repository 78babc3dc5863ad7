"""Generates tests/golden/loss_fixtures.npz by IMPORTING the reference's loss code on CPU in the build container (stubs for
jaxtyping, lpips, skimage and the reference's own heavy packages): seeded images, the values of its `ssim`, of LossMse's
arithmetic and of `compute_psnr`, and the gradients torch autograd gives through the reference's own functions."""
import importlib
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
R = "/root/reference/"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loss_fixtures.npz")


def main():
    jt = types.ModuleType("jaxtyping")

    class _T:
        def __class_getitem__(cls, item):
            return cls

    for n in ("Float", "Bool", "Int64", "Shaped", "Int", "UInt8"):
        setattr(jt, n, _T)
    sys.modules["jaxtyping"] = jt
    for name in ("lpips", "skimage", "skimage.metrics"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["lpips"].LPIPS = object
    sys.modules["skimage.metrics"].structural_similarity = None
    for name, path in [("src", "src"), ("src.loss", "src/loss"), ("src.evaluation", "src/evaluation"), ("src.dataset", "src/dataset"),
                       ("src.model", "src/model"), ("src.model.decoder", "src/model/decoder")]:
        m = types.ModuleType(name)
        m.__path__ = [R + path]
        sys.modules[name] = m
    for name, attrs in (("src.dataset.types", ("BatchedExample",)), ("src.model.decoder.decoder", ("DecoderOutput",)),
                        ("src.model.types", ("Gaussians",))):
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, object)
        sys.modules[name] = m
    lm = types.ModuleType("src.loss.loss")

    class Loss:
        def __class_getitem__(cls, item):
            return cls

    lm.Loss = Loss
    sys.modules["src.loss.loss"] = lm
    ms = importlib.import_module("src.loss.loss_multissim")
    mt = importlib.import_module("src.evaluation.metrics")
    g = torch.Generator().manual_seed(21)
    out = {}
    for tag, (n, h, w) in (("A", (2, 40, 56)), ("B", (1, 16, 16)), ("C", (3, 33, 21))):
        target = torch.rand((n, 3, h, w), generator=g)
        pred = (target + 0.15 * torch.randn((n, 3, h, w), generator=g)).requires_grad_(True)
        s = ms.ssim(pred, target)
        (gs,) = torch.autograd.grad(s, pred)
        delta = pred - target
        mse = (delta ** 2).mean()
        (gm,) = torch.autograd.grad(mse, pred)
        out[tag + "_pred"], out[tag + "_target"] = pred.detach().numpy(), target.numpy()
        out[tag + "_ssim"], out[tag + "_ssim_grad"] = s.detach().numpy(), gs.numpy()
        out[tag + "_mse"], out[tag + "_mse_grad"] = mse.detach().numpy(), gm.numpy()
        out[tag + "_psnr"] = mt.compute_psnr(target, pred.detach()).numpy()
    np.savez_compressed(OUT, **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
