"""GPU augmentation front-end: YAML transform block -> fused HIP pipeline.

Replaces get_transform / TRANSFORM_HELPER of the reference (utils/augmentations.py:113-144), which builds a
torchvision Compose that runs on PIL images in DataLoader workers.  Here the same YAML block (key order =
transform order) is compiled into the parameters of three kernels (csrc/augment.hip) that read the uint8
dataset resident in HBM and write the normalised fp32 views directly: no CPU work, no H2D copy per step.

Supported chains (everything the shipped simclr / byol / barlow configs use):
  train: [color_jitter(apply_prob)] [random_gray] random_resized_crop [random_flip] to_tensor normalize
  test : center_crop to_tensor normalize
Transforms registered by the reference but used by none of its configs on this path (gaussian_blur, cutout,
rand_aug, random_crop, resize) raise NotImplementedError.
"""
import ctypes as C

import torch

from .. import _lib

_TRAIN_ORDER = ["color_jitter", "random_gray", "random_resized_crop", "random_flip", "to_tensor", "normalize"]
_TEST_ORDER = ["center_crop", "to_tensor", "normalize"]
_KNOWN = {"gaussian_blur", "color_jitter", "random_gray", "random_crop", "random_resized_crop", "center_crop", "resize",
          "random_flip", "to_tensor", "normalize", "rand_aug", "cutout"}


def _f3(vals):
    return (C.c_float * 3)(*[float(v) for v in vals])


class GpuTransform:
    def __init__(self, config, seed=420):
        keys = list(config.keys())
        for k in keys:
            if k not in _KNOWN:
                raise KeyError(k)
        self.kind = "test" if "center_crop" in keys else "train"
        order = _TEST_ORDER if self.kind == "test" else _TRAIN_ORDER
        if [k for k in order if k in keys] != keys or "to_tensor" not in keys or "normalize" not in keys:
            raise NotImplementedError(f"transform chain {keys} is not one the fused GPU pipeline implements "
                                      f"(supported order: {order}, optional entries may be dropped)")
        norm = config["normalize"]
        self.mean, self.std = _f3(norm["mean"]), _f3(norm["std"])
        self.seed = seed
        if self.kind == "test":
            self.size = tuple(config["center_crop"]["size"])
            return
        if "random_resized_crop" not in keys:
            raise NotImplementedError("the train chain needs random_resized_crop")
        cj = config.get("color_jitter") or {}
        rrc = config["random_resized_crop"]
        self.size = tuple(rrc["size"])
        scale, ratio = rrc.get("scale", (0.08, 1.0)), rrc.get("ratio", (3.0 / 4.0, 4.0 / 3.0))
        if rrc.get("interpolation", "bilinear") not in ("bilinear", 2):
            raise NotImplementedError("only bilinear resampling is implemented")
        has_cj = "color_jitter" in keys
        self.cfg = _lib.AugCfg(
            float(cj.get("brightness", 0.0)), float(cj.get("contrast", 0.0)), float(cj.get("saturation", 0.0)), float(cj.get("hue", 0.0)),
            (float(cj.get("apply_prob", 1.0)) if has_cj else 0.0),
            float((config.get("random_gray") or {}).get("p", 0.1)) if "random_gray" in keys else 0.0,
            float((config.get("random_flip") or {}).get("p", 0.5)) if "random_flip" in keys else 0.0,
            float(scale[0]), float(scale[1]), float(ratio[0]), float(ratio[1]))

    # ------------------------------------------------------------------------------------------
    def draw(self, images, idx, step, nviews=2):
        """Per-(view, sample) parameter records [nviews, B, 16] from the Philox stream keyed by dataset index."""
        b = idx.numel()
        _, hs, ws, _ = images.shape
        params = torch.empty((nviews, b, 16), dtype=torch.float32, device=images.device)
        _lib.call("ssv_augment_params", b, hs, ws, nviews, C.byref(self.cfg), self.seed, int(step), _lib.ptr(idx), 0,
                  _lib.ptr(params), _lib.stream())
        return params

    def apply(self, images, idx, params):
        """images: uint8 [N,Hs,Ws,3] on the GPU; idx: int64 [B]; params [V,B,16] -> fp32 [V,B,3,Ho,Wo] (channels_last memory)."""
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3 or not images.is_contiguous():
            raise _lib.SsvError("augmentation source must be a contiguous uint8 [N,H,W,3] device tensor")
        _lib._dev(images, idx, params)
        v, b = params.shape[0], params.shape[1]
        _, hs, ws, _ = images.shape
        ho, wo = self.size
        out = torch.empty((v, b, ho, wo, 3), dtype=torch.float32, device=images.device)
        ws_t = _lib.workspace.get(_lib.load().ssv_augment_workspace_bytes(b, v, ho, wo), images.device)
        _lib.call("ssv_augment_views", b, v, hs, ws, ho, wo, _lib.ptr(images), _lib.ptr(idx), _lib.ptr(params.contiguous()),
                  self.mean, self.std, _lib.ptr(out), _lib.ptr(ws_t), ws_t.numel(), _lib.stream())
        return out.permute(0, 1, 4, 2, 3)

    def two_views(self, images, idx, step):
        out = self.apply(images, idx, self.draw(images, idx, step, 2))
        return out[0], out[1]

    def one_view(self, images, idx):
        if self.kind != "test":
            raise _lib.SsvError("one_view needs a center_crop chain")
        _lib._dev(images, idx)
        b = idx.numel()
        _, hs, ws, _ = images.shape
        ho, wo = self.size
        out = torch.empty((b, ho, wo, 3), dtype=torch.float32, device=images.device)
        _lib.call("ssv_center_view", b, hs, ws, ho, wo, _lib.ptr(images), _lib.ptr(idx), self.mean, self.std, _lib.ptr(out), _lib.stream())
        return out.permute(0, 3, 1, 2)


def get_transform(config):
    """YAML transform block -> GpuTransform (same entry point name as the reference)."""
    return GpuTransform(config)


class MultiCrop:
    """GPU form of the reference's MultiCrop (utils/augmentations.py:156-173): the two-view chain produces two augmented,
    already normalised copies of every image; each copy then yields `num_global_views` crops (area scale (threshold, 1)) and
    `num_local_views` crops (scale (0.08, threshold)), all resized bicubically on the float tensors.  Returns the batch dict
    entries global_1, global_2 [B, Vg, 3, Hg, Wg] and local_1, local_2 [B, Vl, 3, Hl, Wl]."""

    def __init__(self, config, seed=420):
        self.num_local = int(config.get("num_local_views", 6))
        self.num_global = int(config.get("num_global_views", 2))
        self.scale = float(config.get("scale_threshold", 0.3))
        self.global_size, self.local_size = tuple(config["global_size"]), tuple(config["local_size"])
        self.transforms = get_transform(config["train_transforms"])
        self.seed = seed

    def __call__(self, images, idx, step, sample_ids=None):
        """idx: rows of `images`; sample_ids (default idx): the GLOBAL sample ids that key the random streams."""
        from .. import ops
        ids = idx if sample_ids is None else sample_ids
        views = self.transforms.apply(images, idx, self.transforms.draw(images, ids, step, 2))     # [2, B, 3, H, W], NHWC memory
        hs, ws = self.transforms.size
        out = {}
        for copy in range(2):
            nhwc = views[copy].permute(0, 2, 3, 1)
            for name, ncrop, scale, size, base in (("global", self.num_global, (self.scale, 1.0), self.global_size, 16),
                                                   ("local", self.num_local, (0.08, self.scale), self.local_size, 144)):
                boxes = ops.multicrop_params(idx.numel(), hs, ws, ncrop, base + 256 * copy, scale, self.seed, step, sample_ids=ids)
                out[f"{name}_{copy + 1}"] = ops.multicrop(nhwc, boxes, size).permute(0, 1, 4, 2, 3)
        return out
