"""GPU: the fused augmentation kernels against the Pillow-pinned oracle (bit-exact pixels, identical
parameter stream) and through the loader that replaces DoubleAugmentedDataset."""
import numpy as np
import pytest
import torch

from oracle import augment as A

pytestmark = pytest.mark.gpu
MEAN, STD = [0.4914, 0.4822, 0.4465], [0.2470, 0.2435, 0.2616]
TRAIN = {"color_jitter": {"brightness": 0.4, "contrast": 0.4, "saturation": 0.4, "hue": 0.1, "apply_prob": 0.8},
         "random_gray": {"p": 0.2}, "random_resized_crop": {"size": [32, 32], "scale": [0.2, 1.0]},
         "random_flip": None, "to_tensor": None, "normalize": {"mean": MEAN, "std": STD}}
TEST = {"center_crop": {"size": [32, 32]}, "to_tensor": None, "normalize": {"mean": MEAN, "std": STD}}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _cfg(size):
    c = {k: (dict(v) if isinstance(v, dict) else v) for k, v in TRAIN.items()}
    c["random_resized_crop"] = {"size": list(size), "scale": [0.2, 1.0]}
    return c


@pytest.mark.parametrize("src_hw,out_hw,n", [((32, 32), (32, 32), 48), ((40, 56), (24, 24), 24), ((72, 64), (56, 56), 12), ((256, 256), (224, 224), 3)])
def test_views_bit_exact_vs_pillow_oracle(dev, src_hw, out_hw, n):
    from ssv_amd.utils import augmentations
    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, (n + 5, src_hw[0], src_hw[1], 3), dtype=np.uint8)
    imgs[1] = imgs[1][:, :, :1]                                          # a grey image
    imgs[2] = (imgs[2] // 85) * 85                                       # few levels: saturated blends
    idx = torch.tensor(rng.permutation(n + 5)[:n].astype(np.int64))
    tf = augmentations.get_transform(_cfg(out_hw))
    d_imgs, d_idx = torch.from_numpy(imgs).to(dev), idx.to(dev)
    params = tf.draw(d_imgs, d_idx, step=7)
    # 1. identical parameter stream (Philox restatement on the CPU)
    ref_p = np.stack([[A.draw_params(420, 7, int(i), v, src_hw[0], src_hw[1]) for i in idx] for v in range(2)])
    np.testing.assert_array_equal(params.cpu().numpy(), ref_p)
    # 2. force the rare branches on a few records, then compare pixels bit for bit
    p = params.cpu().numpy().copy()
    p[0, 0, 0], p[0, 0, 1:5] = 1, [3, 1, 0, 2]
    p[1, 0, 0], p[1, 0, 1:5], p[1, 0, 9] = 1, [1, 2, 3, 0], 1
    p[0, 1, 0], p[0, 1, 14] = 0, 1
    p[1, 1, 10:14] = [0, 0, src_hw[0], src_hw[1]]                        # whole image: the down-scaling path
    out = tf.apply(d_imgs, d_idx, torch.from_numpy(p).to(dev)).cpu().numpy()
    assert out.shape == (2, n, 3, out_hw[0], out_hw[1])
    for v in range(2):
        for k in range(n):
            ref = A.view_numpy(imgs[int(idx[k])], p[v, k], out_hw, MEAN, STD)
            np.testing.assert_array_equal(out[v, k], ref, err_msg=f"view {v} sample {k} params {p[v, k]}")
    ref_pil = A.view_pil(imgs[int(idx[0])], p[0, 0], out_hw, MEAN, STD)   # and directly against Pillow
    np.testing.assert_array_equal(out[0, 0], ref_pil)


def test_center_view_and_loader(dev):
    from ssv_amd.utils import augmentations, data_utils
    rng = np.random.default_rng(6)
    imgs = rng.integers(0, 256, (10, 37, 41, 3), dtype=np.uint8)
    te = augmentations.get_transform(TEST)
    idx = torch.arange(10, device=dev)
    got = te.one_view(torch.from_numpy(imgs).to(dev), idx).cpu().numpy()
    for k in range(10):
        np.testing.assert_array_equal(got[k], A.center_view_pil(imgs[k], (32, 32), MEAN, STD))
    # loader: reference batch keys, last batch kept, views differ, channels_last memory (NHWC) behind an NCHW shape
    labels = np.arange(10) % 3
    loader = data_utils.GpuTwoViewLoader(imgs, labels, {"train": _cfg((32, 32)), "test": TEST}, batch_size=4, shuffle=True, device=dev)
    assert len(loader) == 3
    seen = []
    for batch in loader:
        assert set(batch) == {"index", "img", "aug_1", "aug_2", "label"}
        b = batch["index"].numel()
        assert batch["aug_1"].shape == (b, 3, 32, 32) and batch["aug_1"].is_contiguous(memory_format=torch.channels_last)
        assert not torch.equal(batch["aug_1"], batch["aug_2"])
        assert torch.equal(batch["label"].cpu(), torch.from_numpy(labels)[batch["index"].cpu()])
        seen += batch["index"].cpu().tolist()
    assert sorted(seen) == list(range(10))


def test_streamed_loader_yields_the_resident_loaders_batches(dev):
    """A dataset above `max_resident_bytes` stays in pinned host memory; chunks of it are gathered + uploaded by a background thread
    into a double-buffered device chunk.  Every batch of two epochs (ragged last chunk and last batch included) equals the resident
    loader's bit for bit - indices, labels, both views and the centre view."""
    from ssv_amd.utils import data_utils
    rng = np.random.default_rng(3)
    imgs, labels = rng.integers(0, 256, size=(150, 40, 40, 3), dtype=np.uint8), rng.integers(0, 10, size=150)
    tfs = {"train": _cfg((32, 32)), "test": TEST}
    res = data_utils.GpuTwoViewLoader(imgs, labels, tfs, batch_size=16, shuffle=True, device=dev)
    stm = data_utils.GpuTwoViewLoader(imgs, labels, tfs, batch_size=16, shuffle=True, device=dev, max_resident_bytes=1000, stream_chunk_batches=3)
    assert stm.streamed and not res.streamed and stm.images is None and len(stm) == len(res) == 10
    for epoch in range(2):
        n = 0
        for a, b in zip(res, stm):
            for k in ("index", "label", "aug_1", "aug_2", "img"):
                assert torch.equal(a[k], b[k]), (epoch, n, k)
            n += 1
        assert n == 10
    ea, eb = list(res.eval_batches()), list(stm.eval_batches())
    assert len(ea) == len(eb) == 10 and all(torch.equal(x["img"], y["img"]) and torch.equal(x["label"], y["label"]) for x, y in zip(ea, eb))
