"""GPU-resident two-view data path: replaces DoubleAugmentedDataset + PIL worker processes
(reference utils/data_utils.py:56-73,113-121).

The dataset lives in HBM as uint8 [N,H,W,3]; a "loader" iteration picks a permutation slice and
the fused augmentation kernels (utils/augmentations.py) turn it into the two normalised fp32
views, so there is no host->device copy of fp32 images and no CPU augmentation in the loop.
Batches keep the reference keys: index, img, aug_1, aug_2, label.

Datasets: CIFAR-10/100 are read from the standard python pickles under ``root`` if they are
already there (no download: the target boxes have no network); ``synthetic: {...}`` in the
``data`` block of the YAML generates a seeded uint8 dataset of the requested shape instead.

A dataset that does not fit the HBM budget (``data.max_resident_gb``, default: everything resident) is STREAMED: it stays in pinned host
memory as uint8, the epoch's permutation is cut into chunks of ``stream_chunk_batches`` steps, and a background thread gathers the next
chunk's images (this rank's rows only) into a pinned staging buffer and uploads them on a side stream into the other half of a
double-buffered device chunk while the current chunk trains.  Batches are bit-identical to the resident path (same permutation, same
augmentation streams keyed by dataset index).  JPEG decoding is not part of this path: the dataset is decoded uint8 (as CIFAR's pickles
are); an ImageNet-scale run feeds it a pre-decoded uint8 array.
"""
import os
import pickle
import threading

import numpy as np
import torch

from . import augmentations


def _load_cifar(root, name, train):
    if name == "cifar10":
        base = os.path.join(root, "cifar-10-batches-py")
        files = [f"data_batch_{i}" for i in range(1, 6)] if train else ["test_batch"]
        key = b"labels"
    else:
        base = os.path.join(root, "cifar-100-python")
        files = ["train"] if train else ["test"]
        key = b"fine_labels"
    if not os.path.isdir(base):
        raise FileNotFoundError(
            f"{base} not found. There is no network on this box, so datasets are not downloaded: place the "
            f"extracted CIFAR python archive there, or add a `synthetic:` block to the `data` section of the config")
    xs, ys = [], []
    for f in files:
        with open(os.path.join(base, f), "rb") as fh:
            d = pickle.load(fh, encoding="bytes")
        xs.append(np.asarray(d[b"data"], dtype=np.uint8).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1))
        ys.append(np.asarray(d[key], dtype=np.int64))
    return np.ascontiguousarray(np.concatenate(xs)), np.concatenate(ys)


def _synthetic(spec, train, seed=420):
    """`kind: noise` (default): i.i.d. uint8 pixels, random labels - for throughput runs and smoke tests.
    `kind: patterns`: every class is a smooth random colour field (a 4x4 grid of colours, bilinearly enlarged) seen under a random
    shift, a random contrast and pixel noise - a dataset on which the label is learnable, for end-to-end training checks."""
    n = int(spec.get("num_train", 4096) if train else spec.get("num_test", 1024))
    h, w = spec.get("image_size", [32, 32])
    classes = int(spec.get("num_classes", 10))
    rng = np.random.default_rng(seed + (0 if train else 1))
    if spec.get("kind", "noise") == "noise":
        return rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8), rng.integers(0, classes, size=n)
    if spec["kind"] != "patterns":
        raise ValueError(f"unknown synthetic kind {spec['kind']!r} (noise | patterns)")
    grids = np.random.default_rng(seed + 7).uniform(0.0, 1.0, size=(classes, 4, 4, 3))      # the same templates for train and test
    ys, xs = np.linspace(0, 3, 2 * h), np.linspace(0, 3, 2 * w)
    y0, x0 = np.floor(ys).astype(int).clip(0, 2), np.floor(xs).astype(int).clip(0, 2)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    big = np.stack([(g[y0][:, x0] * (1 - fy) * (1 - fx) + g[y0 + 1][:, x0] * fy * (1 - fx) + g[y0][:, x0 + 1] * (1 - fy) * fx + g[y0 + 1][:, x0 + 1] * fy * fx)
                    for g in grids])                                                        # [classes, 2h, 2w, 3]
    labels = rng.integers(0, classes, size=n)
    oy, ox = rng.integers(0, h, size=n), rng.integers(0, w, size=n)
    contrast = rng.uniform(0.6, 1.0, size=(n, 1, 1, 1))
    imgs = np.stack([big[c, a:a + h, b:b + w] for c, a, b in zip(labels, oy, ox)])
    imgs = 0.5 + (imgs - 0.5) * contrast + rng.normal(0.0, 0.08, size=imgs.shape)
    return np.ascontiguousarray((imgs.clip(0, 1) * 255).astype(np.uint8)), labels


DATASETS = ("cifar10", "cifar100")


def _gather_rows(host, ids, out):
    """out[i] = host[ids[i]] for uint8 image rows.  Viewed as int64 words when the row size allows it: index_select then moves 8 bytes
    per element instead of 1 (measured 3.4x faster on the host)."""
    n, m = host.shape[0], ids.numel()
    row = host[0].numel()
    if row % 8 == 0:
        torch.index_select(host.view(n, row).view(torch.int64), 0, ids, out=out[:m].view(m, row).view(torch.int64))
    else:
        torch.index_select(host, 0, ids, out=out[:m])


class GpuTwoViewLoader:
    """Iterates {index, img, aug_1, aug_2, label} batches produced on the GPU.

    Data parallel (one process per GPU): every rank draws the SAME permutation (shared seed), walks it in
    global batches of ``batch_size * world`` and takes rows ``[rank*B, (rank+1)*B)`` of each - the ranks' batches are disjoint
    and their union is the global batch.  Augmentation streams are keyed by (dataset index, view, global step), so the views
    of a sample do not depend on the world size.  The ragged last global batch is split evenly; the at most world-1 samples
    that do not divide are left out of that epoch (equal shard sizes are what the all-gather of the embeddings needs).
    ``eval_batches()`` is never sharded: BatchNorm runs on batch statistics in the reference's evaluation too, so every
    rank walks the same full batches and gets the same features as a single-GPU run."""

    def __init__(self, images_u8, labels, transforms, batch_size, shuffle, device, seed=420, rank=None, world=None,
                 max_resident_bytes=None, stream_chunk_batches=8):
        self.device = device
        host = images_u8 if isinstance(images_u8, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(images_u8))
        self.shape = tuple(host.shape)
        self.streamed = max_resident_bytes is not None and host.numel() > max_resident_bytes and torch.device(device).type == "cuda"
        if self.streamed:
            self.host = host.pin_memory()                                    # uint8 [N,H,W,3] in pinned host memory
            self.images = None
            self.chunk_batches = max(1, int(stream_chunk_batches))
            self._copy_stream = torch.cuda.Stream(device)
        else:
            self.host = None
            self.images = host.to(device)                                    # uint8 [N,H,W,3] resident in HBM
        self.labels = torch.from_numpy(np.asarray(labels, dtype=np.int64)).to(device)
        if torch.device(device).type == "cuda":
            from .. import nn as hnn
            hnn.data_ready(device)                                           # the uploads above: every later input_stream block is ordered behind them
        self.batch_size, self.shuffle = int(batch_size), shuffle
        self._rank, self._world = rank, world
        self._setup_transforms(transforms)
        self.gen = torch.Generator().manual_seed(seed)
        self.step = 0

    def _setup_transforms(self, transforms):
        self.train_tf = augmentations.get_transform(transforms["train"])
        self.test_tf = augmentations.get_transform(transforms["test"])

    # ---- sharding ------------------------------------------------------------------------------------------------
    def _shard(self):
        from .. import distributed as hdist
        rank = hdist.rank() if self._rank is None else self._rank
        world = hdist.world_size() if self._world is None else self._world
        return rank, world

    @property
    def num_classes(self):
        return int(self.labels.max().item()) + 1

    def _rank_slices(self, n):
        """[(start, stop)] into the epoch's permutation: this rank's rows of every global batch."""
        rank, world = self._shard()
        b = self.batch_size
        out = []
        for s in range(0, n, b * world):
            rows = min(b * world, n - s)
            per = b if rows == b * world else rows // world       # ragged tail: equal shards, remainder (< world samples) left out
            if per > 0:
                out.append((s + rank * per, s + (rank + 1) * per))
        return out

    def __len__(self):
        return len(self._rank_slices(self.shape[0]))                        # per-rank steps; world 1: the last batch is NOT dropped

    def _order(self):
        n = self.shape[0]
        return torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)

    def _make(self, idx, step, images=None, rows=None):
        """idx: dataset indices (they key the augmentation streams and pick the labels); images / rows: where the pixels are - the
        resident dataset indexed by idx, or a streamed chunk indexed by its local rows."""
        images = self.images if images is None else images
        rows = idx if rows is None else rows
        views = self.train_tf.apply(images, rows, self.train_tf.draw(images, idx, step, 2))
        img = self.test_tf.one_view(images, rows)
        return {"index": idx, "img": img, "aug_1": views[0], "aug_2": views[1], "label": self.labels[idx]}

    def __iter__(self):
        order = self._order()                                               # identical on every rank (shared seed)
        slices = self._rank_slices(order.numel())
        if self.streamed:
            yield from self._iter_streamed(order, slices)
            return
        from .. import nn as hnn
        for a, b in slices:
            with hnn.input_stream(self.device) as ins:                      # built beside the previous step's backward (resident data only)
                idx = order[a:b].to(self.device)
                batch = self._make(idx, self.step)                          # self.step is the GLOBAL step: same on every rank
                ins.publish(*batch.values())
            self.step += 1
            yield batch

    # ---- streaming ----------------------------------------------------------------------------------------------------
    def _iter_streamed(self, order, slices):
        """Double-buffered chunks: while chunk c trains out of one device buffer, a background thread gathers chunk c+1 (pinned
        staging) and uploads it into the other on the side stream.  Events order (i) upload -> first use, (ii) last use -> overwrite."""
        kb = self.chunk_batches
        chunks = [slices[i:i + kb] for i in range(0, len(slices), kb)]
        cap = max(sum(b - a for a, b in ch) for ch in chunks)
        if getattr(self, "_cap", 0) < cap:
            self._stage = [torch.empty((cap,) + self.shape[1:], dtype=torch.uint8).pin_memory() for _ in range(2)]
            self._dev = [torch.empty((cap,) + self.shape[1:], dtype=torch.uint8, device=self.device) for _ in range(2)]
            self._cap = cap
        uploaded = [torch.cuda.Event(), torch.cuda.Event()]
        released = [None, None]                                             # compute-stream events: the buffer's last reader has been enqueued

        failure = []

        def fetch(c, k):
            try:
                ids = torch.cat([order[a:b] for a, b in chunks[c]])
                _gather_rows(self.host, ids, self._stage[k])                                # host gather (releases the GIL)
                with torch.cuda.stream(self._copy_stream):
                    if released[k] is not None:
                        self._copy_stream.wait_event(released[k])
                    self._dev[k][:ids.numel()].copy_(self._stage[k][:ids.numel()], non_blocking=True)
                    uploaded[k].record(self._copy_stream)
            except BaseException as exc:                                    # surfaced on the training thread at the next join
                failure.append(exc)

        worker = threading.Thread(target=fetch, args=(0, 0))
        worker.start()
        for c, ch in enumerate(chunks):
            k = c & 1
            worker.join()
            if failure:
                raise failure[0]
            if c + 1 < len(chunks):
                uploaded[k ^ 1].synchronize() if c >= 1 else None           # the staging buffer's previous upload must have left the host
                worker = threading.Thread(target=fetch, args=(c + 1, k ^ 1))
                worker.start()
            torch.cuda.current_stream(self.device).wait_event(uploaded[k])
            r0 = 0
            for a, b in ch:
                idx = order[a:b].to(self.device)
                rows = torch.arange(r0, r0 + (b - a), device=self.device)
                batch = self._make(idx, self.step, images=self._dev[k], rows=rows)
                r0 += b - a
                self.step += 1
                yield batch
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            released[k] = ev

    def eval_batches(self):
        """{index, img, label} over the whole set in order, full batches on every rank (feature extraction / kNN / linear eval)."""
        n = self.shape[0]
        for s in range(0, n, self.batch_size):
            e = min(s + self.batch_size, n)
            idx = torch.arange(s, e, device=self.device)
            if self.streamed:
                chunk = self.host[s:e].to(self.device, non_blocking=True)
                img = self.test_tf.one_view(chunk, torch.arange(e - s, device=self.device))
            else:
                img = self.test_tf.one_view(self.images, idx)
            yield {"index": idx, "img": img, "label": self.labels[idx]}

    def num_eval_batches(self):
        return (self.shape[0] + self.batch_size - 1) // self.batch_size


def _load(dataset_name, root, synthetic):
    assert synthetic is not None or dataset_name in DATASETS, \
        f"Unrecognized dataset {dataset_name}, expected one of {list(DATASETS)}"
    if synthetic is not None:
        return _synthetic(synthetic, True), _synthetic(synthetic, False)
    return _load_cifar(root, dataset_name, True), _load_cifar(root, dataset_name, False)


def _stream_kwargs(max_resident_gb, stream_chunk_batches):
    return dict(max_resident_bytes=None if max_resident_gb is None else int(float(max_resident_gb) * (1 << 30)), stream_chunk_batches=stream_chunk_batches)


def get_double_augment_dataloaders(dataset_name, root, transforms, batch_size, device=None, synthetic=None, max_resident_gb=None, stream_chunk_batches=8):
    (xtr, ytr), (xte, yte) = _load(dataset_name, root, synthetic)
    kw = _stream_kwargs(max_resident_gb, stream_chunk_batches)
    train_loader = GpuTwoViewLoader(xtr, ytr, transforms, batch_size, True, device, **kw)
    test_loader = GpuTwoViewLoader(xte, yte, transforms, batch_size, False, device, **kw)
    return train_loader, test_loader


class GpuMultiCropLoader(GpuTwoViewLoader):
    """Iterates {img, global_1, global_2, local_1, local_2, label} batches (reference MultiCropDataset, utils/data_utils.py:76-92);
    sharded over ranks like GpuTwoViewLoader."""

    def _setup_transforms(self, multicrop_config):
        self.multi_crop = augmentations.MultiCrop(multicrop_config)
        self.test_tf = augmentations.get_transform(multicrop_config["test_transforms"])

    def _make(self, idx, step, images=None, rows=None):
        images = self.images if images is None else images
        rows = idx if rows is None else rows
        batch = self.multi_crop(images, rows, step, sample_ids=idx)         # random streams keyed by the dataset index
        batch.update(index=idx, img=self.test_tf.one_view(images, rows), label=self.labels[idx])
        return batch


def get_multicrop_dataloaders(dataset_name, root, multicrop_config, batch_size, device=None, synthetic=None, max_resident_gb=None, stream_chunk_batches=8):
    (xtr, ytr), (xte, yte) = _load(dataset_name, root, synthetic)
    kw = _stream_kwargs(max_resident_gb, stream_chunk_batches)
    return (GpuMultiCropLoader(xtr, ytr, multicrop_config, batch_size, True, device, **kw),
            GpuMultiCropLoader(xte, yte, multicrop_config, batch_size, False, device, **kw))
