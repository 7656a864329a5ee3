#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running THE REFERENCE ITSELF.

Run only in the build container, where /root/reference exists (read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

The reference's Python never travels: only the numbers written here (small .npz files) are
committed.  wandb / faiss / torchvision are absent, so they are registered as empty stub
modules before importing ``models/*``; trainers are built with ``object.__new__`` and wired
exactly as their ``__init__`` does (models/simclr.py:50-58, models/byol.py:75-89,
models/barlow.py:50-58), skipping the dataloaders / wandb.  Inputs are seeded torch CPU tensors
and are regenerated (not stored) by the tests from the seeds recorded in each file.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("SSV_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _stub_modules():
    for name in ("wandb", "faiss"):
        sys.modules[name] = types.ModuleType(name)
    tv = types.ModuleType("torchvision")
    tv.datasets = types.ModuleType("torchvision.datasets")
    tv.datasets.CIFAR10 = tv.datasets.CIFAR100 = object
    tv.transforms = types.ModuleType("torchvision.transforms")
    for n in ("ColorJitter", "RandomGrayscale", "RandomCrop", "RandomResizedCrop", "CenterCrop", "Resize",
              "RandomHorizontalFlip", "ToTensor", "Normalize", "RandomApply", "Compose"):
        setattr(tv.transforms, n, object)
    tv.transforms.functional = types.ModuleType("torchvision.transforms.functional")
    tv.transforms.functional.InterpolationMode = types.SimpleNamespace(BICUBIC="bicubic")
    sys.modules.update({"torchvision": tv, "torchvision.datasets": tv.datasets, "torchvision.transforms": tv.transforms,
                        "torchvision.transforms.functional": tv.transforms.functional})


def checksum(t):
    d = t.detach().double().flatten()
    head = d[:4].tolist() + [0.0] * (4 - min(4, d.numel()))
    return [float(d.sum()), float((d * d).sum())] + head


def randn(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def state_checksums(sd):
    keys = [k for k, v in sd.items() if v.dtype.is_floating_point]
    return np.array(keys), np.array([checksum(sd[k]) for k in keys])


def main():
    sys.path.insert(0, REF)
    _stub_modules()
    from networks import resnet as ref_resnet
    from utils import losses as ref_losses, train_utils as ref_tu
    from models import simclr as ref_simclr, byol as ref_byol, barlow as ref_barlow

    # ------------------------------------------------------------------ 1. loss level
    out = {}
    cases = [("a", 64, 128, 11, True, 0.5), ("b", 64, 128, 12, False, 1.0), ("c", 64, 128, 13, True, 0.07),
             ("d", 16, 32, 14, True, 0.5), ("e", 5, 20, 15, False, 0.3)]
    out["ntxent_cases"] = np.array([[n, d, seed, int(norm), temp] for _, n, d, seed, norm, temp in cases], dtype=np.float64)
    for tag, n, d, seed, norm, temp in cases:
        zi, zj = randn(seed, n, d).requires_grad_(), randn(seed + 1000, n, d).requires_grad_()
        loss = ref_losses.SimclrLoss(normalize=norm, temperature=temp)(zi, zj)
        loss.backward()
        out[f"ntxent_{tag}_loss"] = np.float64(loss.item())
        out[f"ntxent_{tag}_dzi"], out[f"ntxent_{tag}_dzj"] = zi.grad.numpy(), zj.grad.numpy()
    bcases = [("a", 32, 64, 21, False, 0.005), ("b", 16, 256, 22, True, 0.005), ("c", 48, 32, 23, False, 0.05)]
    out["barlow_cases"] = np.array([[b, d, seed, int(norm), lm] for _, b, d, seed, norm, lm in bcases], dtype=np.float64)
    for tag, b, d, seed, norm, lm in bcases:
        zi, zj = randn(seed, b, d).requires_grad_(), randn(seed + 1000, b, d).requires_grad_()
        loss = ref_losses.BarlowLoss(normalize=norm, off_diagonal_weight=lm)(zi, zj)
        loss.backward()
        out[f"barlow_{tag}_loss"] = np.float64(loss.item())
        out[f"barlow_{tag}_dzi"], out[f"barlow_{tag}_dzj"] = zi.grad.numpy(), zj.grad.numpy()
    # BYOL loss: nn.MSELoss on L2-normalised vectors (models/byol.py:47,59,89,129-130)
    import torch.nn.functional as F
    p1, p2 = randn(31, 16, 128).requires_grad_(), randn(32, 16, 128).requires_grad_()
    t1, t2 = F.normalize(randn(33, 16, 128), dim=-1, p=2), F.normalize(randn(34, 16, 128), dim=-1, p=2)
    mse = torch.nn.MSELoss()
    loss = mse(F.normalize(p1, dim=-1, p=2), t2) + mse(F.normalize(p2, dim=-1, p=2), t1)
    loss.backward()
    out["byol_loss"] = np.float64(loss.item())
    out["byol_dp1"], out["byol_dp2"] = p1.grad.numpy(), p2.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "loss_level.npz"), **out)

    # ------------------------------------------------------------------ 2. init checksums
    out = {}
    for tag, fn, kw in (("r18rbc", ref_resnet.resnet18, dict(reduce_bottom_conv=True)),
                        ("r50", ref_resnet.resnet50, {}), ("r50rbc", ref_resnet.resnet50, dict(reduce_bottom_conv=True))):
        torch.manual_seed(420)
        enc = fn(**kw)
        head = ref_simclr.ProjectionHead(512 if "18" in tag else 2048, 128)
        out[f"{tag}_enc_keys"], out[f"{tag}_enc_sums"] = state_checksums(enc.state_dict())
        out[f"{tag}_head_keys"], out[f"{tag}_head_sums"] = state_checksums(head.state_dict())
        out[f"{tag}_all_keys"] = np.array(list(enc.state_dict().keys()))
        out[f"{tag}_nparams"] = np.int64(sum(p.numel() for p in enc.parameters()) + sum(p.numel() for p in head.parameters()))
    np.savez_compressed(os.path.join(OUT, "init_checksums.npz"), **out)

    # ------------------------------------------------------------------ 3. block level (Bottleneck w/ downsample, BasicBlock)
    out = {}
    torch.manual_seed(7)
    ds = torch.nn.Sequential(ref_resnet.conv1x1(64, 256, 2), torch.nn.BatchNorm2d(256))
    blk = ref_resnet.Bottleneck(64, 64, stride=2, downsample=ds)
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(m.weight, 0.5, 1.5)
            torch.nn.init.uniform_(m.bias, -0.5, 0.5)
    x = randn(71, 2, 64, 8, 8).requires_grad_()
    y = blk(x)
    dy = randn(72, *y.shape)
    y.backward(dy)
    for k, v in blk.state_dict().items():
        out[f"bottleneck_sd_{k}"] = v.numpy()
    out["bottleneck_y"], out["bottleneck_dx"] = y.detach().numpy(), x.grad.numpy()
    for k, p in blk.named_parameters():
        out[f"bottleneck_grad_{k}"] = p.grad.numpy()
    torch.manual_seed(8)
    bb = ref_resnet.BasicBlock(32, 32)
    x = randn(81, 2, 32, 6, 6).requires_grad_()
    y = bb(x)
    dy = randn(82, *y.shape)
    y.backward(dy)
    for k, v in bb.state_dict().items():
        out[f"basic_sd_{k}"] = v.numpy()
    out["basic_y"], out["basic_dx"] = y.detach().numpy(), x.grad.numpy()
    for k, p in bb.named_parameters():
        out[f"basic_grad_{k}"] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "block_level.npz"), **out)

    # ------------------------------------------------------------------ 4. step level
    def build_simclr(arch, enc_kw, proj_dim, opt_cfg, sched_cfg, loss_cfg, cls=ref_simclr.SimCLR, mod=ref_simclr, loss_cls=None):
        torch.manual_seed(420)
        m = object.__new__(cls)
        m.device = torch.device("cpu")
        fn, dim = mod.NETWORKS[arch].values()
        m.encoder = fn(**enc_kw)
        m.proj_head = mod.ProjectionHead(dim, proj_dim)
        m.optim = ref_tu.get_optimizer(opt_cfg, params=list(m.encoder.parameters()) + list(m.proj_head.parameters()))
        m.scheduler, m.warmup_epochs = ref_tu.get_scheduler({**sched_cfg, "epochs": 1000}, optimizer=m.optim)
        m.loss_fn = loss_cls(**loss_cfg)
        return m

    def run_steps(m, shape, nsteps, seed0, tag, out, store_z=True):
        losses = []
        for s in range(nsteps):
            batch = {"aug_1": randn(seed0 + 2 * s, *shape), "aug_2": randn(seed0 + 2 * s + 1, *shape)}
            if s == 0 and store_z:
                # z of step 0: recomputing outside train_step would double-update BN stats; capture at the loss call
                captured = []
                orig = m.loss_fn.forward
                m.loss_fn.forward = lambda a, b, _o=orig: (captured.append((a.detach().clone(), b.detach().clone())), _o(a, b))[1]
                losses.append(m.train_step(batch)["loss"])
                m.loss_fn.forward = orig
                out[f"{tag}_z1"], out[f"{tag}_z2"] = captured[0][0].numpy(), captured[0][1].numpy()
            else:
                losses.append(m.train_step(batch)["loss"])
            if s == 0:
                sd = {**{"encoder." + k: v for k, v in m.encoder.state_dict().items()},
                      **{"proj_head." + k: v for k, v in m.proj_head.state_dict().items()}}
                out[f"{tag}_after1_keys"], out[f"{tag}_after1_sums"] = state_checksums(sd)
        out[f"{tag}_losses"] = np.array(losses, dtype=np.float64)
        out[f"{tag}_lr"] = np.float64(m.optim.param_groups[0]["lr"])

    out = {}
    sgd = {"name": "sgd", "lr": 2.0, "momentum": 0.9, "nesterov": True, "weight_decay": 1.0e-4}
    sched = {"name": "cosine", "warmup_epochs": 10}
    # config 1: SimCLR resnet18 rbc 32x32 bs=64, configs/simclr.yaml hyper-parameters (lr seeded to 0.2)
    m = build_simclr("resnet18", dict(reduce_bottom_conv=True), 128, sgd, sched, dict(normalize=True, temperature=0.5),
                     loss_cls=ref_losses.SimclrLoss)
    run_steps(m, (64, 3, 32, 32), 3, 100, "simclr_r18", out)
    # cheap proxy of config 2: resnet50 std stem, 64x64, bs=8
    m = build_simclr("resnet50", {}, 128, sgd, sched, dict(normalize=True, temperature=0.5), loss_cls=ref_losses.SimclrLoss)
    run_steps(m, (8, 3, 64, 64), 3, 200, "simclr_r50", out)
    # Barlow r18, proj 256, bs 32 (configs/barlow.yaml hyper-parameters except proj_dim)
    bsgd = {"name": "sgd", "lr": 0.2, "momentum": 0.9, "nesterov": True, "weight_decay": 1.5e-6}
    m = build_simclr("resnet18", dict(reduce_bottom_conv=True), 256, bsgd, sched, dict(normalize=False, off_diagonal_weight=0.005),
                     cls=ref_barlow.BarlowTwins, mod=ref_barlow, loss_cls=ref_losses.BarlowLoss)
    run_steps(m, (32, 3, 32, 32), 2, 300, "barlow_r18", out)
    # features (build_features body, models/simclr.py:109-111) on a fresh SimCLR r18
    m = build_simclr("resnet18", dict(reduce_bottom_conv=True), 128, sgd, sched, dict(normalize=True, temperature=0.5),
                     loss_cls=ref_losses.SimclrLoss)
    with torch.no_grad():
        img = randn(400, 16, 3, 32, 32)
        out["features_r18"] = F.normalize(m.proj_head(m.encoder(img)), dim=-1, p=2).numpy()
    # BYOL r18 bs 16, 2 steps with the train-loop hooks update_tau(step) + momentum_update()
    torch.manual_seed(420)
    b = object.__new__(ref_byol.BYOL)
    b.device = torch.device("cpu")
    b.config = {}
    fn, dim = ref_byol.NETWORKS["resnet18"].values()
    b.online_network = ref_byol.OnlineNetwork(fn(reduce_bottom_conv=True), dim, 128)
    b.target_network = ref_byol.TargetNetwork(fn(reduce_bottom_conv=True), dim, 128)
    b.max_steps, b.tau = 1000, 0.996
    for p in b.target_network.parameters():
        p.requires_grad = False
    ysgd = {"name": "sgd", "lr": 0.2, "momentum": 0.9, "nesterov": True, "weight_decay": 1.0e-4}
    b.optim = ref_tu.get_optimizer(ysgd, params=b.online_network.parameters())
    b.scheduler, b.warmup_epochs = ref_tu.get_scheduler({**sched, "epochs": 1000}, optimizer=b.optim)
    b.loss_fn = torch.nn.MSELoss()
    losses, taus = [], []
    for s in range(2):
        batch = {"aug_1": randn(500 + 2 * s, 16, 3, 32, 32), "aug_2": randn(501 + 2 * s, 16, 3, 32, 32)}
        losses.append(b.train_step(batch)["loss"])
        b.update_tau(s)
        b.momentum_update()
        taus.append(b.tau)
    out["byol_r18_losses"], out["byol_r18_taus"] = np.array(losses), np.array(taus)
    sd = {**{"online_network." + k: v for k, v in b.online_network.state_dict().items()},
          **{"target_network." + k: v for k, v in b.target_network.state_dict().items()}}
    out["byol_r18_after2_keys"], out["byol_r18_after2_sums"] = state_checksums(sd)
    out["byol_r18_lr"] = np.float64(b.optim.param_groups[0]["lr"])
    np.savez_compressed(os.path.join(OUT, "step_level.npz"), **out)

    # ------------------------------------------------------------------ 5. optimizer / schedule
    out = {}
    torch.manual_seed(5)
    ps = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(33))]
    opt = ref_tu.get_optimizer({"name": "sgd", "lr": 0.3, "weight_decay": 1e-2}, ps)
    out["sgd_p0_init"], out["sgd_p1_init"] = ps[0].detach().numpy().copy(), ps[1].detach().numpy().copy()
    for s in range(3):
        for i, p in enumerate(ps):
            p.grad = randn(50 + 10 * s + i, *p.shape)
        opt.step()
        out[f"sgd_p0_step{s}"], out[f"sgd_p1_step{s}"] = ps[0].detach().numpy().copy(), ps[1].detach().numpy().copy()
    # lr schedule as driven by SimCLR.adjust_learning_rate (models/simclr.py:77-84) for 30 epochs of 40
    m = object.__new__(ref_simclr.SimCLR)
    lin = torch.nn.Linear(2, 2)
    m.optim = ref_tu.get_optimizer({"name": "sgd", "lr": 2.0, "weight_decay": 0.0}, lin.parameters())
    m.scheduler, m.warmup_epochs = ref_tu.get_scheduler({"name": "cosine", "warmup_epochs": 10, "epochs": 40}, optimizer=m.optim)
    m.warmup_rate = (2.0 - 1e-12) / m.warmup_epochs
    lrs = [m.optim.param_groups[0]["lr"]]
    for epoch in range(1, 31):
        m.adjust_learning_rate(epoch)
        lrs.append(m.optim.param_groups[0]["lr"])
    out["lr_schedule"] = np.array(lrs)
    np.savez_compressed(os.path.join(OUT, "optim_level.npz"), **out)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
