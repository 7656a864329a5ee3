"""GPU: SimSiam / ReLIC / MoCo on the HIP path against the reference's fixtures (tests/golden/sibling_level.npz) and the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from oracle import siblings as sib
from conftest import seeded_randn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _bare(cls, dev, config, loader_len=1):
    from ssv_amd.utils import train_utils
    t = object.__new__(cls)
    t.config, t.device, t.train_loader = config, dev, [None] * loader_len
    torch.manual_seed(420)
    t._build("resnet18")
    t.scheduler, t.warmup_epochs = train_utils.get_scheduler({**config["scheduler"], "epochs": config["epochs"]}, optimizer=t.optim)
    return t


BASE = {"epochs": 1000, "encoder": {"reduce_bottom_conv": True}, "scheduler": {"name": "cosine", "warmup_epochs": 10}}


def test_sibling_loss_kernels_match_reference(dev, golden):
    from ssv_amd.utils import losses
    g = golden["sibling_level"]
    o, t = F.normalize(seeded_randn(1001, 12, 64), dim=1), F.normalize(seeded_randn(1002, 12, 64), dim=1)
    od = o.to(dev).requires_grad_()
    l = losses.simsiam_pair_loss(od, od.detach().clone().requires_grad_(), t.to(dev), t.to(dev))          # both pairs identical -> 0.5 * 2 * loss
    l.backward()
    np.testing.assert_allclose(l.item(), g["simsiam_loss"], rtol=1e-5)
    np.testing.assert_allclose(od.grad.cpu().numpy(), 0.5 * g["simsiam_do"], rtol=1e-5, atol=1e-9)
    for tag, norm, temp, alpha in (("a", True, 1.0, 0.5), ("b", False, 0.5, 0.3)):
        zi, zj, zo = (seeded_randn(1010 + i, 10, 32).to(dev).requires_grad_() for i in range(3))
        l = losses.RelicLoss(norm, temp, alpha)(zi, zj, zo)
        l.backward()
        np.testing.assert_allclose(l.item(), g[f"relic_{tag}_loss"], rtol=1e-5)
        for name, t_ in (("dzi", zi), ("dzj", zj), ("dzo", zo)):
            np.testing.assert_allclose(t_.grad.cpu().numpy(), g[f"relic_{tag}_{name}"], rtol=1e-3, atol=2e-7, err_msg=f"{tag} {name}")
    q, k, bank = seeded_randn(1020, 9, 32).to(dev).requires_grad_(), seeded_randn(1021, 9, 32).to(dev), F.normalize(seeded_randn(1022, 50, 32), dim=1)
    bank[40:] = 0.0
    padded = torch.zeros(64, 32)
    padded[:50] = bank                                                                     # queue of 50 in a 64-row buffer
    l = losses.MocoLoss(True, 0.07)(q, k, padded.to(dev), 50)
    l.backward()
    np.testing.assert_allclose(l.item(), g["moco_loss"], rtol=1e-5)
    np.testing.assert_allclose(q.grad.cpu().numpy(), g["moco_dq"], rtol=1e-3, atol=2e-7)


def test_queue_push_wraps_like_the_reference_loop(dev):
    from ssv_amd import ops
    bank, ref, ptr, rp = torch.zeros(48, 8, device=dev), torch.zeros(40, 8), 0, 0
    for step, n in enumerate((16, 16, 16, 7, 90)):
        keys = seeded_randn(50 + step, n, 8)
        if step == 3:
            keys[2] = 0.0                                                                  # a zero key stays zero (eps clamp)
        ptr = ops.queue_push(bank, 40, ptr, keys.to(dev))
        for row in keys:
            ref[rp] = F.normalize(row, dim=-1, p=2)
            rp = (rp + 1) % 40
        assert ptr == rp
        np.testing.assert_allclose(bank[:40].cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)
    assert float(bank[40:].abs().max()) == 0.0


def test_simsiam_steps_match_reference(dev, golden):
    from ssv_amd.models.simsiam import SimSiam
    g = golden["sibling_level"]
    t = _bare(SimSiam, dev, {**BASE, "proj_dim": 256, "bottleneck_dim": 64, "optimizer": {"name": "sgd", "lr": 0.05, "weight_decay": 1e-4}})
    sd = t.online_network.state_dict()
    for k, ref in zip(g["simsiam_init_keys"], g["simsiam_init_sums"]):
        k = str(k)[len("online_network."):]
        np.testing.assert_allclose(np.array(oracle.tensor_checksum(sd[k].cpu().contiguous())), ref, rtol=1e-12, atol=0, err_msg=k)
    o = sib.SimSiamOracle("resnet18", True, 256, 64, lr=0.005, weight_decay=1e-4)
    for s in range(2):
        a1, a2 = seeded_randn(1100 + 2 * s, 16, 3, 32, 32), seeded_randn(1101 + 2 * s, 16, 3, 32, 32)
        got, want = t.train_step({"aug_1": a1, "aug_2": a2})["loss"], o.train_step(a1, a2)["loss"]
        assert abs(got - want) < 2e-5 and abs(got - g["simsiam_losses"][s]) < 5e-5, (s, got, want)    # |loss| ~ 1e-2 (difference of unit-vector products)


def test_relic_steps_match_reference(dev, golden):
    from ssv_amd.models.relic import ReLIC
    g = golden["sibling_level"]
    t = _bare(ReLIC, dev, {**BASE, "proj_dim": 128, "tau": 0.996, "optimizer": {"name": "sgd", "lr": 0.2, "weight_decay": 1e-4},
                           "loss_fn": {"normalize": True, "temperature": 1.0, "alpha": 0.5}}, loader_len=1)
    assert t.max_steps == 1000
    losses = []
    for s in range(2):
        batch = {"img": seeded_randn(1200 + 3 * s, 16, 3, 32, 32), "aug_1": seeded_randn(1201 + 3 * s, 16, 3, 32, 32), "aug_2": seeded_randn(1202 + 3 * s, 16, 3, 32, 32)}
        losses.append(t.train_step(batch)["loss"])
        t._after_step(s)
    np.testing.assert_allclose(losses[0], g["relic_losses"][0], rtol=1e-5)
    np.testing.assert_allclose(losses[1], g["relic_losses"][1], rtol=1e-2)                # one lr-0.02 update later (chaos: see the SimCLR step test)
    state = {**{"online_network." + k: v for k, v in t.online_network.state_dict().items()},
             **{"target_network." + k: v for k, v in t.target_network.state_dict().items()}}
    for k, ref in zip(g["relic_after2_keys"], g["relic_after2_sums"]):
        k = str(k)
        if "running_mean" in k or "running_var" in k:                                      # BN statistics: order of the five passes per network
            got = oracle.tensor_checksum(state[k].cpu())
            np.testing.assert_allclose(got[1], ref[1], rtol=5e-3, err_msg=k)


def test_moco_steps_match_reference(dev, golden):
    from ssv_amd.models.moco import MoCo
    g = golden["sibling_level"]
    t = _bare(MoCo, dev, {**BASE, "proj_dim": 128, "queue_size": 40, "momentum": 0.999, "optimizer": {"name": "sgd", "lr": 0.03, "weight_decay": 1e-4},
                          "loss_fn": {"normalize": True, "temperature": 0.07}})
    losses = [t.train_step({"aug_1": seeded_randn(1300 + 2 * s, 16, 3, 32, 32), "aug_2": seeded_randn(1301 + 2 * s, 16, 3, 32, 32)})["loss"] for s in range(3)]
    np.testing.assert_allclose(losses[0], g["moco_losses"][0], rtol=2e-3, atol=2e-6)      # step 0: every negative is a zero row -> loss ~ 5e-4
    np.testing.assert_allclose(losses[1:], g["moco_losses"][1:], rtol=2e-2)               # after one / two updates: size class only
    assert t.memory_bank.ptr == int(g["moco_ptr_after3"])
    np.testing.assert_allclose(t.memory_bank.bank[:40].cpu().numpy(), g["moco_bank_after3"], rtol=2e-3, atol=5e-4)
