"""CPU: the SimSiam / ReLIC / MoCo oracle (oracle/siblings.py) against fixtures produced by the reference (sibling_level.npz)."""
import numpy as np
import torch
import torch.nn.functional as F

import oracle
from oracle import siblings as sib
from conftest import seeded_randn


def _check_state(state, keys, sums, rtol, atol, skip=()):
    for k, ref in zip(keys, sums):
        k = str(k)
        if any(s in k for s in skip):
            continue
        got = oracle.tensor_checksum(state[k])
        np.testing.assert_allclose(got[:2], ref[:2], rtol=rtol, atol=atol, err_msg=k)


def test_sibling_losses_match_reference(golden):
    g = golden["sibling_level"]
    o, t = F.normalize(seeded_randn(1001, 12, 64), dim=1).requires_grad_(), F.normalize(seeded_randn(1002, 12, 64), dim=1)
    l = sib.simsiam_loss(o, t)
    l.backward()
    np.testing.assert_allclose(l.item(), g["simsiam_loss"], rtol=1e-6)
    np.testing.assert_allclose(o.grad.numpy(), g["simsiam_do"], rtol=1e-6, atol=1e-9)
    for tag, norm, temp, alpha in (("a", True, 1.0, 0.5), ("b", False, 0.5, 0.3)):
        zi, zj, zo = (seeded_randn(1010 + i, 10, 32).requires_grad_() for i in range(3))
        l = sib.relic_loss(zi, zj, zo, norm, temp, alpha)
        l.backward()
        np.testing.assert_allclose(l.item(), g[f"relic_{tag}_loss"], rtol=2e-6)
        for name, t_ in (("dzi", zi), ("dzj", zj), ("dzo", zo)):
            np.testing.assert_allclose(t_.grad.numpy(), g[f"relic_{tag}_{name}"], rtol=2e-4, atol=1e-7, err_msg=f"{tag} {name}")
    q, k, bank = seeded_randn(1020, 9, 32).requires_grad_(), seeded_randn(1021, 9, 32), F.normalize(seeded_randn(1022, 50, 32), dim=1)
    bank[40:] = 0.0
    l = sib.moco_loss(q, k, bank, True, 0.07)
    l.backward()
    np.testing.assert_allclose(l.item(), g["moco_loss"], rtol=2e-6)
    np.testing.assert_allclose(q.grad.numpy(), g["moco_dq"], rtol=2e-4, atol=1e-7)


def test_simsiam_two_steps_match_reference(golden):
    g = golden["sibling_level"]
    m = sib.SimSiamOracle("resnet18", True, 256, 64, lr=0.005, weight_decay=1e-4)
    st = {f"online_network.{k}": v for k, v in m.state().items()}
    for k, ref in zip(g["simsiam_init_keys"], g["simsiam_init_sums"]):
        np.testing.assert_allclose(np.array(oracle.tensor_checksum(st[str(k)])), ref, rtol=1e-12, atol=0, err_msg=str(k))
    losses = [m.train_step(seeded_randn(1100 + 2 * s, 16, 3, 32, 32), seeded_randn(1101 + 2 * s, 16, 3, 32, 32))["loss"] for s in range(2)]
    np.testing.assert_allclose(losses, g["simsiam_losses"], rtol=1e-3, atol=2e-5)       # |loss| ~ 1e-2: a difference of unit-vector dot products


def test_relic_two_steps_match_reference(golden):
    g = golden["sibling_level"]
    m = sib.RelicOracle("resnet18", True, 128, lr=0.02, weight_decay=1e-4)
    losses = []
    for s in range(2):
        losses.append(m.train_step(seeded_randn(1200 + 3 * s, 16, 3, 32, 32), seeded_randn(1201 + 3 * s, 16, 3, 32, 32),
                                   seeded_randn(1202 + 3 * s, 16, 3, 32, 32), step=s)["loss"])
    np.testing.assert_allclose(losses, g["relic_losses"], rtol=1e-4)
    _check_state(m.state(), g["relic_after2_keys"], g["relic_after2_sums"], rtol=2e-3, atol=2e-3, skip=("num_batches",))


def test_moco_three_steps_match_reference(golden):
    g = golden["sibling_level"]
    m = sib.MocoOracle("resnet18", True, 128, queue_size=40, momentum=0.999, lr=0.003, weight_decay=1e-4, temperature=0.07)
    losses = [m.train_step(seeded_randn(1300 + 2 * s, 16, 3, 32, 32), seeded_randn(1301 + 2 * s, 16, 3, 32, 32))["loss"] for s in range(3)]
    np.testing.assert_allclose(losses, g["moco_losses"], rtol=2e-3, atol=1e-6)
    assert m.ptr == int(g["moco_ptr_after3"]) == 8
    np.testing.assert_allclose(m.bank.numpy(), g["moco_bank_after3"], rtol=1e-3, atol=2e-4)
