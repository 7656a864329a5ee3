"""Pin the CPU oracle (oracle/) against fixtures produced by the reference itself
(tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import oracle
from conftest import seeded_randn


def _close(a, b, rtol, atol=0.0):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def test_ntxent_matches_reference(golden):
    g = golden["loss_level"]
    for tag, (n, d, seed, norm, temp) in zip("abcde", g["ntxent_cases"]):
        n, d, seed = int(n), int(d), int(seed)
        zi, zj = seeded_randn(seed, n, d).requires_grad_(), seeded_randn(seed + 1000, n, d).requires_grad_()
        loss = oracle.ntxent_loss(zi, zj, bool(norm), float(temp))
        loss.backward()
        _close(loss.item(), g[f"ntxent_{tag}_loss"], rtol=2e-6)
        _close(zi.grad, g[f"ntxent_{tag}_dzi"], rtol=1e-4, atol=2e-7)
        _close(zj.grad, g[f"ntxent_{tag}_dzj"], rtol=1e-4, atol=2e-7)


def test_barlow_matches_reference(golden):
    g = golden["loss_level"]
    for tag, (b, d, seed, norm, lm) in zip("abc", g["barlow_cases"]):
        b, d, seed = int(b), int(d), int(seed)
        zi, zj = seeded_randn(seed, b, d).requires_grad_(), seeded_randn(seed + 1000, b, d).requires_grad_()
        loss = oracle.barlow_loss(zi, zj, bool(norm), float(lm))
        loss.backward()
        _close(loss.item(), g[f"barlow_{tag}_loss"], rtol=5e-6)
        _close(zi.grad, g[f"barlow_{tag}_dzi"], rtol=2e-4, atol=1e-6)
        _close(zj.grad, g[f"barlow_{tag}_dzj"], rtol=2e-4, atol=1e-6)


def test_byol_loss_matches_reference(golden):
    g = golden["loss_level"]
    p1, p2 = seeded_randn(31, 16, 128).requires_grad_(), seeded_randn(32, 16, 128).requires_grad_()
    t1, t2 = oracle.l2_normalize(seeded_randn(33, 16, 128)), oracle.l2_normalize(seeded_randn(34, 16, 128))
    loss = oracle.byol_mse_loss(oracle.l2_normalize(p1), oracle.l2_normalize(p2), t1, t2)
    loss.backward()
    _close(loss.item(), g["byol_loss"], rtol=2e-6)
    _close(p1.grad, g["byol_dp1"], rtol=1e-4, atol=1e-9)
    _close(p2.grad, g["byol_dp2"], rtol=1e-4, atol=1e-9)


def test_init_matches_reference_rng_stream(golden):
    g = golden["init_checksums"]
    for tag, arch, rbc in (("r18rbc", "resnet18", True), ("r50", "resnet50", False), ("r50rbc", "resnet50", True)):
        torch.manual_seed(420)
        enc = oracle.init_resnet(arch, rbc)
        head = oracle.init_simclr_head(oracle.nets.ENCODER_DIM[arch], 128)
        assert list(enc.keys()) == list(g[f"{tag}_all_keys"])
        for keys, sums, d in ((g[f"{tag}_enc_keys"], g[f"{tag}_enc_sums"], enc), (g[f"{tag}_head_keys"], g[f"{tag}_head_sums"], head)):
            for k, ref in zip(keys, sums):
                _close(oracle.tensor_checksum(d[str(k)]), ref, rtol=0, atol=0)  # bit-exact init
        n = sum(v.numel() for k, v in {**enc, **{"h." + k: v for k, v in head.items()}}.items()
                if k.endswith(".weight") or k.endswith(".bias"))
        assert n == int(g[f"{tag}_nparams"])


def _check_state(state, keys, sums, rtol):
    for k, ref in zip(keys, sums):
        got = oracle.tensor_checksum(state[str(k)])
        scale = float(np.sqrt(ref[1])) + 1e-6      # a sum of +- values cancels: tolerance relative to the l2 norm
        _close(got[0], ref[0], rtol=rtol, atol=2e-5 * scale)
        _close(got[1], ref[1], rtol=rtol, atol=1e-9)
        _close(got[2:], ref[2:], rtol=rtol * 10, atol=2e-6)


def test_simclr_r18_steps_match_reference(golden):
    g = golden["step_level"]
    m = oracle.SimCLROracle("resnet18", True, 128, lr=float(g["simclr_r18_lr"]), weight_decay=1e-4)
    assert abs(m.lr - oracle.seeded_lr(2.0, 10)) < 1e-15
    losses = []
    for s in range(3):
        out = m.train_step(seeded_randn(100 + 2 * s, 64, 3, 32, 32), seeded_randn(101 + 2 * s, 64, 3, 32, 32), return_z=(s == 0))
        losses.append(out["loss"])
        if s == 0:
            _close(out["z_1"], g["simclr_r18_z1"], rtol=1e-4, atol=1e-5)
            _close(out["z_2"], g["simclr_r18_z2"], rtol=1e-4, atol=1e-5)
            _check_state(m.state(), g["simclr_r18_after1_keys"], g["simclr_r18_after1_sums"], rtol=1e-5)
    # step 0 is a pure function of the inputs; later steps amplify fp32 rounding differences of the
    # first update (lr 0.2), so they get the north-star tolerance (1e-4 rel)
    _close(losses[0], g["simclr_r18_losses"][0], rtol=2e-6)
    _close(losses, g["simclr_r18_losses"], rtol=1e-4)


def test_simclr_r50_steps_match_reference(golden):
    g = golden["step_level"]
    m = oracle.SimCLROracle("resnet50", False, 128, lr=float(g["simclr_r50_lr"]), weight_decay=1e-4)
    losses = []
    for s in range(3):
        out = m.train_step(seeded_randn(200 + 2 * s, 8, 3, 64, 64), seeded_randn(201 + 2 * s, 8, 3, 64, 64), return_z=(s == 0))
        losses.append(out["loss"])
        if s == 0:
            _close(out["z_1"], g["simclr_r50_z1"], rtol=1e-4, atol=1e-5)
    # B=8 at 64x64 leaves 2x2x8 = 32 samples per layer4 BatchNorm channel and lr is 0.2: steps >= 1 are
    # chaotic (the reference itself moves 0.3 % at step 1 and 7 % at step 2 between 1 and 8 CPU threads),
    # so only step 0 is a parity statement here; step 1 gets a loose sanity band.
    _close(losses[0], g["simclr_r50_losses"][0], rtol=2e-6)
    _close(losses[1], g["simclr_r50_losses"][1], rtol=2e-2)


def test_barlow_r18_steps_match_reference(golden):
    g = golden["step_level"]
    m = oracle.BarlowOracle("resnet18", True, 256, lr=float(g["barlow_r18_lr"]), weight_decay=1.5e-6)
    losses = [m.train_step(seeded_randn(300 + 2 * s, 32, 3, 32, 32), seeded_randn(301 + 2 * s, 32, 3, 32, 32))["loss"] for s in range(2)]
    _close(losses[0], g["barlow_r18_losses"][0], rtol=1e-5)
    _close(losses, g["barlow_r18_losses"], rtol=1e-4)


def test_features_match_reference(golden):
    g = golden["step_level"]
    m = oracle.SimCLROracle("resnet18", True, 128)
    _close(m.features(seeded_randn(400, 16, 3, 32, 32)), g["features_r18"], rtol=1e-4, atol=1e-6)


def test_byol_r18_steps_match_reference(golden):
    g = golden["step_level"]
    m = oracle.BYOLOracle("resnet18", True, 128, lr=float(g["byol_r18_lr"]), weight_decay=1e-4, max_steps=1000)
    losses, taus = [], []
    for s in range(2):
        losses.append(m.train_step(seeded_randn(500 + 2 * s, 16, 3, 32, 32), seeded_randn(501 + 2 * s, 16, 3, 32, 32), step=s)["loss"])
        taus.append(m.tau)
    _close(losses[0], g["byol_r18_losses"][0], rtol=2e-6)
    _close(losses, g["byol_r18_losses"], rtol=1e-4)
    _close(taus, g["byol_r18_taus"], rtol=1e-12)
    _check_state(m.state(), g["byol_r18_after2_keys"], g["byol_r18_after2_sums"], rtol=1e-5)


def test_sgd_and_lr_schedule_match_reference(golden):
    g = golden["optim_level"]
    ps = [torch.tensor(g["sgd_p0_init"]), torch.tensor(g["sgd_p1_init"])]
    bufs = [None, None]
    for s in range(3):
        grads = [seeded_randn(50 + 10 * s + i, *p.shape) for i, p in enumerate(ps)]
        oracle.sgd_nesterov_step(ps, grads, bufs, lr=0.3, weight_decay=1e-2)
        _close(ps[0], g[f"sgd_p0_step{s}"], rtol=1e-6, atol=1e-7)
        _close(ps[1], g[f"sgd_p1_step{s}"], rtol=1e-6, atol=1e-7)
    assert abs(oracle.seeded_lr(2.0, 10) - g["lr_schedule"][0]) < 1e-15


def test_knn_accuracy_oracle_matches_reference(golden):
    """oracle/evalknn.py == the reference's compute_neighbor_accuracy (run over an exact-search stand-in for faiss)."""
    from oracle import evalknn
    g = golden["eval_level"]
    for (seed, n, d, classes, spread, k), want in zip(g["knn_cases"], g["knn_accuracy"]):
        fvecs, labels = evalknn.clustered_features(int(seed), int(n), int(d), int(classes), float(spread))
        assert evalknn.compute_neighbor_accuracy(fvecs, labels, k=int(k)) == want


def test_resnext_oracle_matches_reference(golden):
    """Grouped 3x3 convolutions (resnext50_32x4d): init draws, features and parameter gradients of the oracle vs the reference."""
    import torch
    g = golden["resnext_level"]
    torch.manual_seed(420)
    p = oracle.init_resnet("resnext50", True)
    for k, ref in zip(g["init_keys"], g["init_sums"]):
        np.testing.assert_allclose(np.array(oracle.tensor_checksum(p[str(k)])), ref, rtol=1e-12, atol=0, err_msg=str(k))
    for k, v in p.items():
        if v.dtype.is_floating_point and not k.split(".")[-1].startswith("running"):
            v.requires_grad_(True)
    y = oracle.resnet_forward(p, seeded_randn(1400, 4, 3, 32, 32), "resnext50", True)
    np.testing.assert_allclose(y.detach().numpy(), g["features"], rtol=1e-4, atol=1e-5)
    y.backward(seeded_randn(1401, 4, 2048))
    for k, ref in zip(g["grad_keys"], g["grad_sums"]):
        got = oracle.tensor_checksum(p[str(k)].grad)
        np.testing.assert_allclose(got[1], ref[1], rtol=2e-3, err_msg=str(k))


@pytest.mark.parametrize("arch", ["wide_resnet50", "wide_resnet101", "resnext101"])
def test_remaining_archs_oracle_matches_reference(golden, arch):
    """The rest of main.py's --arch list (networks/resnet.py:174-193: wide_resnet50_2, wide_resnet101_2, resnext101_32x8d): init draws
    (RNG-stream parity), features and parameter gradients of the oracle vs the reference's own numbers (tests/golden/arch_level.npz)."""
    import torch
    g = golden["arch_level"]
    torch.manual_seed(420)
    p = oracle.init_resnet(arch, True)
    assert [k for k, v in p.items() if v.dtype.is_floating_point] == [str(k) for k in g[f"{arch}_init_keys"]]
    for k, ref in zip(g[f"{arch}_init_keys"], g[f"{arch}_init_sums"]):
        np.testing.assert_allclose(np.array(oracle.tensor_checksum(p[str(k)])), ref, rtol=1e-12, atol=0, err_msg=str(k))
    for k, v in p.items():
        if v.dtype.is_floating_point and not k.split(".")[-1].startswith("running"):
            v.requires_grad_(True)
    y = oracle.resnet_forward(p, seeded_randn(1700, 8, 3, 32, 32), arch, True)
    np.testing.assert_allclose(y.detach().numpy(), g[f"{arch}_features"], rtol=1e-4, atol=1e-5)
    y.backward(seeded_randn(1701, 8, 2048))
    for k, ref in zip(g[f"{arch}_grad_keys"], g[f"{arch}_grad_sums"]):
        got = oracle.tensor_checksum(p[str(k)].grad)
        np.testing.assert_allclose(got[1], ref[1], rtol=2e-3, err_msg=str(k))
