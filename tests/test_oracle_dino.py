"""CPU: the ViT / DINO oracle (oracle/vit.py) against fixtures produced by the reference itself (tests/golden/dino_level.npz)."""
import numpy as np
import torch

import oracle
from oracle import vit as ovit
from conftest import seeded_randn

ENC = {"hidden_dim": 384, "embedding_dim": 192, "intermediate_dim": 768, "num_attention_heads": 6, "patch_size": 4,
       "num_local_patches": 4, "num_global_patches": 64, "num_encoder_layers": 6}
HEAD = {"hidden_dim": 512, "proj_dim": 1024}


def _check_sums(params, keys, sums, rtol=1e-6, atol=1e-6):
    for k, ref in zip(keys, sums):
        got = oracle.tensor_checksum(params[str(k)])
        np.testing.assert_allclose(got[:2], ref[:2], rtol=rtol, atol=atol, err_msg=str(k))


def test_vit_init_and_forward_match_reference(golden):
    g = golden["dino_level"]
    torch.manual_seed(420)
    p = ovit.init_vit(ENC)
    assert list(p) == [str(k) for k in g["vit_init_keys"]]                      # same keys, same order
    for k, ref in zip(g["vit_init_keys"], g["vit_init_sums"]):
        np.testing.assert_allclose(np.array(oracle.tensor_checksum(p[str(k)])), ref, rtol=1e-12, atol=0)      # same RNG stream
    with torch.no_grad():
        fg = ovit.vit_forward(p, ENC, seeded_randn(901, 3, 3, 32, 32))
        fl = ovit.vit_forward(p, ENC, seeded_randn(902, 5, 3, 8, 8))
    np.testing.assert_allclose(fg.numpy(), g["vit_global_feats"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(fl.numpy(), g["vit_local_feats"], rtol=1e-5, atol=2e-6)


def test_dino_loss_matches_reference(golden):
    g = golden["dino_level"]
    t, st, c = seeded_randn(903, 4, 2, 64), seeded_randn(904, 4, 5, 64).requires_grad_(), seeded_randn(905, 1, 64)
    l = ovit.dino_loss(t, st, 0.1, 0.04, c)
    l.backward()
    np.testing.assert_allclose(l.item(), g["dinoloss_value"], rtol=1e-6)
    np.testing.assert_allclose(st.grad.numpy(), g["dinoloss_dstudent"], rtol=1e-5, atol=1e-8)


def test_dino_two_steps_match_reference(golden):
    g = golden["dino_level"]
    m = ovit.DinoOracle(ENC, HEAD)
    assert [k for k in m.student if not k.endswith("num_batches_tracked")] == [str(k) for k in g["dino_param_order"]]
    for params, tag in ((m.student, "student"), (m.teacher, "teacher")):
        for k, ref in zip(g[f"dino_{tag}_init_keys"], g[f"dino_{tag}_init_sums"]):
            np.testing.assert_allclose(np.array(oracle.tensor_checksum(params[str(k)])), ref, rtol=1e-12, atol=0)
    np.testing.assert_allclose(np.array(oracle.tensor_checksum(m.center)), g["dino_center_init"], rtol=1e-12, atol=0)
    assert abs(m.lr - float(g["dino_lr"])) < 1e-18
    bs, vl = 4, 3
    losses = []
    for step in range(2):
        b = [seeded_randn(910 + 4 * step, bs, 2, 3, 32, 32), seeded_randn(911 + 4 * step, bs, 2, 3, 32, 32),
             seeded_randn(912 + 4 * step, bs, vl, 3, 8, 8), seeded_randn(913 + 4 * step, bs, vl, 3, 8, 8)]
        if step == 0:
            with torch.no_grad():
                sg = m.forward_student(b[0].view(-1, 3, 32, 32))
                tg = ovit.dino_model_forward(m.teacher, ENC, b[0].view(-1, 3, 32, 32))
            np.testing.assert_allclose(sg.numpy(), g["dino_student_g1_step0"], rtol=1e-5, atol=2e-6)
            np.testing.assert_allclose(tg.numpy(), g["dino_teacher_g1_step0"], rtol=1e-5, atol=2e-6)
        losses.append(m.train_step(*b)["loss"])
        if step == 0:
            clipped = {k: v.clamp(-3.0, 3.0) for k, v in m.last_grads.items()}
            _check_sums(clipped, g["dino_grad_keys"], g["dino_grad_sums_step0"], rtol=2e-4, atol=1e-7)
            _check_sums(m.student, g["dino_after1_keys"], g["dino_after1_sums"], rtol=1e-6, atol=1e-5)
            np.testing.assert_allclose(m.center.numpy(), g["dino_center_after1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(losses, g["dino_losses"], rtol=2e-6)
    m.update_teacher(500, 1000)                                                    # once per epoch in the reference (models/dino.py:228)
    _check_sums(m.teacher, g["dino_teacher_after_ema_keys"], g["dino_teacher_after_ema_sums"], rtol=1e-5, atol=1e-5)


def test_dino_schedules_match_reference(golden):
    g = golden["dino_level"]
    for e, t, wd, lbd in zip(g["dino_sched_epochs"], g["dino_teacher_temps"], g["dino_weight_decays"], g["dino_lambdas"]):
        assert abs(ovit.teacher_temperature(int(e)) - t) < 1e-15
        assert abs(ovit.cosine_ramp(int(e), 1000, 0.4, 0.04) - wd) < 1e-15
        assert abs(ovit.cosine_ramp(int(e), 1000, 1.0, 0.996) - lbd) < 1e-12


def test_multicrop_resize_is_plain_bicubic_interpolation():
    """The tensor path of RandomResizedCrop(BICUBIC): fixed points - a crop of the target size is returned unchanged and a
    constant image stays constant (cubic-convolution weights sum to one)."""
    v = seeded_randn(3, 3, 32, 32)
    np.testing.assert_allclose(ovit.multicrop_resize(v, (4, 6, 8, 8), (8, 8)).numpy(), v[:, 4:12, 6:14].numpy(), atol=1e-6)
    c = torch.full((3, 32, 32), 0.75)
    np.testing.assert_allclose(ovit.multicrop_resize(c, (1, 2, 20, 13), (8, 8)).numpy(), 0.75, rtol=1e-6)
