"""GPU: the ViT encoder and the DINO step on the HIP path against fixtures produced by the reference itself
(tests/golden/dino_level.npz) and against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

import oracle
from oracle import vit as ovit
from conftest import seeded_randn

pytestmark = pytest.mark.gpu

ENC = {"hidden_dim": 384, "embedding_dim": 192, "intermediate_dim": 768, "num_attention_heads": 6, "patch_size": 4,
       "num_local_patches": 4, "num_global_patches": 64, "num_encoder_layers": 6}
HEAD = {"hidden_dim": 512, "proj_dim": 1024}
CFG = {"epochs": 1000, "gradient_clip": 3.0, "encoder": ENC, "proj_head": HEAD,
       "optimizer": {"name": "adamw", "lr": 1e-4, "amsgrad": False, "epsilon": 1e-6, "weight_decay": 0.04},
       "scheduler": {"name": "cosine", "warmup_epochs": 10}}


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _trainer(dev):
    from ssv_amd.models.dino import DINO
    from ssv_amd.utils import train_utils
    t = object.__new__(DINO)
    t.config, t.device, t.train_loader = CFG, dev, [None]
    torch.manual_seed(420)
    t._build("vit")
    t.scheduler, t.warmup_epochs = train_utils.get_scheduler({**CFG["scheduler"], "epochs": CFG["epochs"]}, optimizer=t.optim)
    return t


def _batch(step, bs=4, vl=3):
    return {"global_1": seeded_randn(910 + 4 * step, bs, 2, 3, 32, 32), "global_2": seeded_randn(911 + 4 * step, bs, 2, 3, 32, 32),
            "local_1": seeded_randn(912 + 4 * step, bs, vl, 3, 8, 8), "local_2": seeded_randn(913 + 4 * step, bs, vl, 3, 8, 8)}


def test_vit_encoder_forward_matches_reference(dev, golden):
    from ssv_amd.networks import vit
    g = golden["dino_level"]
    torch.manual_seed(420)
    enc = vit.TransformerEncoder(dict(ENC))
    sd = enc.state_dict()
    assert [k for k in sd] == [str(k) for k in g["vit_init_keys"]]
    for k, ref in zip(g["vit_init_keys"], g["vit_init_sums"]):
        np.testing.assert_allclose(np.array(oracle.tensor_checksum(sd[str(k)])), ref, rtol=1e-12, atol=0)   # same RNG stream (sums: host-dependent order)
    enc = enc.to(dev)
    with torch.no_grad():
        fg = enc(seeded_randn(901, 3, 3, 32, 32).to(dev))
        fl = enc(seeded_randn(902, 5, 3, 8, 8).to(dev))
    np.testing.assert_allclose(fg.cpu().numpy(), g["vit_global_feats"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(fl.cpu().numpy(), g["vit_local_feats"], rtol=1e-4, atol=2e-5)
    with pytest.raises(RuntimeError):
        enc(seeded_randn(1, 2, 3, 16, 16).to(dev))                                                  # 16 patches: neither global nor local
    with pytest.raises(NotImplementedError):
        enc(seeded_randn(1, 2, 3, 8, 8).to(dev), return_attn=True)


def test_vit_encoder_gradients_match_oracle(dev):
    """Backward through the whole encoder (embedding scatter, attention, LayerNorm-residuals, GEMMs) vs the fp64 oracle."""
    from ssv_amd.networks import vit
    small = dict(ENC, num_encoder_layers=2)
    torch.manual_seed(3)
    enc = vit.TransformerEncoder(small)
    p64 = {k: v.detach().double().requires_grad_() for k, v in enc.state_dict().items()}
    x, dy = seeded_randn(5, 3, 3, 32, 32), seeded_randn(6, 3, 384)
    ovit.vit_forward(p64, small, x.double()).backward(dy.double())
    enc = enc.to(dev)
    xin = x.to(dev).requires_grad_()
    enc(xin).backward(dy.to(dev))
    assert xin.grad is None or float(xin.grad.abs().max()) == 0.0                                  # images get no gradient
    for name, p in enc.named_parameters():
        ref = p64[name].grad
        if name == "embedding.pos_embedding_local.weight":
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
            continue
        err = float((p.grad.cpu().double() - ref).norm() / (ref.norm() + 1e-30))
        assert err < 2e-5, (name, err)


def test_dino_two_steps_match_reference(dev, golden):
    g = golden["dino_level"]
    t = _trainer(dev)
    for model, tag in ((t.student_model, "student"), (t.teacher_model, "teacher")):
        sd = model.state_dict()
        assert list(sd) == [str(k) for k in g[f"dino_{tag}_init_keys"]]
        for k, ref in zip(g[f"dino_{tag}_init_keys"], g[f"dino_{tag}_init_sums"]):
            np.testing.assert_allclose(np.array(oracle.tensor_checksum(sd[str(k)].cpu())), ref, rtol=1e-12, atol=0)
    assert [n for n, _ in t.student_model.named_parameters()] == [str(k) for k in g["dino_param_order"]]
    np.testing.assert_allclose(np.array(oracle.tensor_checksum(t.teacher_center.cpu())), g["dino_center_init"], rtol=1e-12, atol=0)
    assert abs(t.optim.param_groups[0]["lr"] - float(g["dino_lr"])) < 1e-18
    o = ovit.DinoOracle(ENC, HEAD)
    b0 = _batch(0)
    with torch.no_grad():
        sg = t.student_model(b0["global_1"].flatten(0, 1).to(dev))
        tg = t.teacher_model(b0["global_1"].flatten(0, 1).to(dev))
    np.testing.assert_allclose(sg.cpu().numpy(), g["dino_student_g1_step0"], rtol=1e-4, atol=2e-5)     # "projected features"
    np.testing.assert_allclose(tg.cpu().numpy(), g["dino_teacher_g1_step0"], rtol=1e-4, atol=2e-5)
    losses = []
    for step in range(2):
        b = _batch(step)
        losses.append(t.train_step(b)["loss"])
        o.train_step(b["global_1"], b["global_2"], b["local_1"], b["local_2"])
        if step == 0:
            # raw (unclamped) gradients in the arena vs the oracle's
            for (name, p) in t.student_model.named_parameters():
                ref = o.last_grads[name]
                got = (p.grad + p._grad_alt).cpu()
                err = float((got - ref).norm() / (ref.norm() + 1e-20))
                assert err < 5e-4, (name, err)
            sd = t.student_model.state_dict()
            for k, ref in zip(g["dino_after1_keys"], g["dino_after1_sums"]):
                got = oracle.tensor_checksum(sd[str(k)].cpu())
                np.testing.assert_allclose(got[:2], ref[:2], rtol=1e-5, atol=1e-4, err_msg=str(k))
            np.testing.assert_allclose(t.teacher_center.cpu().numpy(), g["dino_center_after1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(losses, g["dino_losses"], rtol=1e-5)                                    # north-star bar: 1e-4 relative


def test_dino_epoch_schedules_and_teacher_ema(dev, golden):
    g = golden["dino_level"]
    t = _trainer(dev)
    for e, temp, wd in zip(g["dino_sched_epochs"], g["dino_teacher_temps"], g["dino_weight_decays"]):
        t.update_temperature(int(e))
        t.update_weight_decay(int(e))
        assert abs(t.temp_teacher - temp) < 1e-15 and abs(t.optim.param_groups[0]["weight_decay"] - wd) < 1e-15
    o = ovit.DinoOracle(ENC, HEAD)
    t.update_teacher_model(500)
    o.update_teacher(500, 1000)
    sd = t.teacher_model.state_dict()
    for k, v in o.teacher.items():
        np.testing.assert_allclose(sd[k].cpu().numpy(), v.numpy(), rtol=1e-6, atol=1e-7, err_msg=k)


def test_dino_nine_steps_with_epoch_schedules_track_the_oracle(dev):
    """Three "epochs" of three steps with the per-epoch updates in between (teacher EMA, weight-decay ramp, teacher temperature,
    lr warm-up / cosine): there is no BatchNorm / ReLU in this path, so the HIP trainer stays on the CPU oracle's trajectory
    (measured 2e-7 relative at step 9; bar 1e-4)."""
    from ssv_amd.models.dino import DINO
    from ssv_amd.utils import train_utils
    enc = dict(ENC, num_encoder_layers=3)
    cfg = dict(CFG, epochs=8, encoder=enc, scheduler={"name": "cosine", "warmup_epochs": 2})
    t = object.__new__(DINO)
    t.config, t.device, t.train_loader = cfg, dev, [None]
    torch.manual_seed(420)
    t._build("vit")
    t.scheduler, t.warmup_epochs = train_utils.get_scheduler({**cfg["scheduler"], "epochs": cfg["epochs"]}, optimizer=t.optim)
    t.warmup_rate = (cfg["optimizer"]["lr"] - 1e-12) / t.warmup_epochs
    o = ovit.DinoOracle(enc, HEAD, warmup_epochs=2)
    for step in range(9):
        b = {"global_1": seeded_randn(10 + 4 * step, 8, 2, 3, 32, 32), "global_2": seeded_randn(11 + 4 * step, 8, 2, 3, 32, 32),
             "local_1": seeded_randn(12 + 4 * step, 8, 3, 3, 8, 8), "local_2": seeded_randn(13 + 4 * step, 8, 3, 3, 8, 8)}
        got = t.train_step(b)["loss"]
        want = o.train_step(b["global_1"], b["global_2"], b["local_1"], b["local_2"])["loss"]
        np.testing.assert_allclose(got, want, rtol=1e-5, err_msg=f"step {step}")
        if step % 3 == 2:
            epoch = step // 3 + 1
            t._after_epoch(epoch)
            t.adjust_learning_rate(epoch)
            o.update_teacher(epoch, cfg["epochs"])
            o.weight_decay, o.temp_teacher = ovit.cosine_ramp(epoch, cfg["epochs"], 0.4, 0.04), ovit.teacher_temperature(epoch)
            o.lr = 1e-12 + epoch * (1e-4 - 1e-12) / 2 if epoch <= 2 else t.optim.param_groups[0]["lr"]
            assert abs(t.optim.param_groups[0]["lr"] - o.lr) < 1e-18 and abs(t.optim.param_groups[0]["weight_decay"] - o.weight_decay) < 1e-15
