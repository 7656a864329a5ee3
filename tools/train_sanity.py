#!/usr/bin/env python3
"""Many-step sanity run on a small synthetic dataset through the real CLI trainers: the loss must go down and stay finite.
With `patterns` (a dataset whose label is learnable) the kNN accuracy of every epoch is reported too: it must rise above chance (0.1).
usage: train_sanity.py <algo> [epochs] [noise|patterns] [arch = resnet18 | resnet50 ...] [image size = 32]
(resnet50 at 112 px runs the standard 7x7 stem and the 28x28 / 56x56-class feature maps, i.e. every fused path of the bench configuration)"""
import os, sys, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yaml
from ssv_amd import main as cli

algo = sys.argv[1]
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "self-supervised-vision_amd", "configs")
cfg = yaml.safe_load(open(os.path.join(root, f"{algo}.yaml")))
cfg["epochs"], cfg["eval_every"] = epochs, epochs
cfg["data"]["batch_size"] = 128
kind = sys.argv[3] if len(sys.argv) > 3 else "noise"
arch_arg = sys.argv[4] if len(sys.argv) > 4 else None
size = int(sys.argv[5]) if len(sys.argv) > 5 else 32
cfg["data"]["synthetic"] = {"num_train": 2048 if kind == "patterns" else 1024, "num_test": 512 if kind == "patterns" else 256, "image_size": [size, size],
                            "num_classes": 10, "kind": kind}
if kind == "patterns":
    cfg["eval_every"] = 1
if size != 32:                      # the 224-class pipeline: standard stem, crops of the synthetic images at their own size
    cfg["encoder"]["reduce_bottom_conv"] = False
    for split in cfg["data"]["transforms"].values():
        for name, args in split.items():
            if isinstance(args, dict) and "size" in args:
                args["size"] = [size, size]
cfg["linear_eval"]["epochs"] = 1
cfg["scheduler"]["warmup_epochs"] = min(cfg["scheduler"].get("warmup_epochs", 0), 2)
if algo == "barlow":
    cfg["proj_dim"] = 512
os.environ["WANDB_MODE"] = "disabled"
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "cfg.yaml")
    open(path, "w").write(yaml.dump(cfg, sort_keys=False))
    os.chdir(tmp)
    arch = "vit" if algo == "dino" else (arch_arg or "resnet18")
    model = cli.main(["-c", path, "-a", algo, "-m", arch, "-t", "train", "-o", "run"])
    log = open(os.path.join(tmp, "outputs", algo, arch, "run", "trainlogs.txt")).read().splitlines()
    losses = [float(l.split("[loss]")[1].split()[0]) for l in log if "[loss]" in l]
accs = [float(l.split("[accuracy]")[1].split()[0]) for l in log if "[accuracy]" in l]
print(json.dumps({"algo": algo, "data": kind, "epoch_mean_losses": losses, "knn_accuracy": accs}))
