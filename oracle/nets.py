"""Oracle (test infrastructure): encoder + head restatement in plain torch fp32 on CPU.

Functional style over a flat ``{key: tensor}`` dict whose keys and order are the
reference ``state_dict()`` keys, so checkpoints interchange.

Restates:
  * ResNet ctor / init / _make_layer / forward  - reference networks/resnet.py:78-155
  * BasicBlock.forward                          - networks/resnet.py:37-45
  * Bottleneck.forward (stride on the 3x3)      - networks/resnet.py:48-75
  * SimCLR ProjectionHead                       - models/simclr.py:23-36
  * BYOL MLP / OnlineNetwork / TargetNetwork    - models/byol.py:24-59
  * Barlow ProjectionHead                       - models/barlow.py:23-36
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

# arch -> (block kind, blocks per stage, width_per_group[, groups])   networks/resnet.py:158-193
RESNET_SPECS = {
    "resnet18": ("basic", (2, 2, 2, 2), 64),
    "resnet34": ("basic", (3, 4, 6, 3), 64),
    "resnet50": ("bottleneck", (3, 4, 6, 3), 64),
    "resnet101": ("bottleneck", (3, 4, 23, 3), 64),
    "resnet152": ("bottleneck", (3, 8, 36, 3), 64),
    "wide_resnet50": ("bottleneck", (3, 4, 6, 3), 128),
    "wide_resnet101": ("bottleneck", (3, 4, 23, 3), 128),
    "resnext50": ("bottleneck", (3, 4, 6, 3), 4, 32),
    "resnext101": ("bottleneck", (3, 4, 23, 3), 8, 32),
}
ENCODER_DIM = {"resnet18": 512, "resnet34": 512, "resnet50": 2048, "resnet101": 2048, "resnet152": 2048,
               "wide_resnet50": 2048, "wide_resnet101": 2048, "resnext50": 2048, "resnext101": 2048}


def _groups(arch):
    spec = RESNET_SPECS[arch]
    return spec[3] if len(spec) > 3 else 1
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _plan(arch, reduce_bottom_conv):
    """Enumerate convolutions of the encoder.

    Returns (creation_order, modules_order, blocks) where the two orders list
    ``(key_prefix, cout, cin, k)`` - creation order is the order in which
    ``nn.Conv2d`` objects are constructed by the reference ctor (a block's
    downsample conv is built BEFORE the block, networks/resnet.py:131-137) and
    modules order is ``self.modules()`` / state_dict order (downsample is the
    block's last attribute, networks/resnet.py:33,61).
    """
    kind, counts, base_width = RESNET_SPECS[arch][:3]
    groups = _groups(arch)
    expansion = 1 if kind == "basic" else 4
    stem = ("conv1", 64, 3, 3 if reduce_bottom_conv else 7)
    creation, modules, blocks = [stem], [stem], []
    in_planes = 64
    for stage, (planes, n) in enumerate(zip((64, 128, 256, 512), counts), start=1):
        for b in range(n):
            stride = 2 if (b == 0 and stage > 1) else 1
            prefix = f"layer{stage}.{b}"
            width = int(planes * base_width / 64) * groups               # networks/resnet.py:55
            out_planes = planes * expansion
            has_ds = b == 0 and (stride != 1 or in_planes != out_planes)
            if kind == "basic":
                convs = [(f"{prefix}.conv1", planes, in_planes, 3), (f"{prefix}.conv2", planes, planes, 3)]
            else:
                convs = [(f"{prefix}.conv1", width, in_planes, 1), (f"{prefix}.conv2", width, width // groups, 3),   # grouped 3x3: [width, width/groups, 3, 3]
                         (f"{prefix}.conv3", out_planes, width, 1)]
            ds = (f"{prefix}.downsample.0", out_planes, in_planes, 1)
            if has_ds:
                creation.append(ds)
            creation.extend(convs)
            modules.extend(convs)
            if has_ds:
                modules.append(ds)
            blocks.append(dict(prefix=prefix, kind=kind, stride=stride, downsample=has_ds))
            in_planes = out_planes
    return creation, modules, blocks


def _bn_entries(params, prefix, c):
    params[f"{prefix}.weight"] = torch.ones(c)
    params[f"{prefix}.bias"] = torch.zeros(c)
    params[f"{prefix}.running_mean"] = torch.zeros(c)
    params[f"{prefix}.running_var"] = torch.ones(c)
    params[f"{prefix}.num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def init_resnet(arch, reduce_bottom_conv=False):
    """Build the encoder parameter dict, consuming the global torch CPU RNG exactly as
    the reference ctor does (networks/resnet.py:96-115): every nn.Conv2d first draws its
    default kaiming_uniform_(a=sqrt(5)) in creation order, then every conv is re-drawn with
    kaiming_normal_(fan_out, relu) in modules() order; BN gamma=1, beta=0."""
    creation, modules, blocks = _plan(arch, reduce_bottom_conv)
    scratch = {}
    for key, co, ci, k in creation:
        w = torch.empty(co, ci, k, k)
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        scratch[key] = w
    for key, co, ci, k in modules:
        torch.nn.init.kaiming_normal_(scratch[key], mode="fan_out", nonlinearity="relu")
    params = OrderedDict()
    params["conv1.weight"] = scratch["conv1"]
    _bn_entries(params, "bn1", 64)
    for blk in blocks:
        p = blk["prefix"]
        names = ("conv1", "conv2") if blk["kind"] == "basic" else ("conv1", "conv2", "conv3")
        for i, name in enumerate(names, start=1):
            w = scratch[f"{p}.{name}"]
            params[f"{p}.{name}.weight"] = w
            _bn_entries(params, f"{p}.bn{i}", w.shape[0])
        if blk["downsample"]:
            w = scratch[f"{p}.downsample.0"]
            params[f"{p}.downsample.0.weight"] = w
            _bn_entries(params, f"{p}.downsample.1", w.shape[0])
    return params


def init_linear(params, prefix, din, dout):
    """nn.Linear default init (weight kaiming_uniform_(a=sqrt5), then bias U(-1/sqrt(fan_in), ..))."""
    w = torch.empty(dout, din)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    bound = 1.0 / math.sqrt(din)
    b = torch.empty(dout)
    torch.nn.init.uniform_(b, -bound, bound)
    params[f"{prefix}.weight"] = w
    params[f"{prefix}.bias"] = b


def _bn_train(x, params, prefix, stats=True):
    """nn.BatchNorm{1,2}d in train mode (the reference never calls .eval()):
    batch statistics, biased var for normalisation, running stats momentum 0.1 / unbiased var."""
    rm, rv = params[f"{prefix}.running_mean"], params[f"{prefix}.running_var"]
    if stats:
        params[f"{prefix}.num_batches_tracked"] += 1
    return F.batch_norm(x, rm if stats else None, rv if stats else None,
                        params[f"{prefix}.weight"], params[f"{prefix}.bias"],
                        training=True, momentum=BN_MOMENTUM, eps=BN_EPS)


def resnet_forward(params, x, arch, reduce_bottom_conv=False):
    """ResNet.forward (networks/resnet.py:146-155): NCHW fp32 [B,3,H,W] -> [B,dim]; no fc."""
    _, _, blocks = _plan(arch, reduce_bottom_conv)
    if reduce_bottom_conv:
        x = F.conv2d(x, params["conv1.weight"], stride=1, padding=1)
    else:
        x = F.conv2d(x, params["conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn_train(x, params, "bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for blk in blocks:
        p, s = blk["prefix"], blk["stride"]
        identity = x
        if blk["kind"] == "basic":
            out = F.conv2d(x, params[f"{p}.conv1.weight"], stride=s, padding=1)
            out = F.relu(_bn_train(out, params, f"{p}.bn1"))
            out = F.conv2d(out, params[f"{p}.conv2.weight"], stride=1, padding=1)
            out = _bn_train(out, params, f"{p}.bn2")
        else:
            out = F.conv2d(x, params[f"{p}.conv1.weight"])
            out = F.relu(_bn_train(out, params, f"{p}.bn1"))
            out = F.conv2d(out, params[f"{p}.conv2.weight"], stride=s, padding=1, groups=_groups(arch))
            out = F.relu(_bn_train(out, params, f"{p}.bn2"))
            out = F.conv2d(out, params[f"{p}.conv3.weight"])
            out = _bn_train(out, params, f"{p}.bn3")
        if blk["downsample"]:
            identity = F.conv2d(x, params[f"{p}.downsample.0.weight"], stride=s)
            identity = _bn_train(identity, params, f"{p}.downsample.1")
        x = F.relu(out + identity)
    return x.mean(dim=(2, 3))


# ----------------------------------------------------------------------------- heads
def init_simclr_head(din, dout):
    """models/simclr.py:25-31: fc1(d->d), bn1, fc2(d->D), bn2."""
    p = OrderedDict()
    init_linear(p, "fc1", din, din)
    _bn_entries(p, "bn1", din)
    init_linear(p, "fc2", din, dout)
    _bn_entries(p, "bn2", dout)
    return p


def simclr_head_forward(p, x):
    """models/simclr.py:33-36: bn2(fc2(relu(bn1(fc1(x)))))."""
    x = F.relu(_bn_train(F.linear(x, p["fc1.weight"], p["fc1.bias"]), p, "bn1"))
    return _bn_train(F.linear(x, p["fc2.weight"], p["fc2.bias"]), p, "bn2")


def init_byol_mlp(din, dout, prefix=""):
    """models/byol.py:26-31: fc1(d->d), bn1, fc2(d->D)."""
    p = OrderedDict()
    init_linear(p, f"{prefix}fc1", din, din)
    _bn_entries(p, f"{prefix}bn1", din)
    init_linear(p, f"{prefix}fc2", din, dout)
    return p


def byol_mlp_forward(p, x, prefix=""):
    """models/byol.py:33-34: fc2(relu(bn1(fc1(x))))."""
    x = F.relu(_bn_train(F.linear(x, p[f"{prefix}fc1.weight"], p[f"{prefix}fc1.bias"]), p, f"{prefix}bn1"))
    return F.linear(x, p[f"{prefix}fc2.weight"], p[f"{prefix}fc2.bias"])


def init_barlow_head(din, dproj):
    """models/barlow.py:25-29: (Linear,BN,ReLU) x2 then Linear, all width dproj."""
    p = OrderedDict()
    init_linear(p, "layer1.0", din, dproj)
    _bn_entries(p, "layer1.1", dproj)
    init_linear(p, "layer2.0", dproj, dproj)
    _bn_entries(p, "layer2.1", dproj)
    init_linear(p, "layer3", dproj, dproj)
    return p


def barlow_head_forward(p, x):
    """models/barlow.py:31-36: layer1 -> layer2 -> layer3 -> L2-normalise rows."""
    x = F.relu(_bn_train(F.linear(x, p["layer1.0.weight"], p["layer1.0.bias"]), p, "layer1.1"))
    x = F.relu(_bn_train(F.linear(x, p["layer2.0.weight"], p["layer2.0.bias"]), p, "layer2.1"))
    x = F.linear(x, p["layer3.weight"], p["layer3.bias"])
    return F.normalize(x, p=2, dim=-1)


def tensor_checksum(t):
    """(sum, sum of squares, first 4 values) in float64 - used to pin init / grads / params."""
    d = t.detach().double().flatten()
    head = d[:4].tolist() + [0.0] * (4 - min(4, d.numel()))
    return [float(d.sum()), float((d * d).sum())] + head
