"""Parity under every switch of the Python host.  The switches are read once at import, so each runs in a child interpreter: one pytest
child per switch executes the golden-step body of tests/test_gpu_step.py - the resnet18 / 32 x 32 / bs 64 steps against the reference's goldens and the oracle
(models/simclr.py:86-95 at configs/simclr.yaml's shapes) - plus, where the switch's kernels only exist in another network, that network's golden steps (ResNet-50
step 0, DINO, ResNeXt) - at UNCHANGED tolerances.  A switch combination nobody ran is a product
state nobody tested: the parameter list below IS the documented list (test_switch_list_is_the_documented_one, no GPU needed)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (variable, value the child runs with).  Numeric thresholds get a value that changes the kernel selection of the tested networks.
SWITCHES = [
    ("SSV_ARITHMETIC", "f32"),
    ("SSV_SINGLE_STREAM", "1"),
    ("SSV_NO_WINOGRAD", "1"), ("SSV_WINOGRAD_MIN_CHANNELS", "64"), ("SSV_WINOGRAD_MIN_TILES", "16"),
    ("SSV_WINOGRAD44", "0"), ("SSV_WINOGRAD44_WGRAD", "0"), ("SSV_WINOGRAD44_WGRAD_CHUNK", "0"), ("SSV_WINOGRAD44_WGRAD_FLUSH", "0"), ("SSV_WINOGRAD44_FWD_RATIO", "0.6"),
    ("SSV_WINOGRAD44_DY_BOTH", "0"), ("SSV_WINOGRAD44_MIN_TILES", "16"), ("SSV_WINOGRAD44_MIN_CHANNELS", "64"), ("SSV_WINOGRAD_KEEP_V", "0"),
    ("SSV_NO_NARROW_WINO_INPUT_FUSION", "1"), ("SSV_NO_NARROW_3X3_INPUT_FUSION", "1"),
    ("SSV_NTXENT_SPLITS", "1"),
    ("SSV_STEP_GRAPH", "0"),
    ("SSV_NO_BN_STATS_FUSION", "1"), ("SSV_NO_BN_APPLY_FUSION", "1"), ("SSV_NO_BN_APPLY_FUSION_3X3", "1"), ("SSV_NO_BN_BWD_FUSION", "1"),
    ("SSV_NO_BN_DY_FUSION", "1"), ("SSV_BN_DY_MIN_HW", "0"), ("SSV_BN_DY_MIN_K", "64"),
    ("SSV_NO_CLOSING_FUSION", "1"), ("SSV_CLOSING_HW", "0,100000"),
    ("SSV_NO_SHORTCUT_GATE", "1"), ("SSV_NO_POOLED_STEM_REDUCE", "1"),
    ("SSV_NO_STEM_POOL_FUSION", "1"), ("SSV_NO_ROW_STEM", "1"), ("SSV_NO_STEM_PADDING", "1"),
    ("SSV_NO_BIAS_GRAD_FUSION", "1"), ("SSV_NO_GELUGRAD_FWD_KERNEL", "1"), ("SSV_NO_GELU_DACT", "1"), ("SSV_NO_GROUP_AWARE_TILES", "1"),
    ("SSV_NO_INPUT_STREAM", "1"), ("SSV_LATE_LOSS_READ", "1"), ("SSV_NO_COMPACT_S2_DGRAD", "1"),
    ("SSV_DIST_FORCE", "1"), ("SSV_DIST_BUCKETS", "0"), ("SSV_DIST_BACKEND", "gloo"),
    ("SSV_HIP_LIB", os.path.join(ROOT, "self-supervised-vision_amd", "csrc", "libssv_hip.so")),
]
# variables of the section that configure a harness, not the product's kernels: listed in INTEGRATION.md, nothing to run
HARNESS_ONLY = {"SSV_BENCH_BATCH", "SSV_BENCH_VIA_STEP"}
R18 = "tests/test_gpu_step.py::test_simclr_r18_steps_match_reference_and_oracle"          # the reviewer's body: BASELINE config 1's shapes, reference goldens + oracle
R50 = "tests/test_gpu_step.py::test_simclr_r50_step0_matches_reference"                   # bottleneck units, 7x7 stem + max-pool, projection shortcuts: step-0 goldens
DINO = "tests/test_gpu_dino.py::test_dino_two_steps_match_reference"                      # the reference's ViT + DINO head: two golden steps
RESNEXT = "tests/test_gpu_step.py::test_resnext50_grouped_convs_match_reference"
# which golden bodies a switch can change (every child runs the r18 body; a second body only where the switch's kernels are not in resnet18 / 32 x 32)
EXTRA = {**{n: [R50] for n in ("SSV_ARITHMETIC", "SSV_NO_BN_APPLY_FUSION", "SSV_NO_BN_DY_FUSION", "SSV_BN_DY_MIN_HW", "SSV_BN_DY_MIN_K", "SSV_NO_CLOSING_FUSION", "SSV_CLOSING_HW",
                                 "SSV_NO_SHORTCUT_GATE", "SSV_NO_POOLED_STEM_REDUCE", "SSV_NO_STEM_POOL_FUSION", "SSV_NO_ROW_STEM", "SSV_NO_STEM_PADDING",
                                 "SSV_NO_COMPACT_S2_DGRAD", "SSV_NO_BN_BWD_FUSION", "SSV_NO_NARROW_3X3_INPUT_FUSION", "SSV_WINOGRAD44_MIN_CHANNELS")},
         "SSV_NO_GELUGRAD_FWD_KERNEL": [DINO], "SSV_NO_GELU_DACT": [DINO], "SSV_NO_BIAS_GRAD_FUSION": [DINO], "SSV_NO_GROUP_AWARE_TILES": [RESNEXT]}


def _documented():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## C. Diagnostic switches"):text.index("## D. Multi-GPU launch")]
    names = set()
    for line in sec.splitlines():
        if line.startswith("| `SSV_"):
            cell, tick = "", False                              # the first cell, split at the first '|' OUTSIDE backticks (`gloo|nccl` has one inside)
            for ch in line[1:]:
                if ch == "`":
                    tick = not tick
                if ch == "|" and not tick:
                    break
                cell += ch
            names.update(re.findall(r"`(SSV_[A-Z0-9_]+)", cell))
    return names


def test_switch_list_is_the_documented_one():
    """The table of INTEGRATION.md section C and the parameter list of the parity test name the same variables - and the package reads no other."""
    listed = {n for n, _ in SWITCHES} | HARNESS_ONLY
    assert _documented() == listed, (sorted(_documented() - listed), sorted(listed - _documented()))
    read = set()
    for dirpath, _, files in os.walk(os.path.join(ROOT, "self-supervised-vision_amd")):
        for f in files:
            if f.endswith(".py"):
                read.update(re.findall(r'os\.environ(?:\.get)?[\(\[]\s*"(SSV_[A-Z0-9_]+)"', open(os.path.join(dirpath, f)).read()))
    read -= {"SSV_LAUNCHED_BY"}                                    # set BY the launcher for its ranks, not a switch
    assert read <= listed, sorted(read - listed)


def _child(name, value):
    env = dict(os.environ)
    env[name] = value
    env["OMP_NUM_THREADS"] = env["MKL_NUM_THREADS"] = "4"           # four children at a time on the GPU box's CPU share: the CPU oracle of each keeps to its cores
    if name in ("SSV_DIST_BUCKETS", "SSV_DIST_BACKEND"):            # meaningful only with the collectives running
        env["SSV_DIST_FORCE"] = "1"
    nodes = [R18] + EXTRA.get(name, [])
    res = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", *nodes], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    return name, value, res.returncode, res.stdout[-2500:] + "\n" + res.stderr[-1000:]


@pytest.mark.gpu
def test_golden_steps_hold_under_every_switch():
    """One child interpreter per row of SWITCHES (the variables are read at import), four at a time (the box allows six processes on its GPU); every child must
    pass its golden bodies.  The failure message names each failing switch with the tail of its child's output."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(lambda nv: _child(*nv), SWITCHES))
    bad = [(n, v, out) for n, v, rc, out in results if rc != 0]
    assert not bad, "\n\n".join(f"=== {n}={v} ===\n{out}" for n, v, out in bad)
    assert len(results) == len(SWITCHES)
