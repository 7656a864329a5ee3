"""CPU: `python bench.py --gpus N` / `SSV_GPUS=N python main.py` start their N ranks themselves (ssv_amd/launch.py) - and the parent that
does so never imports torch, so it can never have touched HIP when it starts GPU workers."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_FAKE_TORCH = '''import os, sys
sys.stderr.write("FAKE_TORCH_IMPORTED pid=%d argv=%r\\n" % (os.getpid(), sys.argv))
raise ImportError("fake torch: this process imported torch")
'''


def _no_gpu_env(**extra):
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", **extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "SSV_GPUS"):
        if k not in extra:
            env.pop(k, None)
    return env


@pytest.mark.parametrize("script,argv,extra", [("bench.py", ["--gpus", "2", "--steps", "1", "--warmup", "0"], {}),
                                               ("main.py", ["-c", "x.yaml", "-m", "resnet18", "-a", "simclr", "-t", "train"], {"SSV_GPUS": "2"})])
def test_parent_starts_the_launcher_without_importing_torch(tmp_path, script, argv, extra):
    """A poisoned `torch` package first on PYTHONPATH reports who imports it: only the CHILD (`-m torch.distributed.run ... <script>`)
    may - the parent reaches the spawn with the standard library alone and hands on the child's exit code."""
    fake = tmp_path / "fake" / "torch"
    fake.mkdir(parents=True)
    (fake / "__init__.py").write_text(_FAKE_TORCH)
    env = _no_gpu_env(PYTHONPATH=str(tmp_path / "fake"), **extra)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, script), *argv], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out, err = p.communicate(timeout=120)
    marks = [ln for ln in err.splitlines() if ln.startswith("FAKE_TORCH_IMPORTED")]
    assert marks, err[-2000:]                                           # the launcher child was started and tried to import torch
    assert all(f"pid={p.pid} " not in ln for ln in marks), marks       # ... the parent never did
    assert all("torch.distributed.run" in ln or "torch/distributed/run" in ln or "-m" in ln for ln in marks), marks
    assert p.returncode != 0                                            # the child's failure is the parent's exit code


def test_spawn_ranks_refuses_once_torch_is_imported_and_builds_the_documented_command():
    import torch  # noqa: F401
    from ssv_amd import launch
    with pytest.raises(RuntimeError, match="before torch is imported"):
        launch.spawn_ranks("bench.py", ["--gpus", "2"], 2)
    cmd = launch.rank_command("/x/bench.py", ["--gpus", "4", "--steps", "3"], 4, port=29511)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1", "--master-port", "29511",
                       "/x/bench.py", "--gpus", "4", "--steps", "3"]
    assert launch.gpus_flag(["--steps", "3", "--gpus=8"]) == 8 and launch.gpus_flag(["--gpus", "2"]) == 2 and launch.gpus_flag([]) == 1


def test_inside_a_process_group_nothing_is_spawned(monkeypatch):
    from ssv_amd import launch
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    assert launch.maybe_spawn_ranks("bench.py", [], 2) is None          # the outer launcher's rank runs in-process (would sys.exit otherwise)
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.delenv("RANK")
    assert launch.maybe_spawn_ranks("bench.py", [], 1) is None


def test_bench_gpus_2_reaches_two_ranks_of_a_process_group():
    """Without a GPU the two ranks stop at bench.py's device check - after joining a world of TWO: the plain `python bench.py --gpus 2`
    invocation (the driver's form) no longer dies on the world-size check (round 3: bench.py:326-329)."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--prof-steps", "0"],
                         env=_no_gpu_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode != 0
    assert "no HIP device visible (rank 0 of 2)" in res.stderr and "no HIP device visible (rank 1 of 2)" in res.stderr, res.stderr[-3000:]
    assert "the process group has" not in res.stderr


def test_bench_replays_counters_only_with_their_build_identity(tmp_path):
    """bench.py's reader of the committed counter files: the build identity (src_sha16 = ssv_source_sha16() of the profiled library) comes back with the data, from the
    JSON files of tools/pmc_*.py and from the last row of tools/bench_conv.py's CSV; a file without it, or a missing file, yields None - the caller then prints
    `traffic: null` / `counters_stale: true` instead of another build's numbers.  The committed round-4 files carry an identity."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    (tmp_path / "a.json").write_text(json.dumps({"src_sha16": "0123456789abcdef", "per_step_gb": {"conv_fwd": {"fetch": 1.0, "write": 2.0}}}))
    (tmp_path / "old.json").write_text(json.dumps({"per_step_gb": {}}))                       # a round-3 file: no identity
    (tmp_path / "l.csv").write_text("layer,count,fwd_GB\nTOTAL fwd,,61.5\nBUILD src_sha16=0123456789abcdef lib_sha16=ffff,batch 512\n")
    (tmp_path / "l_old.csv").write_text("layer,count,fwd_GB\nTOTAL fwd,,61.5\n")
    data, src, rel = bench.read_committed_counters(str(tmp_path), "a.json")
    assert src == "0123456789abcdef" and data["per_step_gb"]["conv_fwd"]["write"] == 2.0 and rel.endswith("a.json")
    assert bench.read_committed_counters(str(tmp_path), "old.json")[1] is None
    rows, src, _ = bench.read_committed_counters(str(tmp_path), "l.csv")
    assert src == "0123456789abcdef" and [r for r in rows if r["layer"].startswith("TOTAL")][0]["fwd_GB"] == "61.5"
    assert bench.read_committed_counters(str(tmp_path), "l_old.csv")[1] is None
    assert bench.read_committed_counters(str(tmp_path), "missing.json") == (None, None, None)
    prof = os.path.join(ROOT, "profiles")
    for fname in (bench.PMC_FILES["simclr"] % 512, bench.PMC_MFMA_FILES["simclr"] % 512, bench.CONV_LAYER_FILE % 512, bench.PMC_FILES["dino"] % 128, bench.PMC_MFMA_FILES["dino"] % 128):
        _, src, rel = bench.read_committed_counters(prof, fname)
        assert rel is not None and src and len(src) == 16, fname


def test_committed_counters_were_measured_on_the_current_sources():
    """The build identity is the sha256 over the kernel sources in link order + the two headers + the compile flags (csrc/Makefile: SRC_SHA).  When the sources have moved on since the
    counters under profiles/ were measured, bench.py prints `counters_stale: true` - this test then SKIPS with the reason (a reminder to re-run tools/profile_step.sh
    and tools/bench_conv.py), it does not fail: stale counters are withheld, never wrong."""
    sys.path.insert(0, ROOT)
    import bench
    csrc = os.path.join(ROOT, "self-supervised-vision_amd", "csrc")
    # the Makefile's own recipe (sources in link order + the two headers + the compile flags): `make print-src-sha`
    now = subprocess.run(["make", "-s", "-C", csrc, "print-src-sha"], capture_output=True, text=True, check=True).stdout.strip()
    assert len(now) == 16
    from ssv_amd import _lib
    assert _lib.source_sha16() == now, "libssv_hip.so was not rebuilt after the last source change (run make -C self-supervised-vision_amd/csrc)"
    stale = [f for f in (bench.PMC_FILES["simclr"] % 512, bench.PMC_MFMA_FILES["simclr"] % 512, bench.CONV_LAYER_FILE % 512)
             if bench.read_committed_counters(os.path.join(ROOT, "profiles"), f)[1] != now]
    if stale:
        pytest.skip(f"counters under profiles/ were measured on another build than the current sources ({now}): {stale} - bench.py will print counters_stale")


def test_a_stopped_parent_takes_its_ranks_with_it(tmp_path):
    """SIGTERM to the waiting parent (``timeout -k`` around ``python bench.py --gpus N``, a harness watchdog) reaches the launcher's whole process group: the
    child runs in its own session and the parent forwards TERM, then KILL - no rank is left holding the GPUs and the rendezvous port."""
    import signal
    import time
    pidfile = tmp_path / "pids.txt"
    grandchild = tmp_path / "rank.py"
    grandchild.write_text("import os, sys, time\nopen(sys.argv[1], 'a').write(str(os.getpid()) + '\\n')\ntime.sleep(120)\n")
    child = tmp_path / "launcher.py"         # stands in for torch.distributed.run: starts two 'ranks' and waits for them
    child.write_text("import subprocess, sys\nps = [subprocess.Popen([sys.executable, sys.argv[1], sys.argv[2]]) for _ in range(2)]\n[p.wait() for p in ps]\n")
    parent = tmp_path / "parent.py"
    parent.write_text(f"import sys\nsys.path.insert(0, {ROOT!r})\nfrom ssv_amd import launch\n"
                      f"sys.exit(launch.run_and_forward_signals([sys.executable, {str(child)!r}, {str(grandchild)!r}, {str(pidfile)!r}], grace=2.0))\n")
    p = subprocess.Popen([sys.executable, str(parent)])
    for _ in range(100):
        if pidfile.exists() and len(pidfile.read_text().split()) == 2:
            break
        time.sleep(0.1)
    pids = [int(x) for x in pidfile.read_text().split()]
    assert len(pids) == 2
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    time.sleep(0.3)
    for pid in pids:
        alive = True
        try:
            os.kill(pid, 0)
            with open(f"/proc/{pid}/stat") as fh:
                alive = fh.read().split()[2] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, f"rank {pid} outlived its parent"
    # an undisturbed run hands back the child's own exit code
    ok = tmp_path / "ok.py"
    ok.write_text("import sys\nsys.exit(7)\n")
    from ssv_amd import launch
    assert launch.run_and_forward_signals([sys.executable, str(ok)]) == 7
