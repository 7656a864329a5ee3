"""CPU: the C ABI refuses bad arguments instead of crashing ("returns 0 on success, negative ssv_status otherwise; never
throws across the ABI"), for EVERY entry point include/ssv_hip.h declares - once on the shipped library and once on the host-side
AddressSanitizer + UBSan build (`make -C self-supervised-vision_amd/csrc asan`: launch / validation / workspace-sizing code only,
no device code).  No GPU is needed: validation happens before any launch, and a launch that is reached fails with SSV_ERR_LAUNCH.

The calls run in a child process (a crash must fail the test, not the test session); the child loads the library with plain ctypes.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "self-supervised-vision_amd", "csrc")

_DRIVER = r'''
import ctypes as C, json, sys
lib_path, sig_path = sys.argv[1], sys.argv[2]
sigs = json.load(open(sig_path))
lib = C.CDLL(lib_path)
TYPES = {"void_p": C.c_void_p, "i32": C.c_int32, "i64": C.c_int64, "f32": C.c_float, "f64": C.c_double, "u64": C.c_uint64, "size": C.c_size_t,
         "int": C.c_int, "char_p": C.c_char_p}

class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "H", "W", "C", "K", "R", "S", "stride", "pad", "Ho", "Wo")]
class AugCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("brightness", "contrast", "saturation", "hue", "p_jitter", "p_gray", "p_flip", "scale_min", "scale_max", "ratio_min", "ratio_max")]
class BnGate(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "scale", "shift", "mask", "mean", "invstd", "psum_g", "psum_gx", "x2", "mean2", "invstd2", "psum_gx2")]
class BnDyin(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "coef")]
PTR = {"p_conv": C.POINTER(ConvDesc), "p_aug": C.POINTER(AugCfg), "p_gate": C.POINTER(BnGate), "p_dyin": C.POINTER(BnDyin), "p_f32": C.POINTER(C.c_float),
       "p_f64": C.POINTER(C.c_double), "p_i64": C.POINTER(C.c_int64)}
host = (C.c_float * 4096)()                      # a HOST buffer: fine for argument checks, never dereferenced by host code
hp = C.cast(host, C.c_void_p).value
hp = (hp + 255) // 256 * 256

def build(kind, mode):
    """mode 'null': every pointer NULL, sizes 1.  'desc': inconsistent descriptors, aligned non-null pointers.  'neg': negative sizes."""
    if kind in TYPES:
        t = TYPES[kind]
        if kind == "void_p":
            return t(None) if mode == "null" else t(hp)
        if kind in ("i32", "i64", "int", "size", "u64"):
            return t(1 if mode != "neg" else (0 if kind in ("size", "u64") else -3))
        return t(0.5)
    if kind == "p_conv":
        if mode == "null":
            return C.POINTER(ConvDesc)()
        d = ConvDesc(2, 8, 8, 32, 32, 3, 3, 1, 1, 5, 9) if mode == "desc" else ConvDesc(-1, 8, 8, 32, 32, 3, 3, 1, 1, 8, 8)   # Ho/Wo wrong | N < 0
        return C.pointer(d)
    if kind == "p_gate":
        return C.POINTER(BnGate)() if mode == "null" else C.pointer(BnGate())          # all-NULL members: an incomplete gate
    if kind == "p_dyin":
        return C.POINTER(BnDyin)() if mode == "null" else C.pointer(BnDyin())          # all-NULL members: an incomplete operand description
    if kind == "p_aug":
        return C.POINTER(AugCfg)() if mode == "null" else C.pointer(AugCfg())
    return PTR[kind]() if mode == "null" else C.cast(hp, PTR[kind])

out = {}
for name, (res, args) in sorted(sigs.items()):
    fn = getattr(lib, name)
    fn.restype = TYPES[res]
    fn.argtypes = [TYPES.get(a) or PTR[a] for a in args]
    got = []
    for mode in ("null", "desc", "neg"):
        keep = [build(a, mode) for a in args]
        r = fn(*keep)
        got.append(r.decode() if isinstance(r, bytes) else r)
        sys.stdout.write("")          # keep the interpreter honest about ordering under ASan
    out[name] = got
err = lib.ssv_last_error
err.restype = C.c_char_p
print("RESULT " + json.dumps({"calls": out, "last_error": err().decode()}))
'''

_KIND = {C.c_void_p: "void_p", C.c_int32: "i32", C.c_int64: "i64", C.c_float: "f32", C.c_double: "f64", C.c_uint64: "u64", C.c_size_t: "size",
         C.c_int: "int", C.c_char_p: "char_p"}


def _signatures():
    from ssv_amd import _lib
    ptrs = {C.POINTER(_lib.ConvDesc): "p_conv", C.POINTER(_lib.AugCfg): "p_aug", C.POINTER(_lib.BnGate): "p_gate", C.POINTER(_lib.BnDyin): "p_dyin", C.POINTER(C.c_float): "p_f32",
            C.POINTER(C.c_double): "p_f64", C.POINTER(C.c_int64): "p_i64"}
    kind = lambda t: _KIND.get(t) or ptrs[t]
    return {name: (kind(res), [kind(a) for a in args]) for name, (res, args) in _lib.SIGNATURES.items()}


def _drive(lib_path, tmp_path, env_extra=None):
    sig = tmp_path / "sigs.json"
    sig.write_text(json.dumps(_signatures()))
    drv = tmp_path / "driver.py"
    drv.write_text(_DRIVER)
    env = dict(os.environ, **(env_extra or {}))
    res = subprocess.run([sys.executable, str(drv), lib_path, str(sig)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, f"the ABI driver died (rc {res.returncode}):\n{res.stdout[-1500:]}\n{res.stderr[-4000:]}"
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):]), res.stderr


# entry points that legitimately succeed on some of the probes (pure host helpers, documented no-ops)
_HOST_HELPERS = {"ssv_version", "ssv_last_error", "ssv_source_sha16", "ssv_device_cus", "ssv_prof_enable", "ssv_prof_reset"}


def _check(result):
    sigs = _signatures()
    bad = []
    for name, got in result["calls"].items():
        res = sigs[name][0]
        if name in _HOST_HELPERS:
            continue
        if res in ("size", "i64"):                     # workspace / group-count helpers: any finite answer, NULL descriptor -> 0
            if got[0] != 0 and "p_conv" in sigs[name][1]:
                bad.append((name, "null descriptor must size to 0", got))
            continue
        # status-returning entry points: never SSV_OK on NULL pointers, and never anything but a declared status
        if not all(g in (0, -1, -2, -3) for g in got):
            bad.append((name, "undeclared status", got))
        if got[0] == 0 and any(a in ("void_p", "p_conv", "p_gate", "p_dyin", "p_aug", "p_f32", "p_f64", "p_i64") for a in sigs[name][1]):
            bad.append((name, "accepted NULL pointers", got))
        if "p_conv" in sigs[name][1] and got[1] != -1:
            bad.append((name, "accepted an inconsistent conv descriptor", got))
        if "p_conv" in sigs[name][1] and got[2] != -1:
            bad.append((name, "accepted a negative dimension", got))
    assert not bad, "\n".join(f"{n}: {why} {g}" for n, why, g in bad)
    assert result["last_error"], "ssv_last_error() must describe the last refusal"


def test_every_entry_point_refuses_bad_arguments(tmp_path):
    from ssv_amd import _lib
    result, _ = _drive(_lib.LIB_PATH, tmp_path)
    assert set(result["calls"]) == set(_lib.SIGNATURES)
    _check(result)


def test_every_entry_point_refuses_bad_arguments_under_asan_ubsan(tmp_path):
    lib = os.path.join(CSRC, "libssv_hip_asan.so")
    rc = subprocess.run(["make", "-C", CSRC, "-j4", "asan"], capture_output=True, text=True, timeout=1200)      # up to date -> a no-op
    assert rc.returncode == 0, rc.stderr[-3000:]
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("no shared ASan runtime in this toolchain")
    result, stderr = _drive(lib, tmp_path, {"LD_PRELOAD": rt, "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=1",
                                           "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    assert "ERROR: AddressSanitizer" not in stderr and "runtime error:" not in stderr, stderr[-4000:]
    _check(result)
