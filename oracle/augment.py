"""Oracle (test infrastructure): the two-view augmentation chain of configs/simclr.yaml:13-29.

The reference builds it from torchvision transforms (utils/augmentations.py:113-144) and runs it per
sample, twice, in DataLoader workers (utils/data_utils.py:68-73).  The arithmetic lives in THIRD-PARTY code
that is neither vendored in /root/reference nor installed here:
    torchvision==0.9.1   (requirements.txt:6)  transforms.{RandomApply,ColorJitter,RandomGrayscale,
                          RandomResizedCrop,RandomHorizontalFlip,ToTensor,Normalize} over functional_pil.py
    Pillow==8.3.1        (requirements.txt:8)  ImageEnhance / Image.convert / Image.resize / Image.blend
so this piece is PARITY UNPINNED against torchvision.  What is pinned instead:
  * ``view_pil``   runs the published torchvision-0.9.1 PIL recipe on the Pillow that IS installed (12.x):
                   every deterministic op given explicit parameters (factors, op order, crop box, flags);
  * ``view_numpy`` restates Pillow's integer arithmetic (Blend.c, Convert.c rgb2hsv/hsv2rgb/L, Resample.c
                   bilinear with 22-bit fixed-point coefficients and uint8 between the two passes);
                   tests/test_augment_cpu.py checks it bit-for-bit against ``view_pil``; the HIP kernel mirrors it;
  * ``draw_params`` restates the parameter DISTRIBUTIONS of torchvision 0.9.1 (ColorJitter.get_params,
                   RandomResizedCrop.get_params, p = .8 / .2 / .5) on a counter-based Philox4x32-10 stream keyed by
                   (seed, step, sample, view) - the reference's own stream (torch RNG inside 4 worker processes)
                   is not reproducible anywhere, so only the distribution can match.
Parameter record (16 float32): [0] jitter on/off, [1..4] op order (0 brightness, 1 contrast, 2 saturation,
3 hue), [5] brightness, [6] contrast, [7] saturation, [8] hue, [9] gray on/off, [10..13] crop top, left,
height, width, [14] flip on/off, [15] unused.
"""
import math

import numpy as np

NPARAM = 16
_PRECISION_BITS = 32 - 8 - 2


# ------------------------------------------------------------------------------------------- PIL recipe
def view_pil(img_u8, p, out_hw, mean, std):
    """img_u8: [H,W,3] uint8.  Returns float32 [3,Ho,Wo] exactly as the torchvision PIL pipeline would, given params."""
    from PIL import Image, ImageEnhance
    img = Image.fromarray(np.ascontiguousarray(img_u8), "RGB")
    if p[0] >= 0.5:
        for op in (int(p[1]), int(p[2]), int(p[3]), int(p[4])):
            if op == 0:
                img = ImageEnhance.Brightness(img).enhance(float(np.float32(p[5])))
            elif op == 1:
                img = ImageEnhance.Contrast(img).enhance(float(np.float32(p[6])))
            elif op == 2:
                img = ImageEnhance.Color(img).enhance(float(np.float32(p[7])))
            else:   # functional_pil.adjust_hue: uint8 wrap-around shift of the H channel
                h, s, v = img.convert("HSV").split()
                shift = int(float(np.float32(p[8])) * 255) % 256          # np.uint8(hue_factor * 255) of numpy 1.19
                nh = (np.array(h, dtype=np.uint8).astype(np.int32) + shift) % 256
                img = Image.merge("HSV", (Image.fromarray(nh.astype(np.uint8), "L"), s, v)).convert("RGB")
    if p[9] >= 0.5:
        g = img.convert("L")
        img = Image.merge("RGB", (g, g, g))
    top, left, ch, cw = int(p[10]), int(p[11]), int(p[12]), int(p[13])
    img = img.crop((left, top, left + cw, top + ch)).resize((out_hw[1], out_hw[0]), Image.BILINEAR)
    if p[14] >= 0.5:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    t = np.asarray(img, dtype=np.uint8).astype(np.float32) / np.float32(255)            # ToTensor
    t = (t - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)                # Normalize
    return np.ascontiguousarray(t.transpose(2, 0, 1))


def center_view_pil(img_u8, out_hw, mean, std):
    """The 'img' entry of the batch: CenterCrop -> ToTensor -> Normalize (configs/simclr.yaml:24-29)."""
    h, w = img_u8.shape[:2]
    top, left = int(round((h - out_hw[0]) / 2.0)), int(round((w - out_hw[1]) / 2.0))
    t = img_u8[top:top + out_hw[0], left:left + out_hw[1]].astype(np.float32) / np.float32(255)
    t = (t - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(t.transpose(2, 0, 1))


# ------------------------------------------------------------------------------------------- numpy restatement
def _gray_l(rgb):
    """Image.convert('L'): ITU-R 601-2 luma in 16.16 fixed point (Convert.c rgb2l)."""
    r, g, b = (rgb[..., k].astype(np.int64) for k in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.int64)


def _blend(degenerate, img, factor):
    """Image.blend(degenerate, img, factor) (Blend.c): float32 lerp, truncation, clip only when extrapolating."""
    a = np.float32(factor)
    d, i = degenerate.astype(np.int32), img.astype(np.int32)
    t = (d.astype(np.float32) + a * (i - d).astype(np.float32)).astype(np.float32)
    if 0.0 <= float(a) <= 1.0:
        return t.astype(np.int32).astype(np.uint8)            # (UINT8) cast of a value already in [0,255]
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def _rgb2hsv(rgb):
    """Convert.c rgb2hsv_row (float h/s, double intermediates where C promotes)."""
    r, g, b = (rgb[..., k].astype(np.int32) for k in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    safe = np.where(cr == 0, np.float32(1), cr)
    s = (cr / np.where(maxc == 0, 1, maxc).astype(np.float32)).astype(np.float32)
    rc = ((maxc - r).astype(np.float32) / safe).astype(np.float32)
    gc = ((maxc - g).astype(np.float32) / safe).astype(np.float32)
    bc = ((maxc - b).astype(np.float32) / safe).astype(np.float32)
    h = np.where(r == maxc, (bc - gc).astype(np.float32),
                 np.where(g == maxc, (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32),
                          (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32)))
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    flat = maxc == minc
    return np.stack([np.where(flat, 0, uh), np.where(flat, 0, us), maxc], -1).astype(np.uint8)


def _c_round(x):
    """C round(): half away from zero."""
    return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int64)


def _hsv2rgb(hsv):
    """Convert.c hsv2rgb."""
    h, s, v = (hsv[..., k].astype(np.float32) for k in range(3))
    h6 = h.astype(np.float64) * 6.0 / 255.0
    i = np.floor(h6).astype(np.float32)
    f = (h6 - i.astype(np.float64)).astype(np.float32)
    fs = (s.astype(np.float64) / 255.0).astype(np.float32)
    v64, fs64, f64 = v.astype(np.float64), fs.astype(np.float64), f.astype(np.float64)
    p = np.clip(_c_round(v64 * (1.0 - fs64)), 0, 255)
    q = np.clip(_c_round(v64 * (1.0 - fs64 * f64)), 0, 255)
    t = np.clip(_c_round(v64 * (1.0 - fs64 * (1.0 - f64))), 0, 255)
    vi = hsv[..., 2].astype(np.int64)
    sel = i.astype(np.int64) % 6
    r = np.choose(sel, [vi, q, p, p, t, vi])
    g = np.choose(sel, [t, vi, vi, q, p, p])
    b = np.choose(sel, [p, p, t, vi, vi, q])
    gray = hsv[..., 1] == 0
    return np.stack([np.where(gray, vi, r), np.where(gray, vi, g), np.where(gray, vi, b)], -1).astype(np.uint8)


def color_ops_numpy(img, p):
    """ColorJitter (in the drawn order) and RandomGrayscale on a uint8 [H,W,3] image."""
    if p[0] >= 0.5:
        for op in (int(p[1]), int(p[2]), int(p[3]), int(p[4])):
            if op == 0:
                img = _blend(np.zeros_like(img), img, p[5])
            elif op == 1:
                l = _gray_l(img)
                mean = int(l.sum() / l.size + 0.5)
                img = _blend(np.full_like(img, mean), img, p[6])
            elif op == 2:
                img = _blend(np.repeat(_gray_l(img)[..., None], 3, -1).astype(np.uint8), img, p[7])
            else:
                hsv = _rgb2hsv(img)
                shift = int(float(np.float32(p[8])) * 255) % 256
                hsv[..., 0] = ((hsv[..., 0].astype(np.int32) + shift) % 256).astype(np.uint8)
                img = _hsv2rgb(hsv)
    if p[9] >= 0.5:
        img = np.repeat(_gray_l(img)[..., None], 3, -1).astype(np.uint8)
    return img


def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs for the bilinear (triangle) filter + normalize_coeffs_8bpc.
    Returns (xmin[out], count[out], kk[out][kmax] int32 fixed point)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmins, counts, kk = np.zeros(out_size, np.int32), np.zeros(out_size, np.int32), np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        k = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - a if a < 1.0 else 0.0
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        xmins[xx], counts[xx] = xmin, xmax
        for x in range(xmax):
            v = k[x] * (1 << _PRECISION_BITS)
            kk[xx, x] = int(v - 0.5) if k[x] < 0 else int(v + 0.5)
    return xmins, counts, kk


def _clip8(v):
    return np.clip(v >> _PRECISION_BITS, 0, 255)


def resize_bilinear_numpy(img, out_hw):
    """Image.resize(BILINEAR) for uint8: horizontal pass, uint8, then vertical pass (Resample.c ImagingResampleInner)."""
    h, w = img.shape[:2]
    ho, wo = out_hw
    x = img.astype(np.int64)
    if wo != w:
        xmins, counts, kk = resample_coeffs(w, wo)
        out = np.zeros((h, wo, 3), np.int64)
        for xx in range(wo):
            acc = np.full((h, 3), 1 << (_PRECISION_BITS - 1), np.int64)
            for t in range(counts[xx]):
                acc += x[:, xmins[xx] + t, :] * int(kk[xx, t])
            out[:, xx, :] = _clip8(acc)
        x = out
    if ho != h:
        ymins, counts, kk = resample_coeffs(h, ho)
        out = np.zeros((ho, x.shape[1], 3), np.int64)
        for yy in range(ho):
            acc = np.full((x.shape[1], 3), 1 << (_PRECISION_BITS - 1), np.int64)
            for t in range(counts[yy]):
                acc += x[ymins[yy] + t] * int(kk[yy, t])
            out[yy] = _clip8(acc)
        x = out
    return x.astype(np.uint8)


def view_numpy(img_u8, p, out_hw, mean, std):
    img = color_ops_numpy(np.ascontiguousarray(img_u8), p)
    top, left, ch, cw = int(p[10]), int(p[11]), int(p[12]), int(p[13])
    img = resize_bilinear_numpy(img[top:top + ch, left:left + cw], out_hw)
    if p[14] >= 0.5:
        img = img[:, ::-1]
    t = img.astype(np.float32) / np.float32(255)
    t = (t - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(t.transpose(2, 0, 1))


# ------------------------------------------------------------------------------------------- parameter stream
_PHILOX_M0, _PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
_PHILOX_W0, _PHILOX_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32(counter, key):
    """Philox4x32-10 (Salmon et al. 2011).  counter: 4 uint32, key: 2 uint32 -> 4 uint32."""
    c = [int(v) & 0xFFFFFFFF for v in counter]
    k = [int(v) & 0xFFFFFFFF for v in key]
    for _ in range(10):
        p0, p1 = _PHILOX_M0 * c[0], _PHILOX_M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + _PHILOX_W0) & 0xFFFFFFFF, (k[1] + _PHILOX_W1) & 0xFFFFFFFF]
    return c


class _Stream:
    """Uniform doubles in [0,1) from 32-bit words of Philox blocks keyed by (seed) and counted by (sample, view, step, block)."""

    def __init__(self, seed, step, sample, view):
        self.key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        self.ctr = [sample & 0xFFFFFFFF, ((sample >> 32) & 0xFFFF) | ((view & 0xFFFF) << 16), step & 0xFFFFFFFF, 0]
        self.buf = []

    def u32(self):
        if not self.buf:
            self.buf = philox4x32(self.ctr, self.key)
            self.ctr[3] += 1
        return self.buf.pop(0)

    def uniform(self):
        return self.u32() * (1.0 / 4294967296.0)


def draw_params(seed, step, sample, view, hs, ws, cfg=None):
    """One parameter record; distributions of torchvision 0.9.1 for the chain in configs/simclr.yaml:13-22."""
    cfg = cfg or {}
    b, c, s, hu = cfg.get("brightness", 0.4), cfg.get("contrast", 0.4), cfg.get("saturation", 0.4), cfg.get("hue", 0.1)
    p_jit, p_gray, p_flip = cfg.get("apply_prob", 0.8), cfg.get("gray_prob", 0.2), cfg.get("flip_prob", 0.5)
    smin, smax = cfg.get("scale", (0.2, 1.0))
    rmin, rmax = cfg.get("ratio", (3.0 / 4.0, 4.0 / 3.0))
    st = _Stream(seed, step, sample, view)
    p = np.zeros(NPARAM, np.float32)
    p[0] = 1.0 if st.uniform() < p_jit else 0.0
    order = [0, 1, 2, 3]
    for i in range(3, 0, -1):                                   # Fisher-Yates = a uniform random permutation (torch.randperm(4))
        j = int(st.uniform() * (i + 1))
        order[i], order[j] = order[j], order[i]
    p[1:5] = order
    p[5] = np.float32(max(0.0, 1 - b) + st.uniform() * (1 + b - max(0.0, 1 - b)))
    p[6] = np.float32(max(0.0, 1 - c) + st.uniform() * (1 + c - max(0.0, 1 - c)))
    p[7] = np.float32(max(0.0, 1 - s) + st.uniform() * (1 + s - max(0.0, 1 - s)))
    p[8] = np.float32(-hu + st.uniform() * 2 * hu)
    p[9] = 1.0 if st.uniform() < p_gray else 0.0
    area = float(hs * ws)
    done = False
    for _ in range(10):                                         # RandomResizedCrop.get_params
        target = area * (smin + st.uniform() * (smax - smin))
        ratio = math.exp(math.log(rmin) + st.uniform() * (math.log(rmax) - math.log(rmin)))
        w = int(math.floor(math.sqrt(target * ratio) + 0.5))
        h = int(math.floor(math.sqrt(target / ratio) + 0.5))
        u_i, u_j = st.uniform(), st.uniform()
        if 0 < w <= ws and 0 < h <= hs:
            top, left = int(u_i * (hs - h + 1)), int(u_j * (ws - w + 1))
            done = True
            break
    if not done:                                                # fallback: central crop clamped to the ratio range
        in_ratio = ws / hs
        if in_ratio < rmin:
            w, h = ws, int(math.floor(ws / rmin + 0.5))
        elif in_ratio > rmax:
            h, w = hs, int(math.floor(hs * rmax + 0.5))
        else:
            w, h = ws, hs
        top, left = (hs - h) // 2, (ws - w) // 2
    p[10:14] = (top, left, h, w)
    p[14] = 1.0 if st.uniform() < p_flip else 0.0
    return p
