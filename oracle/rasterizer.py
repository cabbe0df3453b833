"""Gaussian tile rasterizer + the reference's splat wrappers, restated on CPU.

* ``rasterize_c``      : ctypes front end of oracle/raster_ref.c (float32 or float64 build).
* ``rasterize_torch``  : an independent brute-force torch restatement of the same algorithm (sequential over the
                         depth-sorted Gaussians, vectorised over pixels) used to cross-check the C code on small cases.
* ``render_one`` / ``render_batch`` / ``render_multi_view`` : src/models/renderer.py:431-569 line by line
  (camera setup, SCALE_BIAS / OPACITY_BIAS activations, colour clamp, rasterize, clamp(0,1), HWC stack).

diff_gaussian_rasterization is absent and unpinned (README.md:117-121): PARITY UNPINNED, see oracle/__init__.py.
Test infrastructure only.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

from . import camera

SCALE_BIAS = 3.9  # renderer.py:428
OPACITY_BIAS = 0.0  # renderer.py:429
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build():
    """Compile raster_ref.c (both precisions) with oracle/Makefile."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def _lib(dtype):
    name = "f64" if dtype == np.float64 else "f32"
    if name not in _LIBS:
        path = os.path.join(_HERE, "_build", f"liboracle_raster_{name}.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.oracle_rasterize.restype = ctypes.c_long
        _LIBS[name] = lib
    return _LIBS[name]


def rasterize_c(means3d, rotations, scales, opacities, colors, viewmatrix, projmatrix, tanfovx, tanfovy, bg, height,
                width, scale_modifier=1.0, antialiasing=False, dtype=np.float32):
    """One frame through raster_ref.c.  Inputs array-like (already activated, as renderer.py:557-566 passes them).

    Returns dict(color [3,H,W], alpha [H,W], inv_depth [H,W], radii [N] int32, instances int, unstable [H,W] u8:
    pixels with a blend decision within 1e-4 (relative) of a discontinuity of the algorithm, see raster_ref.c).
    """
    dtype = np.dtype(dtype).type
    lib = _lib(dtype)
    c_real = ctypes.c_double if dtype == np.float64 else ctypes.c_float
    arr = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(dtype))
    m, r, s, o, c = arr(means3d), arr(rotations), arr(scales), arr(opacities).reshape(-1), arr(colors)
    v, p, b = arr(viewmatrix).reshape(16), arr(projmatrix).reshape(16), arr(bg).reshape(3)
    N = m.shape[0]
    color = np.zeros((3, height, width), dtype)
    alpha = np.zeros((height, width), dtype)
    invd = np.zeros((height, width), dtype)
    radii = np.zeros(N, np.int32)
    unstable = np.zeros((height, width), np.uint8)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    n = lib.oracle_rasterize(
        ctypes.c_int(N), ctypes.c_int(height), ctypes.c_int(width), ptr(m), ptr(r), ptr(s), ptr(o), ptr(c), ptr(v),
        ptr(p), c_real(tanfovx), c_real(tanfovy), ptr(b), c_real(scale_modifier), ctypes.c_int(int(antialiasing)),
        ptr(color), ptr(alpha), ptr(invd), ptr(radii), ptr(unstable))
    if n < 0:
        raise MemoryError("oracle_rasterize: allocation failed")
    return dict(color=color, alpha=alpha, inv_depth=invd, radii=radii, instances=int(n), unstable=unstable)


def preprocess_torch(means3d, rotations, scales, opacities, view, proj, tanfovx, tanfovy, H, W, scale_modifier=1.0):
    """Per-Gaussian preprocess in torch.  view/proj: transposed [4,4] tensors."""
    dt = means3d.dtype
    N = means3d.shape[0]
    ones = torch.ones(N, 1, dtype=dt)
    ph = torch.cat([means3d, ones], 1) @ proj  # row-vector convention == column-major read of proj^T
    pv = (torch.cat([means3d, ones], 1) @ view)[:, :3]
    pw = 1.0 / (ph[:, 3] + 0.0000001)
    ppx, ppy = ph[:, 0] * pw, ph[:, 1] * pw
    r, x, y, z = rotations.unbind(1)
    Rm = torch.stack(
        [1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z),
         1 - 2 * (x * x + z * z), 2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x),
         1 - 2 * (x * x + y * y)], 1).view(N, 3, 3)
    S = torch.diag_embed(scale_modifier * scales)
    Sigma = Rm @ S @ S @ Rm.transpose(1, 2)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    tz = pv[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tx = torch.clamp(pv[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(pv[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).view(N, 2, 3)
    Wv = view[:3, :3].transpose(0, 1)  # rotation part of E
    Tm = J @ Wv
    cov = Tm @ Sigma @ Tm.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], 1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam))
    px = ((ppx + 1.0) * W - 1.0) * 0.5
    py = ((ppy + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    rx0 = torch.clamp(((px - radius) / 16).to(torch.int32), 0, gx)
    ry0 = torch.clamp(((py - radius) / 16).to(torch.int32), 0, gy)
    rx1 = torch.clamp(((px + radius + 15) / 16).to(torch.int32), 0, gx)
    ry1 = torch.clamp(((py + radius + 15) / 16).to(torch.int32), 0, gy)
    visible = (tz > 0.2) & (det != 0) & ((rx1 - rx0) * (ry1 - ry0) > 0)
    return dict(xy=torch.stack([px, py], 1), depth=tz, conic=conic, radius=radius, rect=(rx0, ry0, rx1, ry1),
                visible=visible, opacity=opacities.reshape(-1))


def rasterize_torch(means3d, rotations, scales, opacities, colors, view, proj, tanfovx, tanfovy, bg, H, W):
    """Brute-force restatement: Gaussians in global (depth, index) order, all pixels at once; a pixel only sees a
    Gaussian when its 16x16 tile lies inside the Gaussian's tile rectangle."""
    g = preprocess_torch(means3d, rotations, scales, opacities, view, proj, tanfovx, tanfovy, H, W)
    dt = means3d.dtype
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    tx, ty = xs // 16, ys // 16
    xf, yf = xs.to(dt), ys.to(dt)
    T = torch.ones(H, W, dtype=dt)
    C = torch.zeros(3, H, W, dtype=dt)
    D = torch.zeros(H, W, dtype=dt)
    done = torch.zeros(H, W, dtype=torch.bool)
    ids = torch.nonzero(g["visible"]).reshape(-1)
    order = ids[torch.argsort(g["depth"][ids], stable=True)]
    rx0, ry0, rx1, ry1 = g["rect"]
    for i in order.tolist():
        in_rect = (tx >= rx0[i]) & (tx < rx1[i]) & (ty >= ry0[i]) & (ty < ry1[i])
        dx, dy = g["xy"][i, 0] - xf, g["xy"][i, 1] - yf
        A, B, Cc = g["conic"][i]
        power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
        alpha = torch.clamp(g["opacity"][i] * torch.exp(power), max=0.99)
        ok = in_rect & ~done & (power <= 0) & (alpha >= 1.0 / 255.0)
        test_T = T * (1 - alpha)
        newly_done = ok & (test_T < 0.0001)
        done |= newly_done
        ok &= ~newly_done
        w = torch.where(ok, alpha * T, torch.zeros_like(T))
        C += colors[i].view(3, 1, 1) * w
        D += w / g["depth"][i]
        T = torch.where(ok, test_T, T)
    color = C + T * torch.as_tensor(bg, dtype=dt).view(3, 1, 1)
    radii = torch.where(g["visible"], g["radius"], torch.zeros_like(g["radius"])).to(torch.int32)
    return dict(color=color, alpha=1 - T, inv_depth=D, radii=radii)


def render_one(xyzs, rots, scales, opacities, colors, K, E, image_size, bg_color=None, debug=False,
               dtype=np.float32, full=False):
    """renderer.py:481-569 (rgb=True branch; the SH branch is dead code)."""
    height, width = image_size[0], image_size[1]
    view, proj, tanfovx, tanfovy, _ = camera.camera_setup(K, E, height, width)
    if bg_color is None:
        bg_color = [1, 1, 1]
    scales = torch.min(torch.exp(scales - SCALE_BIAS), torch.tensor(0.1, dtype=scales.dtype))
    opacities = torch.sigmoid(opacities - OPACITY_BIAS)
    if debug:
        scales = torch.ones_like(scales) * 0.01
        opacities = torch.ones_like(opacities) * 0.1
    colors_precomp = torch.clamp(colors, 0.0, 1.0)
    out = rasterize_c(xyzs.numpy(), rots.numpy(), scales.numpy(), opacities.numpy(), colors_precomp.numpy(),
                      view.numpy(), proj.numpy(), tanfovx, tanfovy, bg_color, height, width, dtype=dtype)
    tdt = torch.float64 if np.dtype(dtype) == np.float64 else torch.float32
    img = torch.from_numpy(out["color"]).to(tdt).clamp(0, 1)
    if full:
        return img, out
    return img


def render_batch(gaussians, K, E, image_size, bg_color=None, debug=False, dtype=np.float32, full=False):
    """renderer.py:447-479 -> [B,T,H,W,3] (with full=True also alpha [B,T,H,W] and the unstable-pixel mask)."""
    B, T = E.shape[0], E.shape[1]
    tdt = torch.float64 if np.dtype(dtype) == np.float64 else torch.float32
    E_flat, K_flat = E.reshape(-1, 4, 4).to(tdt), K.reshape(-1, 3, 3).to(tdt)
    xyzs = gaussians["xyz"].reshape(B * T, -1, 3).to(tdt)
    rots = gaussians["rot"].reshape(B * T, -1, 4).to(tdt)
    scales = gaussians["scale"].reshape(B * T, -1, 3).to(tdt)
    opac = gaussians["opacity"].reshape(B * T, -1, 1).to(tdt)
    cols = gaussians["color"].reshape(B * T, -1, 3).to(tdt)
    imgs, alphas, unstable = [], [], []
    for i in range(B * T):
        img, out = render_one(xyzs[i], rots[i], scales[i], opac[i], cols[i], K_flat[i], E_flat[i], image_size,
                              bg_color, debug, dtype=dtype, full=True)
        imgs.append(img.permute(1, 2, 0))
        alphas.append(torch.from_numpy(out["alpha"]))
        unstable.append(torch.from_numpy(out["unstable"]))
    images = torch.stack(imgs).reshape(B, T, image_size[0], image_size[1], 3)
    if full:
        hw = (B, T, image_size[0], image_size[1])
        return images, torch.stack(alphas).reshape(hw), torch.stack(unstable).reshape(hw).bool()
    return images


def render_multi_view(gaussians, K, E, image_size, bg_color=None, debug=False, dtype=np.float32):
    """renderer.py:431-445."""
    B, T = E.shape[0], E.shape[1]
    ex = {k: v.unsqueeze(1).expand(-1, T, -1, -1) for k, v in gaussians.items()}
    ex = {k: v.reshape(B * T, -1, v.shape[-1]) for k, v in ex.items()}
    return render_batch(ex, K, E, image_size, bg_color, debug, dtype=dtype).reshape(B, T, image_size[0],
                                                                                    image_size[1], 3)
