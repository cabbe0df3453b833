#!/usr/bin/env python3
"""Diagnostic: does the library's 3x3 convolution give the same result for a batch as for its items one by one?
(It does, up to the kernel's rounding, at these sizes: the corruption tools/upsampler_debug.py shows needs an
activation larger than 4 GiB.)"""
import torch

torch.manual_seed(0)
C = 256
conv = torch.nn.Conv2d(C, C, 3, padding=1).cuda().eval()
for (h, w) in ((40, 48), (80, 96), (160, 192), (320, 384), (64, 64), (40, 64), (160, 256), (320, 512)):
    for N in (1, 2, 3, 4, 6, 8):
        x = torch.randn(N, C, h, w, device="cuda")
        with torch.no_grad():
            y = conv(x)
            ref = torch.cat([conv(x[i:i + 1]) for i in range(N)])
        err = (y - ref).abs().amax(dim=(1, 2, 3)).tolist()
        bad = [i for i, e in enumerate(err) if e > 1e-3]
        print(f"H x W {h}x{w} N={N}: max err per item {['%.1e' % e for e in err]} {'<-- WRONG items ' + str(bad) if bad else ''}", flush=True)
