"""CPU oracle for the audio-driven avatar rendering hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker.  The product package
(``audio-motion-avatar_amd/``) never imports it and has no CPU fallback.

PARITY UNPINNED.  The reference (liubingqi7/audio-motion-avatar @ 2025-08-15) ships no tests, fixtures or
golden vectors, and every non-torch operator on the path lives in un-vendored third-party packages that
are absent from this image:

    smplx==0.1.28                (requirements.txt:12)       -> oracle/lbs.py
    diff_gaussian_rasterization  (unpinned, README.md:117)   -> oracle/raster_ref.c, oracle/rasterizer.py
    diffusers Attention          (unpinned, unlisted)        -> oracle/transformer.py
    pytorch3d>=0.7.8             (requirements.txt:10)       -> oracle/rotation.py, oracle/subdivide.py

Each module restates the published algorithm of its package and follows the
reference's own call sites line by line (cited per function).  torch operators that the reference calls
directly (F.grid_sample, nn.Linear, LayerNorm, GroupNorm, SDPA, MultiheadAttention, Conv3d) are used as
they are: on CPU they ARE the reference's arithmetic.  What pins the oracle instead of reference vectors:
the analytic known-answer tests in tests/test_oracle_*.py and the fp64 builds of the same code.
"""
