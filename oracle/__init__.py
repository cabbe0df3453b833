"""CPU oracle for the audio-driven avatar rendering hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker.  The product package
(``audio-motion-avatar_amd/``) never imports it and has no CPU fallback.

PINNED WHERE THE REFERENCE CAN RUN, UNPINNED ELSEWHERE.  The reference (liubingqi7/audio-motion-avatar @ 2025-08-15)
ships no tests, fixtures or golden vectors, and every non-torch operator on the path lives in un-vendored third-party
packages that are absent from this image:

    smplx==0.1.28                (requirements.txt:12)       -> oracle/lbs.py                    PARITY UNPINNED
    diff_gaussian_rasterization  (unpinned, README.md:117)   -> oracle/raster_ref.c, rasterizer.py  PARITY UNPINNED
    diffusers Attention          (unpinned, unlisted)        -> oracle/transformer.py (attention)  PARITY UNPINNED
    pytorch3d>=0.7.8             (requirements.txt:10)       -> oracle/rotation.py, subdivide.py  PARITY UNPINNED
    spconv, torch_scatter        (unlisted)                  -> oracle/ptv3.py (subm_conv3d), triplane_net.py (scatter)
                                                                                                  PARITY UNPINNED

Everything AROUND those operators is pinned by vectors produced by running the reference's own Python in the build
container (tests/golden/make_reference_golden.py -> tests/golden/ref_*.npz; tier 1 = reference code as shipped, tier
2 = reference constructors / forwards with the absent operator injected): camera, temporal reducers, FeedForward /
GEGLU, triplane sampling / construct_gaussians / upsampler, the autoregressive audio net, the SMPL-X decoder, the
stage-1 encoder parts and the PTv3 serialisation codes and network.  DESIGN.md section 2 has the table.  Each
module restates the published algorithm of its package and follows the reference's own call
sites line by line (cited per function).  torch operators that the reference calls directly (F.grid_sample,
nn.Linear, LayerNorm, GroupNorm, SDPA, MultiheadAttention, Conv3d) are used as they are: on CPU they ARE the
reference's arithmetic.  The unpinned operators are held by analytic known-answer tests (tests/test_oracle_*.py),
independent second restatements and fp64 builds of the same code.
"""
