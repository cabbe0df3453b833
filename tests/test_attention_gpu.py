"""GPU parity of the MFMA self-attention kernel (through the C ABI; default form: fp16 x 2 split operands, DESIGN.md
section 4.4) against torch SDPA in fp64 on CPU (the reference's diffusers Attention calls the fp32 one,
transformers.py:329-336).  Tolerance 2e-5 absolute on O(1) outputs: far above the kernel's 1.1e-7, and what the
exact-product fp32 kernel (AMAV_ATTN=f32) was held to."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def reference(q, k, v, heads):
    B, S, HD = q.shape
    sp = lambda t: t.view(B, S, heads, HD // heads).transpose(1, 2).double()
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v))
    return o.transpose(1, 2).reshape(B, S, HD)


@pytest.mark.parametrize("B,S,H", [(1, 128, 1), (2, 404, 2), (1, 1000, 8), (1, 33, 3), (1, 6304, 8)])
def test_selfattn_matches_sdpa(B, S, H):
    from audio_motion_avatar_amd import ops

    g = torch.Generator().manual_seed(S)
    q, k, v = (torch.randn(B, S, H * 64, generator=g) for _ in range(3))
    q[0, 0] *= 6.0  # one peaked row: exercises the running-max rescale
    out = ops.selfattn(q.cuda(), k.cuda(), v.cuda(), H).cpu()
    ref = reference(q, k, v, H)
    assert (out.double() - ref).abs().max() <= 2e-5


def test_selfattn_reads_a_fused_qkv_buffer_in_place():
    from audio_motion_avatar_amd import ops

    g = torch.Generator().manual_seed(1)
    B, S, H = 2, 300, 4
    qkv = torch.randn(B, S, 3 * H * 64, generator=g).cuda()
    i = H * 64
    out = ops.selfattn(qkv[..., :i], qkv[..., i:2 * i], qkv[..., 2 * i:], H).cpu()
    ref = reference(qkv[..., :i].cpu().contiguous(), qkv[..., i:2 * i].cpu().contiguous(),
                    qkv[..., 2 * i:].cpu().contiguous(), H)
    assert (out.double() - ref).abs().max() <= 2e-5


def test_scale_argument_and_asymmetric_values():
    """A = I style check with asymmetric V: a transposed output map would not survive this."""
    from audio_motion_avatar_amd import ops

    S, H = 64, 1
    q = torch.zeros(1, S, 64)
    k = torch.zeros(1, S, 64)
    idx = torch.arange(S)
    q[0, idx, idx % 64] = 30.0          # query i attends (almost) only to key i
    k[0, idx, idx % 64] = 30.0
    v = torch.arange(S * 64, dtype=torch.float32).view(1, S, 64) / 100.0
    out = ops.selfattn(q.cuda(), k.cuda(), v.cuda(), H, scale=1.0).cpu()
    assert (out - v).abs().max() < 1e-4


def test_fused_geglu_matches_torch():
    """ops.geglu == hidden * F.gelu(gate) of transformers.py:484-508 (exact-erf GELU)."""
    import torch.nn.functional as F

    from audio_motion_avatar_amd import ops

    g = torch.Generator().manual_seed(0)
    x = (torch.randn(3, 257, 2 * 2048, generator=g) * 2.0).cuda()
    got = ops.geglu(x)
    h, gate = x.chunk(2, dim=-1)
    want = h * F.gelu(gate)
    want64 = (h.double() * F.gelu(gate.double())).float()
    assert got.shape == (3, 257, 2048)
    assert torch.allclose(got, want, rtol=2e-6, atol=2e-6) and torch.allclose(got, want64, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("dim", [256, 512])
def test_fused_residual_adds_and_layernorm(dim):
    """ops.add_layernorm == the block's `attn1 + h`, `attn2 + h` (one row per batch item) and norm3
    (transformers.py:292-399)."""
    import torch.nn.functional as F

    from audio_motion_avatar_amd import ops

    g = torch.Generator().manual_seed(dim)
    B, S = 2, 777
    h = torch.randn(B, S, dim, generator=g).cuda()
    a = torch.randn(B, S, dim, generator=g).cuda()
    row = torch.randn(B, 1, dim, generator=g).cuda()
    w = (1 + 0.1 * torch.randn(dim, generator=g)).cuda()
    b = (0.1 * torch.randn(dim, generator=g)).cuda()
    h_out, n = ops.add_layernorm(h, a, row, w, b, 1e-5)
    want_h = row + (a + h)
    assert torch.equal(h_out, want_h)
    want_n = F.layer_norm(want_h.double(), (dim,), w.double(), b.double(), 1e-5)
    assert (n.double() - want_n).abs().max().item() <= 5e-6
    h2, n2 = ops.add_layernorm(h, None, None, w, b, 1e-5)
    assert torch.equal(h2, h) and (n2.double() - F.layer_norm(h.double(), (dim,), w.double(), b.double(), 1e-5)).abs().max() <= 5e-6


@pytest.mark.parametrize("slack", [1.0, 40.0])
def test_selfattn_with_caller_bounds_matches_the_measured_scaling(slack):
    from audio_motion_avatar_amd import ops

    g = torch.Generator().manual_seed(77)
    B, S, H = 1, 700, 4
    q, k, v = (torch.randn(B, S, H * 64, generator=g) * s for s in (3.0, 0.02, 50.0))  # very different magnitudes
    bounds = tuple(t.abs().max().item() * slack for t in (q, k, v))
    out = ops.selfattn(q.cuda(), k.cuda(), v.cuda(), H, bounds=bounds).cpu()
    ref = reference(q, k, v, H)
    assert (out.double() - ref).abs().max() <= 2e-5 * 50.0
    measured = ops.selfattn(q.cuda(), k.cuda(), v.cuda(), H).cpu()
    assert (out - measured).abs().max() <= 1e-5 * 50.0


def test_selfattn_rejects_partial_bounds():
    from audio_motion_avatar_amd import ops
    from audio_motion_avatar_amd._lib import AmavError

    x = torch.randn(1, 64, 64).cuda()
    with pytest.raises(AmavError):
        ops.selfattn(x, x, x, 1, bounds=(1.0, 0.0, 1.0))


@pytest.mark.parametrize("B,S,H", [(1, 6304, 8), (1, 200, 2), (2, 777, 4)])
def test_split_output_equals_the_split_of_the_fp32_result(B, S, H):
    """amav_selfattn_forward_split_out: the kernel's last pass writes the next projection's fp16 x 2 operand ([h2 | h1 |
    h1], x 2^e = h1 + h2) -- bit for bit what amav_split_operand makes of the fp32 result, with the key range split over
    workgroups (6304 keys: the merge pass writes it) and without (200 keys: the library splits the result itself)."""
    from audio_motion_avatar_amd import ops

    g = torch.Generator(device="cuda").manual_seed(S + H)
    qkv = torch.randn(B, S, 3 * H * 64, device="cuda", generator=g)
    i = H * 64
    q, k, v = qkv[..., :i], qkv[..., i:2 * i], qkv[..., 2 * i:]
    bounds = (float(q.abs().max()) * 1.1, float(k.abs().max()) * 1.1, float(v.abs().max()) * 1.1)
    exp = 9  # |out| <= max |v| < 8: 2^9 keeps h1 far inside fp16
    want = ops.split_operand(ops.selfattn(q, k, v, H, bounds=bounds).view(-1, i), fmt=ops.SPLIT_FP16X2, scale_exp=exp)
    got = ops.selfattn(q, k, v, H, bounds=bounds, split_out_exp=exp)
    assert got.dtype == torch.float16 and tuple(got.shape) == (B * S, 3 * i)
    assert torch.equal(got, want)
    got2 = ops.selfattn(q, k, v, H, split_out_exp=exp)  # measured magnitudes
    assert torch.equal(got2, ops.split_operand(ops.selfattn(q, k, v, H).view(-1, i), fmt=ops.SPLIT_FP16X2, scale_exp=exp))
