"""The near-exact "f16x3" mode (split-fp16 operands, three fp16 MFMAs per product, fp32 accumulation): each kernel against an fp64
reference of the SAME fp32 operation at fp32-level tolerances, then the extractor against the CPU oracle and against the
library's exact-fp32 mode.  What it stands in for: the reference's fp32 backbone arithmetic (/root/reference/scripts/infer.py:468-473
through utils/dinov2_utils.py:257) at ~3.5x the speed of the fp32-input MFMA path."""
import math

import numpy as np
import pytest
import torch

from foundpose_amd import _lib, ops, synthetic
from foundpose_amd.vit_config import ARCHS, VitArch
from oracle import vit as ov
from tests.helpers import experiments_build

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def test_split16_pack_unpack_round_trip_and_layout():
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(7, 96, generator=g) * torch.logspace(-4, 3, 96)[None, :]).cuda()
    s = ops.pow2_scale(x)
    assert s == 2.0 ** math.floor(math.log2(16384.0 / float(x.abs().max())))
    p = ops.split16_pack(x, s)
    assert p.shape == (7, 192) and p.dtype == torch.float16
    # group g of 32 columns = halves [64 g, 64 g + 32) hi, [64 g + 32, 64 g + 64) lo
    hi = (x * s).half()
    assert torch.equal(p[:, 64:96], hi[:, 32:64]) and torch.equal(p[:, 96:128], ((x * s) - hi.float()).half()[:, 32:64])
    back = ops.split16_unpack(p, s)
    big = x.abs() * s > 0.25      # lo is a normal fp16 number there: 22 mantissa bits
    assert float(((back - x).abs() / x.abs())[big].max()) < 2.0 ** -21
    assert float((back - x).abs()[~big].max()) * s <= 2.0 ** -24   # below: the fp16 subnormal spacing (absolute)
    padded = ops.split16_pack(x, s, pad=64)
    assert padded.stride(0) == 192 + 64 and torch.equal(padded, p)


def test_mfma_f16_subnormal_inputs_are_honoured():
    """Hardware fact the scales of the mode do not depend on (they keep lo halves normal) but worth pinning: fp16 subnormal MFMA inputs
    are multiplied, not flushed."""
    M, K, N = 256, 64, 128
    a = torch.full((M, K), 2.0 ** -20, device="cuda")          # fp16 subnormal (min normal 2^-14), exactly representable
    w = torch.full((N, K), 1024.0, device="cuda")
    out = ops.gemm_split(ops.split16_pack(a), ops.split16_pack(w), torch.zeros(N, device="cuda"), 1.0, epilogue=5)
    want = K * 2.0 ** -20 * 1024.0
    assert torch.allclose(out, torch.full_like(out, want), rtol=1e-6), f"subnormal operands flushed? got {float(out[0, 0])}, want {want}"


@pytest.mark.parametrize("M,N,K,tile", [(256, 256, 1024, 256), (256, 384, 96, 128), (512, 1024, 4096, 0), (128, 128, 32, 0)])
def test_gemm_split_fp32_epilogue_vs_fp64(M, N, K, tile):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device="cuda") * 1.7
    a[:, 3] *= 40.0                                                     # an outlier channel, as LayerNorm outputs have
    w = torch.randn(N, K, generator=g, device="cuda") * 0.02
    bias = torch.randn(N, generator=g, device="cuda")
    sa, sw = 128.0, ops.pow2_scale(w)
    out = ops.gemm_split(ops.split16_pack(a, sa), ops.split16_pack(w, sw, pad=64), bias, 1.0 / (sa * sw), epilogue=5, tile=tile)
    ref = a.double() @ w.double().T + bias.double()
    mag = a.double().abs() @ w.double().abs().T                         # sum |a_k w_k|: what a rounding error scales with
    err = float(((out.double() - ref).abs() / mag).max())
    # per-product error <= ~3 * 2^-22, plus the fp32 accumulation of K terms; the bf16 path sits at 2^-8
    assert err < 2.0 ** -20, err
    exact32 = ops.gemm_f32(a, w, bias, epilogue=4)                      # the exact-fp32 MFMA path (k-ascending fmaf chains)
    err32 = float(((exact32.double() - ref).abs() / mag).max())
    assert err < 4 * err32 + 2.0 ** -23, (err, err32)                   # same league as fp32 arithmetic itself


@pytest.mark.parametrize("tile", [128, 256])
def test_gemm_split_epilogues(tile):
    g = torch.Generator(device="cuda").manual_seed(tile)
    M, K, N, mv = 512, 256, 512, 391
    a = torch.randn(M, K, generator=g, device="cuda")
    w = torch.randn(N, K, generator=g, device="cuda") * 0.05
    bias = torch.randn(N, generator=g, device="cuda") * 0.3
    gamma = torch.rand(N, generator=g, device="cuda") + 0.5
    sa, sw = 128.0, ops.pow2_scale(w)
    A, W = ops.split16_pack(a, sa), ops.split16_pack(w, sw)
    lin = a.double() @ w.double().T + bias.double()
    # 0: bias -> split rows (scale 64), padding rows untouched
    o0 = ops.gemm_split(A, W, bias, 1.0 / (sa * sw), epilogue=0, out_scale=64.0, m_valid=mv, tile=tile)
    assert o0.shape == (M, 2 * N) and rel_err(ops.split16_unpack(o0[:mv], 64.0), lin[:mv]) < 2e-6
    assert not bool(o0[mv:].any())
    # 1: exact-erf GELU -> split rows
    o1 = ops.gemm_split(A, W, bias, 1.0 / (sa * sw), epilogue=1, out_scale=64.0, tile=tile)
    assert rel_err(ops.split16_unpack(o1, 64.0), torch.nn.functional.gelu(lin)) < 2e-6
    # 6: SwiGLU on interleaved columns -> split rows [M, N/2 logical]
    o6 = ops.gemm_split(A, W, bias, 1.0 / (sa * sw), epilogue=6, out_scale=64.0, tile=tile)
    sw_ref = torch.nn.functional.silu(lin[:, 0::2]) * lin[:, 1::2]
    assert o6.shape == (M, N) and rel_err(ops.split16_unpack(o6, 64.0), sw_ref) < 2e-6
    # 3: x += gamma * (.) in place on the fp32 stream
    x0 = torch.randn(M, N, generator=g, device="cuda")
    x = x0.clone()
    ops.gemm_split(A, W, bias, 1.0 / (sa * sw), gamma=gamma, out=x, epilogue=3, m_valid=mv, tile=tile)
    assert rel_err(x[:mv], x0[:mv].double() + gamma.double() * lin[:mv]) < 2e-6 and torch.equal(x[mv:], x0[mv:])


def test_gemm_split_saturates_instead_of_overflowing():
    """A value beyond 65504 / scale cannot be represented in the fp16 pair: it saturates (finite), it never becomes inf / NaN."""
    M, K, N = 128, 32, 128
    a = torch.zeros(M, K, device="cuda")
    a[:, 0] = 1.0
    w = torch.zeros(N, K, device="cuda")
    w[:, 0] = 5000.0
    out = ops.gemm_split(ops.split16_pack(a), ops.split16_pack(w, 1.0), torch.zeros(N, device="cuda"), 1.0, epilogue=0, out_scale=64.0)
    v = ops.split16_unpack(out, 64.0)
    assert bool(torch.isfinite(out.float()).all()) and torch.allclose(v, torch.full_like(v, 65504.0 / 64.0), rtol=1e-3)


def test_layernorm_split_vs_torch():
    g = torch.Generator(device="cuda").manual_seed(3)
    for D in (384, 1024):
        x = torch.randn(300, D, generator=g, device="cuda") * 3 + 0.5
        wt, b = torch.rand(D, generator=g, device="cuda") + 0.5, torch.randn(D, generator=g, device="cuda") * 0.1
        got = ops.split16_unpack(ops.layernorm_split(x, wt, b, 128.0), 128.0)
        ref = torch.nn.functional.layer_norm(x.double(), (D,), wt.double(), b.double(), 1e-6)
        exact = ops.layernorm(x, wt, b, torch.float32)
        assert rel_err(got, ref) < 2 * rel_err(exact, ref) + 2.0 ** -21


def _ref_attention(q, k, v, B, N, heads):
    """q, k, v fp64 [B*N, D] -> softmax(q k^T / 8) v, [B*N, D]"""
    D = q.shape[1]
    sh = lambda t: t.reshape(B, N, heads, 64).permute(0, 2, 1, 3)
    p = torch.softmax(sh(q) @ sh(k).transpose(-1, -2) * 0.125, dim=-1)
    return (p @ sh(v)).permute(0, 2, 1, 3).reshape(B * N, D)


@pytest.mark.parametrize("B,N,heads", [(2, 77, 2), (1, 1374, 4), (3, 905, 2), (1, 256, 1), (2, 257, 3), (1, 64, 16)])
def test_attention_split_vs_fp64(B, N, heads):
    D = heads * 64
    g = torch.Generator(device="cuda").manual_seed(N + heads)
    qkv = torch.randn(B * N, 3 * D, generator=g, device="cuda") * 1.5
    qkv[:, 5] *= 6.0   # a dominant q dimension: peaked softmax rows
    packed = torch.cat([ops.split16_pack(qkv[:, i * D:(i + 1) * D].contiguous(), 64.0) for i in range(3)], dim=1)
    out = ops.split16_unpack(ops.attention_split(packed, B, N, D, heads, 64.0, 128.0), 128.0)
    q, k, v = (qkv[:, i * D:(i + 1) * D].double() for i in range(3))
    ref = _ref_attention(q, k, v, B, N, heads)
    o32 = ops.attention(qkv, B, N, D, heads)   # the fp32 VALU kernel of the exact mode
    e, e32 = rel_err(out, ref), rel_err(o32, ref)
    assert e < 5e-6 and e < 4 * e32 + 1e-6, (e, e32)


@pytest.mark.parametrize("B,N,heads", [(2, 77, 2), (1, 1374, 4), (3, 905, 2), (1, 256, 1), (2, 257, 3), (1, 64, 16), (1, 33, 1), (2, 129, 8), (1, 2305, 2)])
@pytest.mark.parametrize("f16f8_out", [False, True])
@pytest.mark.skipif(not experiments_build(), reason="attn_split_pp_kernel (measured ~3 % slower) is compiled into FP_EXPERIMENTS builds only")
def test_attention_split_role_split_kernel_equals_lock_step_kernel(B, N, heads, f16f8_out):
    """attn_split_pp_kernel (variant 2: the two waves of a SIMD half a key tile apart, three-slot K / V ring) issues the same MFMAs in the
    same order per accumulator and the same softmax as attn_split_kernel (variant 1): every output bit equal -- one tile, ragged last tiles
    of both halves' widths, short last query tiles (waves without queries), more than one query tile, both output row formats."""
    D = heads * 64
    g = torch.Generator(device="cuda").manual_seed(7 * N + heads)
    qkv = torch.randn(B * N, 3 * D, generator=g, device="cuda") * 1.5
    qkv[:, 5] *= 6.0
    packed = torch.cat([ops.split16_pack(qkv[:, i * D:(i + 1) * D].contiguous(), 64.0) for i in range(3)], dim=1)
    a = ops.attention_split(packed, B, N, D, heads, 64.0, 128.0, f16f8_out=f16f8_out, variant=2)
    b = ops.attention_split(packed, B, N, D, heads, 64.0, 128.0, f16f8_out=f16f8_out, variant=1)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    for _ in range(3):   # the role-split schedule has more ways to race than the lock-step one: the same bits on every launch
        assert torch.equal(ops.attention_split(packed, B, N, D, heads, 64.0, 128.0, f16f8_out=f16f8_out, variant=2).view(torch.int16), a.view(torch.int16))


def test_attention_split_forced_rescale_and_constant_rows():
    """Online-softmax corner cases: a key late in the sequence that dominates every earlier one (the running max jumps at a chosen
    tile and everything accumulated so far is rescaled), and all-equal scores (uniform attention)."""
    B, N, heads, D = 1, 300, 1, 64
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn(N, 3 * D, generator=g, device="cuda")
    qkv[200, D:2 * D] = qkv[7, 0:D] * 4.0          # key 200 aligned with query 7: its score towers over the rest
    packed = torch.cat([ops.split16_pack(qkv[:, i * D:(i + 1) * D].contiguous(), 64.0) for i in range(3)], dim=1)
    out = ops.split16_unpack(ops.attention_split(packed, B, N, D, heads, 64.0, 128.0), 128.0)
    ref = _ref_attention(*(qkv[:, i * D:(i + 1) * D].double() for i in range(3)), B, N, heads)
    assert rel_err(out, ref) < 5e-6
    qkv[:, :2 * D] = 0.0                           # q = k = 0: uniform attention -> the mean of v
    packed = torch.cat([ops.split16_pack(qkv[:, i * D:(i + 1) * D].contiguous(), 64.0) for i in range(3)], dim=1)
    out = ops.split16_unpack(ops.attention_split(packed, B, N, D, heads, 64.0, 128.0), 128.0)
    assert rel_err(out, qkv[:, 2 * D:].double().mean(0, keepdim=True).expand(N, D)) < 5e-6


@pytest.mark.parametrize("step", [0.4, 0.9, 3.0, 20.0])
def test_attention_split_lazy_rescale_staircase(step):
    """attn_split_kernel's reference may trail the maximum by at most 1 in the exponent (p <= 2 keeps P * 2^14 inside fp16): scores that
    climb by `step` per 64-key tile, queries of four gains in one wave -- no rescale for several tiles, one every tile, jumps of 2^20."""
    N, D, heads = 64 * 9 + 17, 64, 1
    g = torch.Generator(device="cuda").manual_seed(int(step * 10))
    qkv = torch.randn(N, 3 * D, generator=g, device="cuda") * 0.3
    gain = torch.tensor([1.0, 2.0, 4.0, 0.5], device="cuda")[torch.arange(N, device="cuda") % 4]
    qkv[:, 0] = gain * 4.0
    tile = (torch.arange(N, device="cuda") // 64).float()
    qkv[:, D] = tile * step / (4.0 * 4.0 * 0.125 * 1.4426950408889634)
    packed = torch.cat([ops.split16_pack(qkv[:, i * D:(i + 1) * D].contiguous(), 64.0) for i in range(3)], dim=1)
    out = ops.split16_unpack(ops.attention_split(packed, 1, N, D, heads, 64.0, 128.0), 128.0)
    ref = _ref_attention(*(qkv[:, i * D:(i + 1) * D].double() for i in range(3)), 1, N, heads)
    o32 = ops.attention(qkv, 1, N, D, heads)   # the exact mode's fp32 kernel: scores of magnitude 10^2..10^3 carry fp32 rounding of their own
    e, e32 = rel_err(out, ref), rel_err(o32, ref)
    assert e < 5e-6 or e < 2 * e32, (e, e32)


TINY = VitArch("tiny-reg", dim=128, depth=3, heads=2, ffn="mlp", hidden=512, registers=4, pretrain_grid=4, interp_antialias=True, interp_offset=0.0)
TINY_G = VitArch("tinyg-reg", dim=128, depth=2, heads=2, ffn="swiglu", hidden=384, registers=4, pretrain_grid=4, interp_antialias=True, interp_offset=0.0)


@pytest.mark.parametrize("arch,layer,size,B", [(TINY, 2, 56, 3), (TINY_G, 1, 70, 2), (ARCHS["vits14-reg"], 9, 224, 2), (ARCHS["vits14-reg"], 9, 420, 1)])
def test_extractor_f16x3_at_fp32_noise_level(arch, layer, size, B):
    """Features of the f16x3 mode against the fp32 CPU oracle: as close as the library's exact-fp32 mode is (both carry fp32
    accumulation noise, in different summation orders), two orders of magnitude closer than the bf16 mode."""
    from foundpose_amd import feature_util
    name = f"dinov2_version={arch.name}_stride=14_facet=token_layer={layer}_norm=1"
    sd = synthetic.make_vit_state_dict(arch, seed=5)
    imgs = synthetic.make_crops(B, size, seed=2)
    ref = ov.extractor_forward(sd, arch, imgs, layer, True)
    outs = {}
    for prec in ("fp32", "f16x3"):
        ex = feature_util.make_feature_extractor(name, state_dict=sd, arch=arch, precision=prec).to("cuda")
        o = ex(imgs.cuda())
        outs[prec] = (o["feature_maps"].cpu(), o["cls_tokens"].cpu())
    e32, e3 = rel_err(outs["fp32"][0], ref["feature_maps"]), rel_err(outs["f16x3"][0], ref["feature_maps"])
    assert e3 < 5e-5 and e3 < 3 * e32 + 1e-5, (e3, e32)
    assert rel_err(outs["f16x3"][1], ref["cls_tokens"]) < 5e-5
    assert rel_err(outs["f16x3"][0], outs["fp32"][0]) < 5e-5


def test_extractor_f16x3_vitl_518_metric_config_and_batch_invariance():
    """The bench configuration (ViT-L/14-reg, layer 18, 518 x 518): against the oracle on one crop, and crop 3 of a batch of 8 equals
    that crop run alone bit for bit (a row's arithmetic never depends on the batch)."""
    from foundpose_amd import feature_util
    arch = ARCHS["vitl14-reg"]
    name = "dinov2_version=vitl14-reg_stride=14_facet=token_layer=18_norm=1"
    sd = synthetic.make_vit_state_dict(arch, seed=1234)
    imgs = synthetic.make_crops(8, 518, seed=0)
    ref = ov.extractor_forward(sd, arch, imgs[3:4], 18, True)["feature_maps"]
    ex = feature_util.make_feature_extractor(name, state_dict=sd, precision="f16x3").to("cuda")
    one = ex(imgs[3:4].cuda())["feature_maps"].clone()
    assert rel_err(one.cpu(), ref) < 5e-5
    batch = ex(imgs.cuda())["feature_maps"]
    assert torch.equal(batch[3], one[0])


def test_extractor_f16x3_facets_vs_fp32_mode():
    from foundpose_amd import feature_util
    sd = synthetic.make_vit_state_dict(TINY, seed=9)
    imgs = synthetic.make_crops(2, 56, seed=4).cuda()
    for facet in ("key", "query", "value"):
        name = f"dinov2_version=tiny-reg_stride=14_facet={facet}_layer=2_norm=1"
        a = feature_util.make_feature_extractor(name, state_dict=sd, arch=TINY, precision="fp32").to("cuda")(imgs)["feature_maps"]
        b = feature_util.make_feature_extractor(name, state_dict=sd, arch=TINY, precision="f16x3").to("cuda")(imgs)["feature_maps"]
        assert rel_err(b, a) < 2e-5


def test_extractor_f16x3_with_massive_activation_channels():
    """Real checkpoints carry a few residual channels hundreds of times larger than the rest (and tokens whose hidden activations are
    in the hundreds).  Plant them -- one fc2 output channel x 400 in block 0, one fc1 unit x 60 in block 1 -- and check that the pairs'
    fixed scales have the head room: f16x3 stays at the fp32 mode's noise level and nothing saturates."""
    from foundpose_amd import feature_util
    arch = ARCHS["vits14-reg"]
    name = "dinov2_version=vits14-reg_stride=14_facet=token_layer=9_norm=1"
    sd = {k: v.clone() for k, v in synthetic.make_vit_state_dict(arch, seed=5).items()}
    sd["blocks.0.mlp.fc2.weight"][17] *= 400.0
    sd["blocks.1.mlp.fc1.weight"][33] *= 60.0
    sd["blocks.1.mlp.fc1.bias"][33] += 150.0
    imgs = synthetic.make_crops(2, 224, seed=2)
    ref = ov.extractor_forward(sd, arch, imgs, 9, True)["feature_maps"]
    hs = ov.hidden_after_block(sd, arch, imgs, 0, None, False, None)
    assert float(hs.abs().max()) > 100.0          # the planted channel really is massive
    outs = {}
    for prec in ("fp32", "f16x3"):
        ex = feature_util.make_feature_extractor(name, state_dict=sd, precision=prec).to("cuda")
        outs[prec] = ex(imgs.cuda())["feature_maps"].cpu()
        assert bool(torch.isfinite(outs[prec]).all())
    e32, e3 = rel_err(outs["fp32"], ref), rel_err(outs["f16x3"], ref)
    assert e3 < 3 * e32 + 1e-5 and e3 < 1e-4, (e3, e32)
    assert ex.saturation_counts() == (0, 0)          # ... and the device-side counters agree: nothing was clamped


@pytest.mark.parametrize("where", ["layernorm", "qkv", "hidden", "nan", "nan_attention"])
def test_f16x3_saturation_is_loud(where):
    """An activation beyond the range of its split-fp16 row (+-4094 for LayerNorm outputs and q / k / v, +-16376 for the hidden
    activations) is clamped by the producer kernel -- and REPORTED: the sticky device-side counter (fp_vit_workspace.sat) goes up, the
    extractor's forward raises FoundPoseSaturationError and keeps raising until reset_saturation(), and a result of the batched engine
    raises when it is read.  The same extractor on weights without the outlier reports nothing (padding rows never do)."""
    from foundpose_amd import _lib, engine as fe, feature_util, workload
    from foundpose_amd.bank import DeviceBank
    arch = ARCHS["vits14-reg"]
    name = "dinov2_version=vits14-reg_stride=14_facet=token_layer=3_norm=1"
    sd = {k: v.clone() for k, v in synthetic.make_vit_state_dict(arch, seed=5).items()}
    if where == "layernorm":
        sd["blocks.1.norm1.bias"][7] = 5000.0          # LayerNorm output channel 7 = 5000 + ... > 4094
    elif where == "qkv":
        sd["blocks.2.attn.qkv.bias"][arch.dim + 3] = -6000.0   # a key channel at -6000
    elif where == "hidden":
        sd["blocks.0.mlp.fc1.bias"][11] = 20000.0      # gelu(20000) = 20000 > 16376
    imgs = synthetic.make_crops(3, 112, seed=2).cuda()
    ex = feature_util.make_feature_extractor(name, state_dict=sd, precision="f16x3")
    # a NaN activation: the packing clamps (v_med3) would turn it into a FINITE operand and the features would look plausible -- the running
    # maxima of the report propagate NaN (v_maximum3_f32), so it is as loud as a clamp.  (A checkpoint with a NaN is refused at load time,
    # weights.validate_state_dict: the NaN is planted behind the validation, into the dict the device copy is made from.)
    if where == "nan":
        sd["blocks.1.norm1.bias"][7] = float("nan")             # -> a LayerNorm output
    elif where == "nan_attention":
        sd["blocks.2.attn.qkv.bias"][2 * arch.dim + 5] = float("nan")   # -> a value channel: surfaces in qkv's packing AND in the attention output's
    ex = ex.to("cuda")
    with pytest.raises(_lib.FoundPoseSaturationError, match="clamped an activation"):
        ex(imgs)
    n16, n8 = ex.saturation_counts()
    assert n16 > 0 and n8 == 0
    with pytest.raises(_lib.FoundPoseSaturationError):    # sticky
        ex.check_saturation()
    ex.reset_saturation()
    assert ex.saturation_counts() == (0, 0)
    if where.startswith("nan"):
        # ... while the fp32 mode (no clamps anywhere) shows the NaN in its features, as the reference's arithmetic would
        clean_sd = {k: (torch.nan_to_num(v, nan=0.0) if v.dtype.is_floating_point else v) for k, v in sd.items()}
        ex32 = feature_util.make_feature_extractor(name, state_dict=clean_sd, precision="fp32")
        ex32._sd = {k: sd[k] for k in ex32._sd}
        assert bool(torch.isnan(ex32.to("cuda")(imgs)["feature_maps"]).any())
    else:
        # the fp32 mode computes the same weights without complaint (no operand rows to leave)
        ex32 = feature_util.make_feature_extractor(name, state_dict=sd, precision="fp32").to("cuda")
        assert bool(torch.isfinite(ex32(imgs)["feature_maps"]).all())
    # through the engine: the result raises when its correspondences are read
    clean = feature_util.make_feature_extractor(name, random_init_seed=5, precision="f16x3").to("cuda")
    wl = workload.build_planted_workload(clean, 3, 112, 1, 40, seed=3, crop_seed=2)
    bank = DeviceBank(wl.repres)
    res = fe.FoundPoseEngine(clean, bank, 14.0, 5, 300, tie_order="torch").infer_batch(wl.crops, wl.masks, wl.det_obj)
    assert len(res.corresp_list(0)) == 5 and clean.saturation_counts() == (0, 0)     # clean weights: nothing reported, token selection included
    res = fe.FoundPoseEngine(ex, bank, 14.0, 5, 300, tie_order="torch").infer_batch(wl.crops, wl.masks, wl.det_obj)
    for b in (0, 1, 0):     # the verdict belongs to the result: every access raises, any detection index, any number of times
        with pytest.raises(_lib.FoundPoseSaturationError):
            res.corresp_list(b)
    # ... and to ITS batch: clamps counted before a batch (here: planted into the sticky counters) do not make a clean batch's result raise,
    # while the extractor's own forward stays sticky until reset_saturation()
    clean._sat += 5
    res = fe.FoundPoseEngine(clean, bank, 14.0, 5, 300, tie_order="torch").infer_batch(wl.crops, wl.masks, wl.det_obj)
    assert len(res.corresp_list(0)) == 5 and len(res.corresp_list(2)) == 5
    with pytest.raises(_lib.FoundPoseSaturationError):
        clean.check_saturation()
    clean.reset_saturation()


def test_fp8_saturation_is_reported():
    """The fp8 mode quantises with STATIC scales: an activation beyond them is clamped at +-448.  With honest calibration scales on the
    calibration maxima (and 2x head room) nothing clamps; with scales 8x too large the counter goes up and the forward warns (once)."""
    from foundpose_amd import feature_util
    name = "dinov2_version=vitb14-reg_stride=14_facet=token_layer=2_norm=1"
    imgs = synthetic.make_crops(2, 112, seed=4).cuda()
    ex = feature_util.make_feature_extractor(name, random_init_seed=7, precision="fp8").to("cuda")
    scales = ex.calibrate_fp8(imgs, headroom=1.0)    # the sample maxima themselves
    roomy = feature_util.make_feature_extractor(name, random_init_seed=7, precision="fp8", act_scales=scales * 0.5).to("cuda")   # 2x head room over the calibration maxima
    roomy(imgs)
    assert roomy.saturation_counts() == (0, 0)
    bad = feature_util.make_feature_extractor(name, random_init_seed=7, precision="fp8", act_scales=scales * 8.0).to("cuda")
    with pytest.warns(UserWarning, match="clamped an activation at"):
        bad(imgs)
    assert bad.saturation_counts()[1] > 0 and bad.saturation_counts()[0] == 0
