"""GPU parity of the "f16" mode: the bf16 pipeline's kernels (same templates, tiles, bytes) on IEEE fp16 operands (include/foundpose_amd.h
"plain fp16 rows").  The reference computes in fp32 (/root/reference/utils/dinov2_utils.py:257, scripts/infer.py:468-473): every test here holds the
mode to a bar that is the bf16 mode's divided by the three mantissa bits fp16 has more, next to the fp64 / oracle value on the SAME 16-bit operands.
"""

import numpy as np
import pytest
import torch

from foundpose_amd import _lib, synthetic
from foundpose_amd.vit_config import ARCHS
from oracle import vit as ov
from tests.helpers import TINY, check_bar

pytestmark = pytest.mark.gpu

H = torch.float16
EPS16 = 2.0 ** -11   # half-ulp of an fp16 result relative to its binade


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 1024), (1408, 3072, 1024), (384, 1024, 4096), (256, 256, 192)])
def test_gemm_f16_epilogues_vs_fp64(M, N, K, tile):
    """bias -> fp32, bias -> fp16, GELU -> fp16, SwiGLU -> fp16, LayerScale + residual (fp32) against fp64 on the same fp16 operands: fp32 accumulation
    error for the fp32 outputs, ONE fp16 rounding of the result for the 16-bit ones (+ 3.8e-5 absolute for the GELU's nine-coefficient polynomial)."""
    from foundpose_amd import ops
    if tile == 256 and (M % 256 or N % 256):
        pytest.skip("256 tile needs M, N multiples of 256")
    t = tile << 8
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(H)
    w = (torch.randn(N, K, generator=g) * 0.05).to(H)
    bias, gamma, resid = torch.randn(N, generator=g), torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().T + bias.double()
    scale = float(ref.abs().max())
    ac, wc, bc = a.cuda(), w.cuda(), bias.cuda()
    out = ops.gemm_bf16(ac, wc, bc, epilogue=5 | t).cpu()
    assert float((out.double() - ref).abs().max()) < 2e-5 * scale * max(1, K / 1024)
    out = ops.gemm_bf16(ac, wc, bc, epilogue=0 | t).cpu()
    assert out.dtype == H and float((out.double() - ref).abs().max()) < 1.01 * EPS16 * scale
    out = ops.gemm_bf16(ac, wc, bc, epilogue=1 | t).cpu()
    assert float((out.double() - torch.nn.functional.gelu(ref)).abs().max()) < 1.01 * EPS16 * scale + 4e-5
    out = ops.gemm_bf16(ac, wc, bc, epilogue=6 | t).cpu()
    sref = torch.nn.functional.silu(ref[:, 0::2]) * ref[:, 1::2]
    assert out.shape == (M, N // 2) and float((out.double() - sref).abs().max()) < 1.01 * EPS16 * float(sref.abs().max())
    x = resid.clone().cuda()
    ops.gemm_bf16(ac, wc, bc, gamma=gamma.cuda(), out=x, epilogue=3 | t)
    rref = resid.double() + gamma.double() * ref
    assert float((x.cpu().double() - rref).abs().max()) < 3e-5 * float(rref.abs().max()) * max(1, K / 1024)
    out = torch.full((M, N), 777.0, dtype=H).cuda()   # rows past M_valid stay untouched
    ops.gemm_bf16(ac, wc, bc, out=out, epilogue=0 | t, m_valid=M - 5)
    assert torch.all(out[M - 5:] == 777.0) and torch.all(out[:M - 5] != 777.0)


def test_gemm_f16_is_closer_to_fp32_than_bf16_on_the_same_fp32_operands():
    """What the mode is for: fp32 operands rounded to fp16 instead of bf16 lose 8x less -- the product error against the fp32 operands' own fp64
    product drops by about that factor."""
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(3)
    a32, w32 = torch.randn(512, 1024, generator=g), torch.randn(768, 1024, generator=g) * 0.03
    ref = a32.double() @ w32.double().T
    z = torch.zeros(768).cuda()
    e = {}
    for dt in (torch.bfloat16, H):
        out = ops.gemm_bf16(a32.to(dt).cuda(), w32.to(dt).cuda(), z, epilogue=5).cpu()
        e[dt] = float((out.double() - ref).abs().max() / ref.abs().max())
    assert e[H] < e[torch.bfloat16] / 5 and e[H] < 6e-4, e


@pytest.mark.parametrize("M,N,K,m_valid", [(1280, 512, 128, 1280), (3840, 3072, 1024, 3140)])
def test_gemm_f16_tile_shapes_agree_bitwise(M, N, K, m_valid):
    """320 x 256 vs 256 x 256 (bias / GELU, plain and folded-LayerNorm form, (hi, lo) residual) and 64 x 128 vs 128 x 128 (residual forms): the same
    k order per output element, the same bits."""
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(H).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(H).cuda()
    bias, cs = torch.randn(N, generator=g).cuda(), torch.randn(N, generator=g).cuda()
    ln_row = torch.stack([1 + torch.rand(M, generator=g), torch.randn(M, generator=g)], 1).contiguous().cuda()
    for epi in (0, 1):
        outs = []
        for tile in (256, 320):
            out = torch.full((M, N), 777.0, dtype=H, device="cuda")
            ops.gemm_bf16(a, w, bias, out=out, epilogue=epi | (tile << 8), m_valid=m_valid)
            outs.append(out)
        assert torch.equal(outs[0], outs[1]) and bool(torch.all(outs[1][m_valid:] == 777.0))
        assert torch.equal(ops.gemm_bf16_ln(a, w, bias, cs, ln_row, epilogue=epi, tile=256, m_valid=m_valid)[:m_valid],
                           ops.gemm_bf16_ln(a, w, bias, cs, ln_row, epilogue=epi, tile=320, m_valid=m_valid)[:m_valid])
    x0 = torch.randn(M, N, generator=g) * 3 + 0.7
    res = []
    for tile in (128, 64, 256, 320):
        xb, xl = x0.to(H).cuda(), (x0 - x0.to(H).float()).to(H).cuda()
        st = ops.gemm_bf16_resid_hilo(a, w, bias, xb, xl, tile=tile, m_valid=m_valid)
        x7 = x0.clone().cuda()
        xb7, st7 = ops.gemm_bf16_resid_ln(a, w, bias, x7, tile=tile if tile != 320 else 256, m_valid=m_valid)
        res.append((xb, xl, st[:, :m_valid].clone(), x7, xb7[:m_valid].clone(), st7[:, :m_valid].clone()))
    for other in res[1:]:
        for t0, t1 in zip(res[0], other):
            assert torch.equal(t0, t1)


@pytest.mark.parametrize("tile,M,D,N2", [(128, 256, 256, 512), (256, 512, 1024, 1024)])
def test_folded_layernorm_pair_and_hi_lo_stream_f16(tile, M, D, N2):
    """The LayerNorm fold at op level on fp16 arrays: the (hi, lo) residual producer against fp64 (hi' = f16(x'), hi' + lo' = x' to 21 bits, row sums),
    the fp32-stream producer's fp16 copy, and the normalising consumer against fp64 on the same fp16 operand at one fp16 rounding of the result."""
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(M + D)
    h = torch.randn(M, 2 * D, generator=g).to(H)
    w_out = (torch.randn(D, 2 * D, generator=g) * 0.05).to(H)
    b_out = torch.randn(D, generator=g)
    x0 = torch.randn(M, D, generator=g) * 3 + 0.7
    x0[:, 5] *= 200.0                                              # a massive-activation channel
    hi, lo = x0.to(H), (x0 - x0.to(H).float()).to(H)
    x32 = hi.float() + lo.float()
    x_ref = x32.double() + h.double() @ w_out.double().T + b_out.double()
    mv = M - 3
    xb, xl = hi.clone().cuda(), lo.clone().cuda()
    st = ops.gemm_bf16_resid_hilo(h.cuda(), w_out.cuda(), b_out.cuda(), xb, xl, tile=tile, m_valid=mv)
    x7 = x32.clone().cuda()
    xb7, st7 = ops.gemm_bf16_resid_ln(h.cuda(), w_out.cuda(), b_out.cuda(), x7, tile=tile, m_valid=mv)
    assert rel_err(x7[:mv].cpu(), x_ref[:mv]) < 3e-5
    assert torch.equal(xb[:mv], xb7[:mv]) and torch.equal(xb7[:mv].cpu(), x7[:mv].cpu().to(H))
    got = (xb.float() + xl.float())[:mv].cpu().double()            # the pair represents epilogue 7's fp32 x' (the same value, asserted through xb above) ...
    x7d = x7[:mv].cpu().double()
    assert bool(((got - x7d).abs() <= 2.0 ** -21 * x7d.abs() + 6e-8).all())   # ... to 21+ bits (a low half below 6e-5 is an fp16 subnormal: 3e-8 absolute)
    assert torch.equal(xb[mv:].cpu(), hi[mv:]) and torch.equal(xl[mv:].cpu(), lo[mv:])
    torch.testing.assert_close(st[:, :mv], st7[:, :mv], rtol=2e-6, atol=2e-2)
    ln_row = ops.ln_finalize(st, D)
    mu, var = x_ref[:mv].mean(1), x_ref[:mv].var(1, unbiased=False)
    assert rel_err(ln_row[:mv, 0].cpu(), 1 / torch.sqrt(var + 1e-6)) < 2e-5
    gain, shift = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    w_in, b_in = torch.randn(N2, D, generator=g) * 0.05, torch.randn(N2, generator=g)
    wf = (w_in * gain[None, :]).to(H)
    bf, cs = b_in + w_in @ shift, wf.float().sum(1)
    for epi in (0, 1):
        out = ops.gemm_bf16_ln(xb, wf.cuda(), bf.cuda(), cs.cuda(), ln_row, epilogue=epi, tile=tile, m_valid=mv)[:mv].cpu().double()
        xn = (xb[:mv].cpu().double() - mu[:, None]) / torch.sqrt(var + 1e-6)[:, None]
        ref = xn @ wf.double().T + bf.double()
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        # (the massive channel makes rstd * (acc - mean * colsum) a cancellation of fp32 terms ~200 x the result: its noise rides on top of the rounding)
        assert float((out - ref).abs().max()) < (1.5 * EPS16 + 2e-4) * float(ref.abs().max()) + 4e-5, (epi, tile)


def _ref_attention(q, k, v, B, N, heads):
    D = q.shape[1]
    sh = lambda t: t.double().reshape(B, N, heads, D // heads).transpose(1, 2)
    p = torch.softmax(sh(q) @ sh(k).transpose(-1, -2) / (D // heads) ** 0.5, dim=-1)
    return (p @ sh(v)).transpose(1, 2).reshape(B * N, D)


@pytest.mark.parametrize("B,N,heads", [(2, 77, 2), (1, 1374, 4), (3, 905, 2), (1, 256, 1), (2, 257, 3), (1, 64, 16)])
def test_attention_f16_vs_fp64_and_bf16(B, N, heads):
    """fp16 q | k | v, P and output against fp64 on the same operands; the same rows through the bf16 kernel (operands rounded to bf16) are
    several times further from the fp32 values' own attention."""
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(B * N + heads)
    D = 64 * heads
    x32 = torch.randn(B * N, 3 * D, generator=g)
    x32[:, :D] *= 2.0     # peaked scores: the softmax is not flat
    e = {}
    for dt in (H, torch.bfloat16):
        x = x32.to(dt)
        out = ops.attention(x.cuda(), B, N, D, heads).cpu()
        assert out.dtype == dt
        ref_same = _ref_attention(x[:, :D], x[:, D:2 * D], x[:, 2 * D:], B, N, heads)
        ref32 = _ref_attention(x32[:, :D], x32[:, D:2 * D], x32[:, 2 * D:], B, N, heads)
        e[dt] = (float((out.double() - ref_same).abs().max() / ref_same.abs().max()), float((out.double() - ref32).abs().max() / ref32.abs().max()))
    assert e[H][0] < 4 * EPS16, e            # P and the output each round once to fp16
    assert e[H][1] < e[torch.bfloat16][1] / 3, e


def test_attention_f16_lazy_rescale_staircase():
    """A score staircase that forces the lazy rescale many times stays at the fp16 bar (p <= 2^8 fits the format)."""
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(5)
    B, N, heads = 2, 300, 2
    D = 64 * heads
    x = torch.randn(B * N, 3 * D, generator=g)
    x[:, D:D + 1] += torch.arange(B * N, dtype=torch.float32)[:, None] * 0.05    # keys that grow along the sequence: the running maximum keeps moving
    x[:, 0:1] = 8.0
    xh = x.to(H).cuda()
    full = ops.attention(xh, B, N, D, heads)
    ref = _ref_attention(xh.cpu()[:, :D], xh.cpu()[:, D:2 * D], xh.cpu()[:, 2 * D:], B, N, heads)
    assert float((full.cpu().double() - ref).abs().max() / ref.abs().max()) < 4 * EPS16


def _mk(arch, name, sd, prec):
    from foundpose_amd import feature_util
    return feature_util.make_feature_extractor(name, state_dict=sd, arch=arch if arch.name not in ARCHS else None, precision=prec).to("cuda")


@pytest.mark.parametrize("arch,layer,size,B", [(TINY, 2, 56, 3), (ARCHS["vits14-reg"], 9, 224, 2), (ARCHS["vitl14-reg"], 18, 518, 1)])
def test_extractor_f16_vs_oracle_a_next_to_bf16(arch, layer, size, B):
    """Features of the f16 mode against the fp32 CPU oracle (the reference's arithmetic): recorded bar, and several times closer than the bf16 mode on
    the same weights and images (11 significant bits per operand against 8)."""
    name = f"dinov2_version={arch.name}_stride=14_facet=token_layer={layer}_norm=1"
    sd = synthetic.make_vit_state_dict(arch, seed=5)
    imgs = synthetic.make_crops(B, size, seed=2)
    ref = ov.extractor_forward(sd, arch, imgs, layer, True)
    e = {}
    for prec in ("bf16", "f16"):
        ex = _mk(arch, name, sd, prec)
        o = ex(imgs.cuda())
        e[prec] = rel_err(o["feature_maps"].cpu(), ref["feature_maps"])
        assert rel_err(o["cls_tokens"].cpu(), ref["cls_tokens"]) < 20 * e[prec] + 1e-3
        if prec == "f16":
            assert ex.fold_layernorm and ex.saturation_counts() == (0, 0)
    print(f"\n{arch.name}@{size} layer {layer}: f16 {e['f16']:.3e}  bf16 {e['bf16']:.3e} of the feature scale vs oracle A")
    check_bar(f"f16_{arch.name}_{size}/vs_oracle_a", e["f16"], 2e-3)
    assert e["f16"] < e["bf16"] / 4, e


def test_extractor_f16_batch_invariance_token_selection_and_swiglu():
    """A crop's features do not depend on the batch it rides in; the engine's token-selected hooked block gives the sampled features of the full
    forward bit for bit; the SwiGLU FFN (ViT-g's) runs in the mode."""
    from foundpose_amd import engine as fe, feature_util, workload
    from foundpose_amd.bank import DeviceBank
    from foundpose_amd.vit_config import VitArch
    TINY_G = VitArch("tinyg-reg", dim=128, depth=2, heads=2, ffn="swiglu", hidden=384, registers=4, pretrain_grid=4, interp_antialias=True, interp_offset=0.0)
    arch = ARCHS["vits14-reg"]
    name = "dinov2_version=vits14-reg_stride=14_facet=token_layer=5_norm=1"
    sd = synthetic.make_vit_state_dict(arch, seed=4)
    ex = feature_util.make_feature_extractor(name, state_dict=sd, precision="f16").to("cuda")
    imgs = synthetic.make_crops(5, 224, seed=1).cuda()
    one = ex(imgs[3:4])["feature_maps"].clone()
    assert torch.equal(ex(imgs)["feature_maps"][3], one[0])
    assert ex.supports_token_selection
    wl = workload.build_planted_workload(ex, 4, 224, 1, 60, seed=3, crop_seed=2)
    bank = DeviceBank(wl.repres)
    outs = [fe.FoundPoseEngine(ex, bank, 14.0, 5, 300, tie_order="torch", select_tokens=sel).infer_batch(wl.crops, wl.masks, wl.det_obj, keep_debug=True)
            for sel in (True, False)]
    for f in ("template_ids", "template_scores", "counts", "q_ids", "feat_ids", "dists", "conf", "coord_2d", "coord_3d", "query_tfidf"):
        x, y = getattr(outs[0], f), getattr(outs[1], f)
        assert torch.equal(x, y) or bool(((x == y) | (x.isnan() & y.isnan())).all()), f
    # SwiGLU architecture
    sdg = synthetic.make_vit_state_dict(TINY_G, seed=7)
    nm = f"dinov2_version={TINY_G.name}_stride=14_facet=token_layer=1_norm=1"
    im = synthetic.make_crops(2, 70, seed=3)
    ref = ov.extractor_forward(sdg, TINY_G, im, 1, True)["feature_maps"]
    got = feature_util.make_feature_extractor(nm, state_dict=sdg, arch=TINY_G, precision="f16").to("cuda")(im.cuda())["feature_maps"].cpu()
    assert rel_err(got, ref) < 2e-3


def test_extractor_f16_with_massive_activation_channels():
    """Residual channels hundreds of times larger than the rest and hidden units in the hundreds (what real checkpoints carry) sit far inside the fp16
    range: nothing is reported, and the features stay several times closer to the fp32 oracle than the bf16 mode's."""
    arch = ARCHS["vits14-reg"]
    name = "dinov2_version=vits14-reg_stride=14_facet=token_layer=9_norm=1"
    sd = {k: v.clone() for k, v in synthetic.make_vit_state_dict(arch, seed=5).items()}
    sd["blocks.0.mlp.fc2.weight"][17] *= 400.0
    sd["blocks.1.mlp.fc1.weight"][33] *= 60.0
    sd["blocks.1.mlp.fc1.bias"][33] += 150.0
    imgs = synthetic.make_crops(2, 224, seed=2)
    ref = ov.extractor_forward(sd, arch, imgs, 9, True)["feature_maps"]
    e = {}
    for prec in ("bf16", "f16"):
        ex = _mk(arch, name, sd, prec)
        out = ex(imgs.cuda())["feature_maps"].cpu()
        assert bool(torch.isfinite(out).all())
        e[prec] = rel_err(out, ref)
    assert ex.saturation_counts() == (0, 0)
    assert e["f16"] < e["bf16"] / 3, e


@pytest.mark.parametrize("where", ["hidden", "qkv", "stream", "nan"])
def test_f16_overflow_is_loud(where):
    """fp16 has bf16's speed, not its range.  A 16-bit activation beyond +-65504 becomes inf, poisons the token's residual stream and -- through the
    keys and values of the next attention -- the whole image: the last kernel of the pipeline finds non-finite features and REPORTS them (a NaN
    likewise): the device-side counter goes up, the extractor's forward raises FoundPoseSaturationError and keeps raising until reset_saturation(), a
    result of the batched engine raises when read; the bf16 mode computes the same weights without complaint."""
    from foundpose_amd import engine as fe, feature_util, workload
    from foundpose_amd.bank import DeviceBank
    arch = ARCHS["vits14-reg"]
    name = "dinov2_version=vits14-reg_stride=14_facet=token_layer=3_norm=1"
    sd = {k: v.clone() for k, v in synthetic.make_vit_state_dict(arch, seed=5).items()}
    if where == "hidden":
        sd["blocks.0.mlp.fc1.bias"][11] = 70000.0              # gelu(70000) = 70000 > 65504
    elif where == "qkv":
        sd["blocks.2.attn.qkv.bias"][arch.dim + 3] = -80000.0  # a key channel
    elif where == "stream":
        sd["blocks.1.mlp.fc2.bias"][9] = 1.0e5                 # a residual channel (the (hi, lo) stream's high half)
        sd["blocks.1.ls2.gamma"][9] = 1.0
    ex = feature_util.make_feature_extractor(name, state_dict=sd, precision="f16")
    if where == "nan":
        sd["blocks.1.attn.qkv.bias"][2 * arch.dim + 5] = float("nan")   # planted behind the load-time validation
    ex = ex.to("cuda")
    imgs = synthetic.make_crops(3, 112, seed=2).cuda()
    with pytest.raises(_lib.FoundPoseSaturationError, match="beyond the fp16 range"):   # (the report comes from the final norm: non-finite features)
        ex(imgs)
    assert ex.saturation_counts()[0] > 0
    with pytest.raises(_lib.FoundPoseSaturationError):
        ex.check_saturation()
    ex.reset_saturation()
    if where != "nan":
        exb = feature_util.make_feature_extractor(name, state_dict=sd, precision="bf16").to("cuda")
        assert bool(torch.isfinite(exb(imgs)["feature_maps"]).all())
    clean = feature_util.make_feature_extractor(name, random_init_seed=5, precision="f16").to("cuda")
    wl = workload.build_planted_workload(clean, 3, 112, 1, 40, seed=3, crop_seed=2)
    bank = DeviceBank(wl.repres)
    res = fe.FoundPoseEngine(clean, bank, 14.0, 5, 300, tie_order="torch").infer_batch(wl.crops, wl.masks, wl.det_obj)
    assert len(res.corresp_list(0)) == 5 and clean.saturation_counts() == (0, 0)
    res = fe.FoundPoseEngine(ex, bank, 14.0, 5, 300, tie_order="torch").infer_batch(wl.crops, wl.masks, wl.det_obj)
    with pytest.raises(_lib.FoundPoseSaturationError):
        res.corresp_list(1)


def test_f16_mode_loud_failures():
    from foundpose_amd import feature_util
    from foundpose_amd.vit_config import VitArch
    odd = VitArch("tiny-odd", dim=192, depth=2, heads=3, ffn="mlp", hidden=768, registers=0, pretrain_grid=4, interp_antialias=False, interp_offset=0.1)
    with pytest.raises(NotImplementedError, match="multiple of 128"):
        feature_util.make_feature_extractor("dinov2_version=tiny-odd_stride=14_facet=token_layer=1_norm=1", random_init_seed=1, arch=odd, precision="f16")
    from foundpose_amd import ops
    a, w = torch.zeros(128, 64, dtype=H, device="cuda"), torch.zeros(128, 64, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError, match="two bf16 or two fp16"):
        ops.gemm_bf16(a, w, torch.zeros(128, device="cuda"))


@pytest.mark.parametrize("head,tail", [("f16", "f16x3"), ("f16f8", "f16")])
def test_precision_schedule_runs_two_models_over_one_stream(head, tail):
    """head_blocks = k: blocks 0..k-1 in one precision, blocks k..layer in another, over one fp32 stream (fp_vit_stream_f32 / fp_vit_forward_blocks) -- a
    measurement device (tools/schedule_sweep.py), not a shipped default.  The composition is exact at its seams: with the SAME precision on both sides
    of the cut (fp32 | fp32) the features equal the single model's to the last bit; mixed precisions land between the two pure modes' distances from the fp32
    oracle; the engine's token-selected path equals the full path bit for bit."""
    from foundpose_amd import engine as fe, feature_util, workload
    from foundpose_amd.bank import DeviceBank
    arch = ARCHS["vits14-reg"]
    name = "dinov2_version=vits14-reg_stride=14_facet=token_layer=6_norm=1"
    sd = synthetic.make_vit_state_dict(arch, seed=4)
    imgs = synthetic.make_crops(3, 224, seed=1)
    ref = ov.extractor_forward(sd, arch, imgs, 6, True)["feature_maps"]
    mk = lambda prec, **kw: feature_util.make_feature_extractor(name, state_dict=sd, precision=prec, **kw).to("cuda")
    whole = mk("fp32")(imgs.cuda())["feature_maps"]
    cut = mk("fp32", head_blocks=3, head_precision="fp32")(imgs.cuda())["feature_maps"]
    assert torch.equal(whole, cut)
    e = {p: rel_err(mk(p)(imgs.cuda())["feature_maps"].cpu(), ref) for p in (head, tail)}
    ex = mk(tail, head_blocks=3, head_precision=head)
    got = rel_err(ex(imgs.cuda())["feature_maps"].cpu(), ref)
    assert got < 1.5 * max(e.values()) + 1e-5, (got, e)
    wl = workload.build_planted_workload(mk("fp32"), 3, 224, 1, 60, seed=3, crop_seed=2)
    bank = DeviceBank(wl.repres)
    outs = [fe.FoundPoseEngine(ex, bank, 14.0, 5, 300, tie_order="torch", select_tokens=sel).infer_batch(wl.crops, wl.masks, wl.det_obj) for sel in (True, False)]
    for f in ("template_ids", "counts", "q_ids", "feat_ids", "dists", "coord_3d"):
        x, y = getattr(outs[0], f), getattr(outs[1], f)
        assert torch.equal(x, y) or bool(((x == y) | (x.isnan() & y.isnan())).all()), f


def test_f16_weight_matrices_carry_power_of_two_scales():
    """LayerScale is folded into proj / fc2 (fp_vit_model.ln_fold): with the small gammas DINOv2 starts from, diag(gamma) W would sit in fp16's subnormal range,
    where the format has FEWER significant bits than bf16.  Every folded matrix is therefore stored times a power of two that brings its largest entry to
    [2^13, 2^14] and the epilogue undoes it exactly (fp_vit_block.act_scale): with gammas of 1e-3 the mode stays several times closer to the fp32 oracle than
    bf16, and the stored matrices are where they should be."""
    arch = ARCHS["vits14-reg"]
    name = "dinov2_version=vits14-reg_stride=14_facet=token_layer=5_norm=1"
    sd = {k: v.clone() for k, v in synthetic.make_vit_state_dict(arch, seed=5).items()}
    for i in range(arch.depth):
        sd[f"blocks.{i}.ls1.gamma"] *= 1e-3
        sd[f"blocks.{i}.ls2.gamma"] *= 1e-3
        sd[f"blocks.{i}.attn.proj.weight"] *= 30.0      # (so that the blocks still move the stream: contributions ~3e-2 of what they were)
        sd[f"blocks.{i}.mlp.fc2.weight"] *= 30.0
    imgs = synthetic.make_crops(2, 224, seed=2)
    ref = ov.extractor_forward(sd, arch, imgs, 5, True)["feature_maps"]
    e = {}
    for prec in ("bf16", "f16"):
        ex = _mk(arch, name, sd, prec)
        e[prec] = rel_err(ex(imgs.cuda())["feature_maps"].cpu(), ref)
    assert e["f16"] < e["bf16"] / 3, e
    for j, key in enumerate(("qkv.wf", "proj.wf", "fc1.wf", "fc2.wf")):
        wmax = float(ex._w["blocks.2." + key].float().abs().max())
        assert 2.0 ** 13 <= wmax <= 2.0 ** 14, (key, wmax)
        inv = float(ex._blocks[2].act_scale[j])
        assert inv > 0 and abs(np.log2(inv) - round(np.log2(inv))) < 1e-9       # a power of two
    assert float(ex._blocks[2].act_scale[1]) < 2.0 ** -15                       # proj: gamma 1e-3 x W 0.6 -> scaled up by >= 2^15
