"""GPU parity: ViT kernels (through the C ABI) vs a plain fp32 torch reference of the same op, and the
extractor vs the oracle / the reference-wrapper fixtures.

Tolerances: fp32 kernels 1e-4 relative (accumulation order); bf16 kernels vs a reference fed the SAME
bf16-rounded operands: 2^-8 relative of the output scale (one bf16 rounding of the result) unless noted.
"""

import numpy as np
import pytest
import torch

from foundpose_amd import _lib, synthetic
from foundpose_amd.vit_config import ARCHS
from oracle import vit as ov
from tests.helpers import NOREG_CASES, TINY, TINY0, assert_features_close, check_bar, load_golden, noreg_case

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_layernorm():
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(0)
    for D in (128, 384, 1024, 1536):
        x = torch.randn(300, D, generator=g) * 3 + 1
        w, b = torch.randn(D, generator=g), torch.randn(D, generator=g)
        ref = torch.nn.functional.layer_norm(x, (D,), w, b, eps=1e-6)
        y32 = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), torch.float32).cpu()
        assert rel_err(y32, ref) < 2e-6
        y16 = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), torch.bfloat16).cpu()
        assert torch.equal(y16, y32.to(torch.bfloat16)) or rel_err(y16.float(), ref) < 2 ** -8


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 1024), (1408, 3072, 1024), (384, 1024, 4096), (128, 128, 640),
                                   (512, 256, 128), (1536, 1024, 1024), (256, 256, 192)])
def test_gemm_bf16_epilogues(M, N, K, tile):
    from foundpose_amd import ops as _ops
    if tile == 256 and (M % 256 or N % 256):
        pytest.skip("256 tile needs M, N multiples of 256")

    class ops:  # force the block tile through the tuning bits of the epilogue argument
        @staticmethod
        def gemm_bf16(a, w, bias, gamma=None, out=None, epilogue=0, m_valid=None):
            return _ops.gemm_bf16(a, w, bias, gamma=gamma, out=out, epilogue=epilogue | (tile << 8), m_valid=m_valid)

    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    gamma = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().T + bias.double()
    scale = float(ref.abs().max())
    out = ops.gemm_bf16(a.cuda(), w.cuda(), bias.cuda(), epilogue=5).cpu()  # bias -> f32
    assert float((out.double() - ref).abs().max()) < 2e-5 * scale * max(1, K / 1024)
    out = ops.gemm_bf16(a.cuda(), w.cuda(), bias.cuda(), epilogue=0).cpu()  # bias -> bf16
    assert float((out.double() - ref).abs().max()) < 2 ** -8 * scale
    out = ops.gemm_bf16(a.cuda(), w.cuda(), bias.cuda(), epilogue=1).cpu()  # gelu -> bf16
    gref = torch.nn.functional.gelu(ref)
    assert float((out.double() - gref).abs().max()) < 2 ** -8 * scale
    x = resid.clone().cuda()
    ops.gemm_bf16(a.cuda(), w.cuda(), bias.cuda(), gamma=gamma.cuda(), out=x, epilogue=3)  # x += gamma * (.)
    rref = resid.double() + gamma.double() * ref
    assert float((x.cpu().double() - rref).abs().max()) < 3e-5 * float(rref.abs().max()) * max(1, K / 1024)
    # M_valid: rows beyond it must be left untouched
    out = torch.full((M, N), 7.0, dtype=torch.float32).cuda()
    ops.gemm_bf16(a.cuda(), w.cuda(), bias.cuda(), out=out, epilogue=5, m_valid=M - 5)
    assert torch.all(out[M - 5:] == 7.0) and torch.all(out[:M - 5] != 7.0)


@pytest.mark.parametrize("M,N,K", [(100, 70, 64), (300, 256, 1024), (37, 130, 36)])
def test_gemm_f32_exact_chain(M, N, K):
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(1)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    out = ops.gemm_f32(a.cuda(), w.cuda(), epilogue=0).cpu()
    assert rel_err(out, a.double() @ w.double().T) < 1e-5
    # bit-exact vs a k-ordered fmaf chain (same property the distance kernels rely on)
    from oracle import clib
    chain = np.stack([clib.dot_rows(w.numpy(), a[i].numpy()) for i in range(min(M, 8))])
    assert np.array_equal(out[:chain.shape[0]].numpy(), chain)


@pytest.mark.parametrize("B,N,heads", [(2, 77, 2), (1, 1374, 6), (3, 905, 2), (2, 64, 1), (1, 130, 16)])
def test_attention_bf16_and_f32(B, N, heads):
    from foundpose_amd import ops
    D = heads * 64
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn(B * N, 3 * D, generator=g) * 1.5
    def ref_attn(x):
        q, k, v = x.double().reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
        p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1)
        return (p @ v).transpose(1, 2).reshape(B * N, D)
    o32 = ops.attention(qkv.cuda(), B, N, D, heads).cpu()
    assert rel_err(o32, ref_attn(qkv)) < 1e-5
    q16 = qkv.to(torch.bfloat16)
    o16 = ops.attention(q16.cuda(), B, N, D, heads).cpu()
    ref = ref_attn(q16.float())
    # P is rounded to bf16 before P@V and the output to bf16: a few 2^-8 of the output scale
    assert float((o16.double() - ref).abs().max()) < 3 * 2 ** -8 * float(ref.abs().max())


@pytest.mark.parametrize("B,N,heads", [(1, 1374, 3), (2, 905, 2), (3, 77, 1), (1, 64, 2), (2, 129, 4), (1, 700, 16), (2, 1, 1), (1, 5, 2), (1, 128, 1)])
def test_attention_f32_mfma_kernel_vs_fp64_and_valu_kernel(B, N, heads):
    """The exact-fp32 mode's attention runs on the fp32 MFMA (variant 0); the thread-per-query VALU kernel (variant 1: one fma chain per
    score, keys in order) is its cross-check.  Both at fp32 rounding level of the fp64 result, ragged tails and idle waves included;
    a dominant key late in the sequence forces the rescale path."""
    from foundpose_amd import ops
    D = heads * 64
    g = torch.Generator().manual_seed(N * 7 + heads)
    qkv = torch.randn(B * N, 3 * D, generator=g) * 1.5
    if N > 8:
        qkv[N - 3, D:D + 64] = qkv[5, 0:64] * 3.0   # key N-3 of image 0 / head 0 aligned with query 5
    q, k, v = qkv.double().reshape(B, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(B * N, D)
    o_m = ops.attention(qkv.cuda(), B, N, D, heads).cpu()
    o_v = ops.attention(qkv.cuda(), B, N, D, heads, variant=1).cpu()
    assert rel_err(o_m, ref) < 1e-5 and rel_err(o_v, ref) < 1e-5
    assert float((o_m.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("step", [0.9, 2.5, 5.0, 12.0])
def test_attention_bf16_lazy_rescale_staircase(step):
    """The exponent's reference moves only when a tile's maximum exceeds it by more than 8 (csrc/attn.hip, lazy rescale).  Keys whose
    scores climb by `step` (in the exponent, per 64-key tile) walk through every regime: never more than 8 above the reference for long
    stretches (p up to 2^8, no rescale), a rescale every few tiles, a rescale every tile; queries of different gain see different regimes
    in one wave, so lanes that move and lanes that do not share the rescale branch.  The work splits (0 and its cross-check 1; 2 and 3 in FP_EXPERIMENTS
    builds) stay bit-identical."""
    from foundpose_amd import ops
    N, D, heads = 64 * 9 + 17, 64, 1
    g = torch.Generator().manual_seed(int(step * 10))
    qkv = torch.randn(N, 3 * D, generator=g) * 0.3
    gain = torch.tensor([1.0, 2.0, 4.0, 0.5])[torch.arange(N) % 4]
    qkv[:, 0] = gain * 4.0                                             # q: dimension 0 carries the gain
    tile = (torch.arange(N) // 64).float()
    qkv[:, D] = tile * step / (4.0 * 4.0 * 0.125 * 1.4426950408889634)  # k: + `step` in the exponent per tile for the gain-4 queries
    q16 = qkv.to(torch.bfloat16)
    q, k, v = q16.double().reshape(1, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(N, D)
    from tests.helpers import experiments_build
    outs = [ops.attention(q16.cuda(), 1, N, D, heads, variant=v_).clone() for v_ in ((0, 1, 2, 3) if experiments_build() else (0, 1))]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(outs[0].view(torch.int16), o.view(torch.int16))
    assert float((outs[0].cpu().double() - ref).abs().max()) < 3 * 2 ** -8 * float(ref.abs().max())


@pytest.mark.parametrize("B,N,heads", [(2, 77, 2), (1, 1374, 8), (3, 905, 2), (1, 256, 1), (2, 257, 4), (1, 321, 16)])
def test_attention_bf16_work_splits_agree_bitwise(B, N, heads):
    """The 64-queries-per-wave kernel (LDS-DMA staging) and the 32-queries-per-wave kernel issue the same MFMAs in the
    same order for every query: their outputs must be bit-identical (ragged tails, idle waves, both block mappings)."""
    from foundpose_amd import ops
    D = heads * 64
    g = torch.Generator().manual_seed(N + heads)
    q16 = (torch.randn(B * N, 3 * D, generator=g) * 1.5).to(torch.bfloat16).cuda()
    o_a = ops.attention(q16, B, N, D, heads, variant=1).clone()   # 32 queries per wave, register staging
    o_b = ops.attention(q16, B, N, D, heads, variant=0).clone()   # the pipeline's kernel
    assert torch.equal(o_a.view(torch.int16), o_b.view(torch.int16))
    from tests.helpers import experiments_build
    if experiments_build():   # the measured-slower work splits of FP_EXPERIMENTS builds
        o_c = ops.attention(q16, B, N, D, heads, variant=2)           # the DMA kernel with one 32-query block per wave, 8 waves per block
        o_d = ops.attention(q16, B, N, D, heads, variant=3)           # 8 waves x 64 queries: 512-query blocks
        assert torch.equal(o_a.view(torch.int16), o_d.view(torch.int16)) and torch.equal(o_a.view(torch.int16), o_c.view(torch.int16))
    else:
        with pytest.raises(_lib.FoundPoseNativeError, match="FP_EXPERIMENTS builds only"):
            ops.attention(q16, B, N, D, heads, variant=2)


@pytest.mark.parametrize("precision,tol", [("fp32", 3e-5), ("f16x3", 3e-5), ("bf16", 3e-2)])
def test_extractor_key_query_value_facets_vs_reference_wrapper_fixture(precision, tol):
    g = load_golden("extractor_tiny_facets")
    imgs = synthetic.make_crops(2, 56, seed=int(g["image_seed"])).cuda()
    for facet in ("key", "query", "value"):
        for layer, norm in ((1, 1), (2, 0)):
            ex = _extractor(TINY, f"dinov2_version=tiny-reg_stride=14_facet={facet}_layer={layer}_logbin=0_norm={norm}", int(g["weights_seed"]), precision)
            o = ex(imgs)
            ref = g[f"fmap_{facet}_l{layer}_n{norm}"]
            assert o["feature_maps"].shape == (2, 128, 4, 4)
            assert_features_close(f"tiny_facet_{facet}_l{layer}_n{norm}/{precision}/fmap", o["feature_maps"].cpu().numpy(), ref, np.abs(ref).max(), tol, precision in ("bf16", "fp8"))
            assert_features_close(f"tiny_facet_{facet}_l{layer}_n{norm}/{precision}/cls", o["cls_tokens"].cpu().numpy(), g[f"cls_{facet}_l{layer}_n{norm}"], np.abs(ref).max(), tol, precision in ("bf16", "fp8"))


def _extractor(arch, name, seed, precision):
    from foundpose_amd import feature_util
    ex = feature_util.make_feature_extractor(name, random_init_seed=seed, precision=precision, arch=arch if arch in (TINY, TINY0) else None)
    return ex.to("cuda")


@pytest.mark.parametrize("precision,tol", [("fp32", 3e-5), ("f16x3", 3e-5), ("bf16", 6e-2)])
def test_extractor_tiny_vs_reference_wrapper_fixture(precision, tol):
    g = load_golden("extractor_tiny")
    imgs = synthetic.make_crops(2, 56, seed=int(g["image_seed"])).cuda()
    for layer, norm in ((1, 1), (2, 1), (0, 0)):
        ex = _extractor(TINY, f"dinov2_version=tiny-reg_stride=14_facet=token_layer={layer}_logbin=0_norm={norm}", int(g["weights_seed"]), precision)
        o = ex(imgs)
        fm = o["feature_maps"]
        assert fm.shape == (2, 128, 4, 4) and not fm.is_contiguous()  # a permuted view, like the reference
        ref = g[f"fmap_l{layer}_n{norm}"]
        assert_features_close(f"tiny_l{layer}_n{norm}/{precision}/fmap", fm.cpu().numpy(), ref, np.abs(ref).max(), tol, precision in ("bf16", "fp8"))
        assert_features_close(f"tiny_l{layer}_n{norm}/{precision}/cls", o["cls_tokens"].cpu().numpy(), g[f"cls_l{layer}_n{norm}"], np.abs(ref).max(), tol, precision in ("bf16", "fp8"))


def test_extractor_tiny_bf16_vs_quantisation_aware_oracle():
    """bf16 path vs oracle B (operands rounded to bf16 at the kernel's cast points): much tighter than vs fp32."""
    sd = synthetic.make_vit_state_dict(TINY, seed=1234)
    imgs = synthetic.make_crops(2, 56, seed=0)
    ex = _extractor(TINY, "dinov2_version=tiny-reg_stride=14_facet=token_layer=2_logbin=0_norm=1", 1234, "bf16")
    fm = ex(imgs.cuda())["feature_maps"].cpu()
    ref_b = ov.extractor_forward(sd, TINY, imgs, 2, True, quant="bf16")["feature_maps"]
    ref_a = ov.extractor_forward(sd, TINY, imgs, 2, True)["feature_maps"]
    eb, ea = rel_err(fm, ref_b), rel_err(fm, ref_a)
    check_bar("tiny_l2/bf16/vs_oracle_b", eb, 1.5e-2)
    assert eb < ea, (eb, ea)     # the quantisation-aware oracle is the closer one


@pytest.mark.parametrize("precision,tol", [("fp32", 5e-5), ("f16x3", 5e-5), ("bf16", 8e-2)])
def test_extractor_vits14reg_518_vs_reference_wrapper_fixture(precision, tol):
    g = load_golden("extractor_vits14reg_518")
    imgs = synthetic.make_crops(1, 518, seed=int(g["image_seed"])).cuda()
    ex = _extractor(None, "dinov2_version=vits14-reg_stride=14_facet=token_layer=9_logbin=0_norm=1", int(g["weights_seed"]), precision)
    o = ex(imgs)
    fm = o["feature_maps"].cpu().numpy()
    assert fm.shape == (1, 384, 37, 37)
    scale = np.abs(g["fmap_sub"]).max()
    assert_features_close(f"vits14reg_518/{precision}/fmap", fm[:, ::8, ::3, ::3], g["fmap_sub"], scale, tol, precision in ("bf16", "fp8"))
    assert_features_close(f"vits14reg_518/{precision}/cls", o["cls_tokens"].cpu().numpy(), g["cls"], scale, tol, precision in ("bf16", "fp8"))


@pytest.mark.parametrize("precision,tol", [("fp32", 5e-5), ("f16x3", 5e-5), ("bf16", 8e-2)])
def test_extractor_vits14reg_420_vs_reference_wrapper_fixture(precision, tol):
    """BASELINE config 1's geometry (the reference's shipped LM-O options, configs/infer/lmo.json:6-12): 420 x 420 crops take the
    pos-embed table interpolated 37 x 37 -> 30 x 30 with the `-reg` hub flags (size mode, bicubic, antialias = True, offset 0).
    The fixture is the reference wrapper's own output at that size."""
    g = load_golden("extractor_vits14reg_420")
    imgs = synthetic.make_crops(1, 420, seed=int(g["image_seed"])).cuda()
    ex = _extractor(None, "dinov2_version=vits14-reg_stride=14_facet=token_layer=9_logbin=0_norm=1", int(g["weights_seed"]), precision)
    o = ex(imgs)
    fm = o["feature_maps"].cpu().numpy()
    assert fm.shape == (1, 384, 30, 30)
    scale = np.abs(g["fmap_sub"]).max()
    assert_features_close(f"vits14reg_420/{precision}/fmap", fm[:, ::4, ::2, ::2], g["fmap_sub"], scale, tol, precision in ("bf16", "fp8"))
    assert_features_close(f"vits14reg_420/{precision}/cls", o["cls_tokens"].cpu().numpy(), g["cls"], scale, tol, precision in ("bf16", "fp8"))


@pytest.mark.parametrize("precision,tol", [("fp32", 3e-5), ("f16x3", 3e-5), ("bf16", 6e-2)])
def test_extractor_tiny_noreg_vs_reference_wrapper_fixture(precision, tol):
    """The NON-register layout (prefix of one row, N = 1 + Np) and the non-register hub entries' pos-embed interpolation (scale-factor
    mode, +0.1 offset, no antialias) on the native, a larger square and a non-square grid: vs the reference wrapper over the
    non-register stand-in backbone (oracle/make_golden.py::gen_extractor_noreg)."""
    g = load_golden("extractor_tiny_noreg")
    for (H, W) in ((56, 56), (84, 84), (70, 42)):
        imgs = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(H * 1000 + W)).cuda()
        for layer, norm in ((2, 1), (0, 0)):
            ex = _extractor(TINY0, f"dinov2_version=tiny_stride=14_facet=token_layer={layer}_logbin=0_norm={norm}", int(g["weights_seed"]), precision)
            assert ex.arch.registers == 0
            o = ex(imgs)
            ref = g[f"fmap_{H}x{W}_l{layer}_n{norm}"]
            assert o["feature_maps"].shape == ref.shape
            assert_features_close(f"tiny_noreg_{H}x{W}_l{layer}_n{norm}/{precision}/fmap", o["feature_maps"].cpu().numpy(), ref, np.abs(ref).max(), tol, precision in ("bf16", "fp8"))
            assert_features_close(f"tiny_noreg_{H}x{W}_l{layer}_n{norm}/{precision}/cls", o["cls_tokens"].cpu().numpy(), g[f"cls_{H}x{W}_l{layer}_n{norm}"], np.abs(ref).max(), tol, precision in ("bf16", "fp8"))


@pytest.mark.parametrize("precision,tol", [("fp32", 5e-5), ("f16x3", 5e-5), ("bf16", 8e-2)])
@pytest.mark.parametrize("version,S", NOREG_CASES)
def test_extractor_noreg_hub_archs_vs_reference_wrapper_fixture(version, S, precision, tol):
    if version == "vitg14" and precision == "f16x3":
        pytest.skip("22 s for a case the fp32 / bf16 runs of the same architecture and five other f16x3 cases cover (GPU suite budget)")
    """The reference's DEFAULT backbone family (InferOpts.extractor_name = "dinov2_vitl14", scripts/infer.py:75: no register tokens,
    short form -> layer 9, dinov2_utils.py:62-64) and its ViT-S / ViT-B siblings (D = 768 / 12 heads) at 518 (N = 1370) and at the
    LM-O crop size 420 (N = 901, interpolated table), against the reference wrapper's own output."""
    g, name, spec, sd, imgs = noreg_case(version, S)
    ex = _extractor(None, name, int(g["weights_seed"]), precision)
    assert ex.arch.registers == 0 and ex.layer == int(g["layer"]) and ex.model_base_name == f"dinov2_{version}"
    o = ex(imgs.cuda())
    fm = o["feature_maps"].cpu().numpy()
    cs, ss = (int(v) for v in g["sub"])
    assert fm.shape == (1, spec.arch.dim, S // 14, S // 14)
    scale = np.abs(g["fmap_sub"]).max()
    assert_features_close(f"noreg_{version}_{S}/{precision}/fmap", fm[:, ::cs, ::ss, ::ss], g["fmap_sub"], scale, tol, precision in ("bf16", "fp8"))
    assert_features_close(f"noreg_{version}_{S}/{precision}/cls", o["cls_tokens"].cpu().numpy(), g["cls"], scale, tol, precision in ("bf16", "fp8"))


@pytest.mark.parametrize("precision,tol", [("fp32", 3e-5), ("f16x3", 5e-5), ("bf16", 6e-2)])
def test_extractor_stride7_vs_reference_fixture(precision, tol):
    """stride != patch size (the reference's patch_vit_resolution / _fix_pos_enc, dinov2_utils.py:313-389): overlapping patches,
    7 x 7 tokens for a 56-px image, the strided position encoding.  Fixture: the reference wrapper + the reference's own position-encoding
    function over the stand-in backbone; plus a non-square image against the oracle (pinned to the same function on CPU)."""
    g = load_golden("extractor_tiny_stride7")
    imgs = synthetic.make_crops(2, 56, seed=int(g["image_seed"])).cuda()
    for layer, norm in ((2, 1), (0, 0)):
        ex = _extractor(TINY, f"dinov2_version=tiny-reg_stride=7_facet=token_layer={layer}_logbin=0_norm={norm}", int(g["weights_seed"]), precision)
        assert ex.stride == 7 and not ex.supports_token_selection
        o = ex(imgs)
        assert o["feature_maps"].shape == (2, TINY.dim, 7, 7)
        scale = np.abs(g[f"fmap_l{layer}_n{norm}"]).max()
        assert_features_close(f"tiny_stride7_l{layer}_n{norm}/{precision}/fmap", o["feature_maps"].cpu().numpy(), g[f"fmap_l{layer}_n{norm}"], scale, tol, precision in ("bf16", "fp8"))
        assert_features_close(f"tiny_stride7_l{layer}_n{norm}/{precision}/cls", o["cls_tokens"].cpu().numpy(), g[f"cls_l{layer}_n{norm}"], scale, tol, precision in ("bf16", "fp8"))
    sd = synthetic.make_vit_state_dict(TINY, seed=int(g["weights_seed"]))
    wide = synthetic.make_crops(1, 70, seed=3)[:, :, :, :56].contiguous()      # 70 x 56: 9 x 7 tokens
    ex = _extractor(TINY, "dinov2_version=tiny-reg_stride=7_facet=token_layer=2_logbin=0_norm=1", int(g["weights_seed"]), precision)
    got = ex(wide.cuda())["feature_maps"].cpu()
    ref = ov.extractor_forward(sd, TINY, wide, 2, True, stride=7)["feature_maps"]
    assert got.shape == (1, TINY.dim, 9, 7)
    assert_features_close(f"tiny_stride7_70x56/{precision}/fmap", got.numpy(), ref.numpy(), float(ref.abs().max()), tol, precision in ("bf16", "fp8"))


def test_extractor_stride_through_the_engine():
    """The batched engine on a strided extractor (token selection switches itself off, the fused norm + sampling reads the finer 15 x 15
    grid) returns what the drop-in single-detection calls return: extractor -> filter_points_by_mask -> sample_feature_map_at_points ->
    project_features -> establish_correspondences, index for index."""
    from foundpose_amd import corresp_util, engine as fe, feature_util, projector_util, workload
    from foundpose_amd.bank import DeviceBank
    name = "dinov2_version=vits14-reg_stride=7_facet=token_layer=3_norm=1"
    ex = feature_util.make_feature_extractor(name, random_init_seed=1234, precision="fp32").to("cuda")
    wl = workload.build_planted_workload(ex, 3, 112, 1, 60, seed=2, crop_seed=1)
    repre = wl.repres[0]
    res = fe.FoundPoseEngine(ex, DeviceBank(wl.repres), 14.0, 5, 300, tie_order="torch").infer_batch(wl.crops, wl.masks, wl.det_obj)
    assert ex.num_patches == (15, 15) and not ex.supports_token_selection
    grid = feature_util.generate_grid_points((112, 112), 14.0).cuda()
    for b in range(3):
        fmap = ex(wl.crops[b:b + 1])["feature_maps"][0]
        assert fmap.shape == (384, 15, 15)
        qp = feature_util.filter_points_by_mask(grid, wl.masks[b])
        qf = feature_util.sample_feature_map_at_points(fmap, qp, (112, 112)).contiguous()
        qfp = projector_util.project_features(qf, repre.feat_raw_projectors).contiguous()
        single = corresp_util.establish_correspondences(qp, qfp, repre, "tfidf", "cyclic_buddies", 5, 300)
        batch = res.corresp_list(b)
        assert len(single) == len(batch) == 5
        for s_, b_ in zip(single, batch):
            assert int(s_["template_id"]) == int(b_["template_id"])
            assert torch.equal(s_["coord_2d_ids"], b_["coord_2d_ids"]) and torch.equal(s_["nn_vertex_ids"], b_["nn_vertex_ids"])


def test_extractor_batch_invariance_and_420():
    """Each image of a batch gets the same features as when run alone; 420x420 crops take the interpolated pos-embed."""
    ex = _extractor(None, "dinov2_version=vits14-reg_stride=14_facet=token_layer=9_logbin=0_norm=1", 1234, "bf16")
    imgs = synthetic.make_crops(3, 420, seed=5).cuda()
    all_ = ex(imgs)["feature_maps"].clone()
    one = ex(imgs[1:2])["feature_maps"]
    assert all_.shape == (3, 384, 30, 30)
    assert torch.equal(all_[1], one[0])
    sd = synthetic.make_vit_state_dict(ARCHS["vits14-reg"], seed=1234)
    ref = ov.extractor_forward(sd, ARCHS["vits14-reg"], imgs[1:2].cpu(), 9, True)["feature_maps"]
    check_bar("vits14reg_420_batch/bf16/vs_oracle_a", rel_err(one.cpu(), ref), 8e-2)


def test_hot_section_composite_fp32():
    """infer.py:468-542 end to end on the MI355X (fp32 mode) vs the reference-run fixture: same templates,
    bit-exact correspondences given the fixture's projected features, features within 1e-4."""
    from foundpose_amd import corresp_util, feature_util, projector_util, repre_util
    g = load_golden("hot_section_tiny")
    S = int(g["image_size"])
    ex = _extractor(TINY, "dinov2_version=tiny-reg_stride=14_facet=token_layer=2_logbin=0_norm=1", int(g["weights_seed"]), "fp32")
    q_img = torch.from_numpy(g["q_img"]).unsqueeze(0).cuda()
    fmap = ex(q_img)["feature_maps"][0]
    scale = np.abs(g["fmap"]).max()
    np.testing.assert_allclose(fmap.cpu().numpy(), g["fmap"], rtol=0, atol=1e-4 * scale)
    grid = feature_util.generate_grid_points((S, S), 14.0).cuda()
    qp = feature_util.filter_points_by_mask(grid, torch.from_numpy(g["tpl_masks"][4]).cuda())
    assert np.array_equal(qp.cpu().numpy(), g["query_points"])
    qf = feature_util.sample_feature_map_at_points(fmap, qp, (S, S)).contiguous()
    np.testing.assert_allclose(qf.cpu().numpy(), g["query_features"], rtol=0, atol=1e-4 * scale)
    proj = projector_util.projector_from_tensordict({"pca_projector": {
        "components": torch.from_numpy(g["pca_components"]), "mean": torch.from_numpy(g["pca_mean"]), "whiten": torch.tensor(False)}})
    qfp = projector_util.project_features(qf, [proj])
    np.testing.assert_allclose(qfp.cpu().numpy(), g["query_features_proj"], rtol=0, atol=2e-4 * np.abs(g["query_features_proj"]).max())
    repre = repre_util.FeatureBasedObjectRepre(
        vertices=torch.from_numpy(g["vertices"]), feat_vectors=torch.from_numpy(g["bank_feats"]),
        feat_to_template_ids=torch.from_numpy(g["f2t"]), feat_cluster_centroids=torch.from_numpy(g["centroids"]),
        feat_cluster_idfs=torch.from_numpy(g["idfs"]), template_descs=torch.from_numpy(g["template_descs"]),
        template_desc_opts=repre_util.TemplateDescOpts(), feat_raw_projectors=[proj])
    corresp = corresp_util.establish_correspondences(qp, qfp, repre, "tfidf", "cyclic_buddies", 5, 300)
    assert [int(c["template_id"]) for c in corresp] == list(g["template_ids"])
    np.testing.assert_allclose([float(c["template_score"]) for c in corresp], g["template_scores"], rtol=0, atol=1e-4)
    # with the fixture's own projected features the correspondence sets are identical (k == Q: no boundary ties)
    corresp2 = corresp_util.establish_correspondences(
        torch.from_numpy(g["query_points"]).cuda(), torch.from_numpy(g["query_features_proj"]).cuda(), repre, "tfidf", "cyclic_buddies", 5, 300)
    for i, c in enumerate(corresp2):
        a = dict(zip(c["coord_2d_ids"].cpu().tolist(), c["nn_vertex_ids"].cpu().tolist()))
        b = dict(zip(g[f"coord_2d_ids_{i}"].tolist(), g[f"nn_vertex_ids_{i}"].tolist()))
        assert a == b


@pytest.mark.parametrize("precision,tol", [("fp32", 5e-5), ("f16x3", 5e-5), ("bf16", 2e-2)])
def test_extractor_swiglu_ffn(precision, tol):
    """ViT-g style blocks (SwiGLU FFN, fused into the w12 GEMM epilogue) vs the oracle."""
    from foundpose_amd import feature_util
    from foundpose_amd.vit_config import VitArch
    arch = VitArch("tinyg-reg", dim=128, depth=2, heads=2, ffn="swiglu", hidden=256, registers=4, pretrain_grid=4,
                   interp_antialias=True, interp_offset=0.0)
    sd = synthetic.make_vit_state_dict(arch, seed=5)
    imgs = synthetic.make_crops(2, 56, seed=3)
    ex = feature_util.make_feature_extractor("dinov2_version=tinyg-reg_stride=14_facet=token_layer=1_norm=1", state_dict=sd,
                                             precision=precision, arch=arch).to("cuda")
    fm = ex(imgs.cuda())["feature_maps"].cpu()
    ref = ov.extractor_forward(sd, arch, imgs, 1, True, quant="bf16" if precision == "bf16" else None)["feature_maps"]
    assert_features_close(f"swiglu_ffn/{precision}/fmap", fm.numpy(), ref.numpy(), float(ref.abs().max()), tol, precision in ("bf16", "fp8"))


def test_vitg_shapes_run():
    """The real ViT-g/14-reg geometry (D=1536, 24 heads, SwiGLU hidden 4096) runs end to end (2 blocks, 1 crop)."""
    from foundpose_amd import feature_util
    ex = feature_util.make_feature_extractor("dinov2_version=vitg14-reg_stride=14_facet=token_layer=1_norm=1", random_init_seed=3).to("cuda")
    fm = ex(synthetic.make_crops(1, 518, seed=0).cuda())["feature_maps"]
    assert fm.shape == (1, 1536, 37, 37) and bool(torch.isfinite(fm).all())


def test_extractor_hipgraph_replay_is_bit_identical():
    """use_graph=True replays the same launch sequence as one hipGraph: outputs equal the eager launches bit for bit,
    also on the second replay with new pixels in the static input buffer."""
    from foundpose_amd import feature_util
    name = "dinov2_version=vits14-reg_stride=14_facet=token_layer=9_norm=1"
    eager = feature_util.make_feature_extractor(name, random_init_seed=11).to("cuda")
    graph = feature_util.make_feature_extractor(name, random_init_seed=11, use_graph=True).to("cuda")
    for seed in (0, 1, 2):
        imgs = synthetic.make_crops(3, 224, seed=seed).cuda()
        a, b = eager(imgs), graph(imgs)
        assert torch.equal(a["feature_maps"], b["feature_maps"]) and torch.equal(a["cls_tokens"], b["cls_tokens"])


def test_extractor_vitl14reg_518_metric_config_vs_oracle():
    """The bench configuration (ViT-L/14-reg, layer 18, 518x518): fp32 parity mode and the bf16 path against the CPU
    oracle on one crop, plus batch invariance of the bf16 path at the metric's batch of 32 (crop 5 of the batch ==
    that crop run alone, bit for bit)."""
    arch = ARCHS["vitl14-reg"]
    name = "dinov2_version=vitl14-reg_stride=14_facet=token_layer=18_norm=1"
    sd = synthetic.make_vit_state_dict(arch, seed=1234)
    imgs = synthetic.make_crops(32, 518, seed=0)
    ref = ov.extractor_forward(sd, arch, imgs[5:6], 18, True)["feature_maps"]
    from foundpose_amd import feature_util
    ex32 = feature_util.make_feature_extractor(name, state_dict=sd, precision="fp32").to("cuda")
    f32 = ex32(imgs[5:6].cuda())["feature_maps"].cpu()
    assert f32.shape == (1, 1024, 37, 37)
    assert rel_err(f32, ref) < 5e-5
    del ex32
    ex16 = feature_util.make_feature_extractor(name, state_dict=sd, precision="bf16").to("cuda")
    one = ex16(imgs[5:6].cuda())["feature_maps"].clone()
    check_bar("vitl14reg_518_layer18/bf16/vs_oracle_a", rel_err(one.cpu(), ref), 3e-2)   # the metric's configuration
    batch = ex16(imgs.cuda())["feature_maps"]
    assert torch.equal(batch[5], one[0])


def test_gemm_bf16_operands_beyond_4gib():
    """A > 4 GiB (the buffer-resource range): tiles deep inside the matrix still read the right rows."""
    from foundpose_amd import ops
    M, K, N = 1179648, 2048, 256  # A = 4.5 GiB of bf16
    a = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    rows = [0, 255, 1048576 + 3, M - 1]  # first tile, and rows past the 4-GiB mark
    for r in rows:
        a[r] = torch.randn(K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    out = ops.gemm_bf16(a, w, torch.zeros(N, device="cuda"), epilogue=5)
    for r in rows:
        ref = a[r].double() @ w.double().T
        assert float((out[r].double() - ref).abs().max()) < 1e-3 * float(ref.abs().max()), r
    assert float(out[300000].abs().max()) == 0.0


# ------------------------------------------------------------------ fp8 operands (BASELINE config 5: ViT-g/14 fp8)
def _fake_quant(x, scale):
    return (x.float() * scale).clamp(-448, 448).to(torch.float8_e4m3fn)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_quantize_fp8_matches_torch_cast(dtype):
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(3, 4099, generator=g) * 3).to(dtype)
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-9, 500.0, -1e6, 447.9, 0.0009765625, 17.0]).to(dtype)
    for scale in (1.0, 37.5):
        got = ops.quantize_fp8(x.cuda(), scale).cpu()
        ref = _fake_quant(x, scale)
        assert got.dtype == torch.float8_e4m3fn and torch.equal(got.view(torch.uint8) & 0x7f, ref.view(torch.uint8) & 0x7f)  # same magnitude bits
        assert torch.equal(got.float(), ref.float())


@pytest.mark.parametrize("M,N,K,epi", [(256, 256, 128, 0), (512, 768, 1536, 0), (256, 512, 1024, 1), (512, 256, 4096, 3), (256, 512, 256, 6),
                                       (256, 4608, 1536, 0), (256, 8192, 1536, 6), (256, 1536, 4096, 3)])  # last three: ViT-g/14 block shapes
def test_gemm_fp8_vs_fp64_on_quantised_operands(M, N, K, epi):
    """fp8 x fp8 products are exact in fp32; only the accumulation order differs from an fp64 reference on the same
    quantised operands -> fp32-accumulation tolerance, not an fp8 one.  Epilogues: bias, GELU, LayerScale-residual, SwiGLU."""
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(M + N + K + epi)
    x, w = torch.randn(M, K, generator=g) * 2, torch.randn(N, K, generator=g) * 0.05
    bias, gamma = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5
    sx = 448.0 / float(x.abs().max())
    sw = 448.0 / w.abs().amax(dim=1)                      # per output channel
    xq, wq = _fake_quant(x, sx), (w * sw[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    deq = 1.0 / (sx * sw)                                 # per column
    y = (xq.double() @ wq.double().T) * deq.double() + bias.double()
    if epi == 3:
        resid = torch.randn(M, N, generator=g)
        ref = resid.double() + gamma.double() * y
        col, b = deq * gamma, bias / deq
        out = ops.gemm_fp8(xq.cuda(), wq.cuda(), b.cuda(), col.cuda(), out=resid.clone().cuda(), epilogue=3, m_valid=M - 3).cpu()
        assert torch.equal(out[M - 3:], resid[M - 3:])    # rows past m_valid untouched
        out, ref = out[:M - 3], ref[:M - 3]
    else:
        out = ops.gemm_fp8(xq.cuda(), wq.cuda(), (bias / deq).cuda(), deq.cuda(), epilogue=epi).float().cpu()
        if epi == 1:
            ref = torch.nn.functional.gelu(y)
        elif epi == 6:
            ref = torch.nn.functional.silu(y[:, 0::2]) * y[:, 1::2]
        else:
            ref = y
    tol = 2e-5 if epi == 3 else 6e-3                      # fp32 output / bf16 output rounding
    assert rel_err(out, ref) < tol
    # and the quantisation error of the whole op against the unquantised product stays at the fp8 level
    if epi == 0:
        exact = x.double() @ w.double().T + bias.double()
        assert rel_err(out, exact) < 6e-2


@pytest.mark.parametrize("epi", [1, 6])
def test_gemm_fp8_quantised_output(epi):
    """out_scale > 0: the GELU / SwiGLU result leaves as e4m3 bytes == quantising the bf16-free fp32 result."""
    from foundpose_amd import ops
    M, N, K = 512, 512, 256
    g = torch.Generator().manual_seed(epi)
    xq, wq = _fake_quant(torch.randn(M, K, generator=g), 60.0), _fake_quant(torch.randn(N, K, generator=g) * 0.05, 2000.0)
    bias, deq = torch.randn(N, generator=g) * 0.1, torch.full((N,), 1.0 / (60.0 * 2000.0))
    y = (xq.double() @ wq.double().T) * deq.double() + bias.double()
    ref = torch.nn.functional.gelu(y) if epi == 1 else torch.nn.functional.silu(y[:, 0::2]) * y[:, 1::2]
    s_out = 448.0 / float(ref.abs().max())
    out = ops.gemm_fp8(xq.cuda(), wq.cuda(), (bias / deq).cuda(), deq.cuda(), epilogue=epi, out_scale=s_out, m_valid=M - 2).cpu()
    assert out.dtype == torch.float8_e4m3fn and out.shape == ref.shape
    assert bool((out[M - 2:].float() == 0).all())         # rows past m_valid untouched (zero-initialised)
    got, want = out[:M - 2].float() / s_out, ref[:M - 2]
    # one e4m3 rounding of the value (2^-4 relative, 2^-10 / s_out below the normal range) + the epilogue's own
    # approximation error (polynomial GELU / exp2-based sigmoid: ~1e-3 absolute, as in the bf16 epilogue tests)
    assert bool(((got - want).abs() <= want.abs() * 2 ** -4 + 2 ** -9 / s_out + 2e-3).all())
    assert float((got - want).abs().mean()) < 0.03 * float(want.abs().mean())


@pytest.mark.parametrize("epi,out_scale", [(0, 0.0), (1, 0.0), (6, 0.0), (1, 3.0), (6, 3.0)])
def test_gemm_fp8_320_row_tile_equals_256_row_tile(epi, out_scale):
    """fp8 GEMMs (bias / GELU / SwiGLU, bf16 and e4m3 outputs): the 320-row block tile gives the 256-row tile's bits."""
    from foundpose_amd import ops
    M, N, K, mv = 2560, 1024, 512, 2000
    g = torch.Generator().manual_seed(epi + 7)
    a = ops.quantize_fp8(torch.randn(M, K, generator=g).cuda(), 60.0)
    w = ops.quantize_fp8((torch.randn(N, K, generator=g) * 0.05).cuda(), 2000.0)
    bias, col = torch.randn(N, generator=g).cuda(), (torch.rand(N, generator=g) * 1e-5 + 1e-6).cuda()
    outs = [ops.gemm_fp8(a, w, bias, col, epilogue=epi | (tile << 8), m_valid=mv, out_scale=out_scale) for tile in (256, 320)]
    assert torch.equal(outs[0][:mv].view(torch.uint8), outs[1][:mv].view(torch.uint8))
    assert not bool(outs[1][mv:].view(torch.uint8).any())


def test_gemm_fp8_loud_failures():
    from foundpose_amd import ops
    from foundpose_amd._lib import FoundPoseNativeError
    a = torch.zeros(256, 128, dtype=torch.float8_e4m3fn, device="cuda")
    v = torch.zeros(256, device="cuda")
    with pytest.raises(FoundPoseNativeError, match="multiple of 128"):
        ops.gemm_fp8(torch.zeros(256, 64, dtype=torch.float8_e4m3fn, device="cuda"), torch.zeros(256, 64, dtype=torch.float8_e4m3fn, device="cuda"), v, v)
    with pytest.raises(FoundPoseNativeError, match="not available"):
        ops.gemm_fp8(a, a, v, v, epilogue=5)
    with pytest.raises(FoundPoseNativeError, match="GELU and SwiGLU"):
        ops.gemm_fp8(a, a, v, v, epilogue=0, out_scale=1.0)
    with pytest.raises(ValueError):
        ops.gemm_fp8(a.view(torch.uint8), a, v, v)


@pytest.mark.parametrize("ffn", ["mlp", "swiglu"])
def test_extractor_fp8_mode_vs_oracle_c(ffn):
    """precision="fp8": e4m3 block GEMMs with static activation scales from an explicit calibration call.  Checked against
    "oracle C" (oracle/vit.py fp8_act=: bf16 activations + e4m3 fake quantisation at the same points with the same
    scales) at bf16-level tolerance, and against the fp32 oracle at the fp8 noise level."""
    from foundpose_amd.vit_config import VitArch
    arch = VitArch(f"f8test-{ffn}-reg", dim=256, depth=3, heads=4, ffn=ffn, hidden=512 if ffn == "swiglu" else 1024, registers=4,
                   pretrain_grid=4, interp_antialias=True, interp_offset=0.0)
    name = f"dinov2_version={arch.name}_stride=14_facet=token_layer=2_logbin=0_norm=1"
    from foundpose_amd import feature_util
    mk = lambda: feature_util.make_feature_extractor(name, random_init_seed=77, precision="fp8", arch=arch).to("cuda")
    ex = mk()
    imgs = synthetic.make_crops(3, 56, seed=5)
    with pytest.raises(_lib.FoundPoseNativeError, match="static activation scales"):
        ex(imgs.cuda())  # no implicit calibration on whatever batch comes first
    ex.calibrate_fp8(imgs.cuda())
    out = ex(imgs.cuda())
    scales = ex.act_scales
    assert scales.shape == (3, 4) and bool((scales > 0).all())
    fm = out["feature_maps"].cpu()
    sd = synthetic.make_vit_state_dict(arch, 77)
    ref_c = ov.extractor_forward(sd, arch, imgs, 2, True, fp8_act=scales)["feature_maps"]
    ref_32 = ov.extractor_forward(sd, arch, imgs, 2, True)["feature_maps"]
    scale = float(ref_32.abs().max())
    check_bar(f"fp8_{ffn}/vs_oracle_c_max", float((fm - ref_c).abs().max()) / scale, 4e-2)       # same quantisation points: bf16-level agreement
    check_bar(f"fp8_{ffn}/vs_fp32_max", float((fm - ref_32).abs().max()) / scale, 0.25)           # fp8 noise against the exact model
    check_bar(f"fp8_{ffn}/vs_fp32_rms", float((fm - ref_32).pow(2).mean().sqrt()) / scale, 3e-2)
    # explicit scales reproduce the calibrated run bit for bit; a second batch reuses the static scales
    ex2 = mk()
    ex2.calibrate_fp8(act_scales=scales)
    assert torch.equal(ex2(imgs.cuda())["feature_maps"].cpu(), fm)
    ex3 = feature_util.make_feature_extractor(name, random_init_seed=77, precision="fp8", arch=arch, act_scales=scales).to("cuda")  # scales as part of the model
    assert torch.equal(ex3(imgs.cuda())["feature_maps"].cpu(), fm)
    assert torch.equal(ex.act_scales, scales) and torch.isfinite(ex(synthetic.make_crops(2, 56, seed=6).cuda())["feature_maps"]).all()


def test_fp8_mode_loud_failures():
    from foundpose_amd import feature_util
    with pytest.raises(NotImplementedError, match="multiples of 256"):
        feature_util.make_feature_extractor("dinov2_version=vits14-reg_stride=14_facet=token_layer=9_norm=1", precision="fp8")


@pytest.mark.parametrize("version,size,layer", [("vits14-reg", 224, 9), ("vitl14-reg", 518, 3)])
def test_layernorm_fold_vs_kernel_sequence(version, size, layer):
    """bf16 mode with the block LayerNorms folded into the GEMMs (default) vs the LayerNorm-kernel sequence: two bf16
    evaluations of the same network whose rounding points differ -- they agree at the bf16 level, and each is as close to the
    fp32 oracle as the other."""
    from foundpose_amd import feature_util
    name = f"dinov2_version={version}_stride=14_facet=token_layer={layer}_norm=1"
    arch = ARCHS[version]
    imgs = synthetic.make_crops(3, size, seed=4)
    sd = synthetic.make_vit_state_dict(arch, seed=21)
    fold = feature_util.make_feature_extractor(name, state_dict=sd, precision="bf16").to("cuda")
    plain = feature_util.make_feature_extractor(name, state_dict=sd, precision="bf16", fold_layernorm=False).to("cuda")
    assert fold.fold_layernorm and not plain.fold_layernorm
    a, b = fold(imgs.cuda())["feature_maps"].cpu(), plain(imgs.cuda())["feature_maps"].cpu()
    ref = ov.extractor_forward(sd, arch, imgs, layer, True)["feature_maps"]
    ea, eb, d = rel_err(a, ref), rel_err(b, ref), rel_err(a, b)
    print(f"\n{version}@{size} layer {layer}: folded vs fp32 oracle {ea:.4f}, kernel sequence vs fp32 oracle {eb:.4f}, folded vs kernel sequence {d:.4f}")
    check_bar(f"ln_fold_{version}_{size}/folded_vs_sequence", d, 2e-2)
    check_bar(f"ln_fold_{version}_{size}/folded_vs_oracle_a", ea, 3e-2)
    assert ea < 1.5 * eb + 2e-3


@pytest.mark.parametrize("M,N,K,m_valid", [(1280, 512, 128, 1280), (3840, 3072, 1024, 3140), (2560, 4096, 1024, 1374)])
def test_gemm_320_row_tile_equals_256_row_tile(M, N, K, m_valid):
    """The 320 x 256 block tile of the wide bf16 outputs (qkv, fc1) walks k in the same order per output element as the 256^2 tile: bias / GELU
    outputs, plain and in the folded-LayerNorm form, are bit-identical; rows past m_valid (incl. whole tiles of padding rows, which leave
    at once) stay untouched."""
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias, cs = torch.randn(N, generator=g).cuda(), torch.randn(N, generator=g).cuda()
    ln_row = torch.stack([1 + torch.rand(M, generator=g), torch.randn(M, generator=g)], 1).contiguous().cuda()
    for epi in (0, 1):
        outs = []
        for tile in (256, 320):
            out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
            ops.gemm_bf16(a, w, bias, out=out, epilogue=epi | (tile << 8), m_valid=m_valid)
            outs.append(out)
        assert torch.equal(outs[0], outs[1]) and bool(torch.all(outs[1][m_valid:] == 7.0)) and bool(torch.any(outs[1][:m_valid] != 7.0))
        f256 = ops.gemm_bf16_ln(a, w, bias, cs, ln_row, epilogue=epi, tile=256, m_valid=m_valid)
        f320 = ops.gemm_bf16_ln(a, w, bias, cs, ln_row, epilogue=epi, tile=320, m_valid=m_valid)
        assert torch.equal(f256[:m_valid], f320[:m_valid])


@pytest.mark.parametrize("M,N,K,m_valid", [(1536, 1024, 1024, 1374), (1536, 1024, 4096, 1374), (256, 384, 1536, 200), (2816, 1024, 1024, 2748),
                                           (128, 128, 64, 100), (256, 256, 128, 250), (256, 128, 192, 256), (128, 256, 256, 70)])   # 1 .. 4 K-tiles: shorter than the 64-row tile's four-stage pipeline
def test_gemm_64_row_tile_equals_128_row_tile(M, N, K, m_valid):
    """The 64 x 128 block tile the residual GEMMs (proj, fc2) of a batch of one or two crops launch -- the reference loop's shape, one detection
    at a time -- walks k in the same order per output element as the 128^2 tile: the fp32 LayerScale-residual stream (epilogue 3), the fp32
    stream with its bf16 copy and LayerNorm row sums (7) and the (hi, lo) bf16 stream (8) come out bit-identical; rows past m_valid untouched."""
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias, gamma = torch.randn(N, generator=g).cuda(), torch.randn(N, generator=g).cuda()
    x0 = (torch.randn(M, N, generator=g) * 3 + 0.7).cuda()
    outs = []
    for tile in (128, 64):
        x3 = x0.clone()
        ops.gemm_bf16(a, w, bias, gamma=gamma, out=x3, epilogue=3 | (tile << 8), m_valid=m_valid)
        x7 = x0.clone()
        xb7, st7 = ops.gemm_bf16_resid_ln(a, w, bias, x7, tile=tile, m_valid=m_valid)
        xb8, xl8 = x0.to(torch.bfloat16), (x0 - x0.to(torch.bfloat16).float()).to(torch.bfloat16)
        st8 = ops.gemm_bf16_resid_hilo(a, w, bias, xb8, xl8, tile=tile, m_valid=m_valid)
        outs.append((x3, x7, xb7[:m_valid].clone(), st7[:, :m_valid].clone(), xb8, xl8, st8[:, :m_valid].clone()))
    for t128, t64 in zip(*outs):
        assert torch.equal(t128, t64)
    assert torch.equal(outs[1][0][m_valid:], x0[m_valid:]) and not torch.equal(outs[1][0][:m_valid], x0[:m_valid])


@pytest.mark.parametrize("tile,M,D,N2", [(128, 256, 256, 512), (256, 512, 1024, 1024), (128, 384, 384, 1152)])
def test_folded_layernorm_gemm_pair(tile, M, D, N2):
    """The two halves of the LayerNorm fold at op level: the residual GEMM (epilogue 7) emits bf16(x) and the row's partial
    sums next to the fp32 stream; the next GEMM normalises in its epilogue.  Together they must equal
    Linear(LayerNorm(x_new)) computed the plain way (fp64 on the same bf16 operands), at one bf16 rounding of the result."""
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(M + D)
    h = torch.randn(M, 2 * D, generator=g).to(torch.bfloat16)
    w_out = (torch.randn(D, 2 * D, generator=g) * 0.05).to(torch.bfloat16)
    b_out = torch.randn(D, generator=g)
    x0 = torch.randn(M, D, generator=g) * 3 + 0.7
    x = x0.clone().cuda()
    xb, stats = ops.gemm_bf16_resid_ln(h.cuda(), w_out.cuda(), b_out.cuda(), x, tile=tile, m_valid=M - 3)
    x_ref = x0.double() + h.double() @ w_out.double().T + b_out.double()
    x_ref[M - 3:] = x0[M - 3:].double()                           # rows past M_valid stay untouched
    assert rel_err(x.cpu(), x_ref) < 3e-5
    assert torch.equal(xb[:M - 3].cpu(), x[:M - 3].cpu().to(torch.bfloat16))
    parts = x[:M - 3].cpu().double().reshape(M - 3, D // 128, 128)
    assert rel_err(stats[:, :M - 3, 0].cpu().T, parts.sum(-1)) < 1e-5 and rel_err(stats[:, :M - 3, 1].cpu().T, (parts ** 2).sum(-1)) < 1e-5
    ln_row = ops.ln_finalize(stats, D)
    mu, var = x_ref[:M - 3].mean(1), x_ref[:M - 3].var(1, unbiased=False)
    assert rel_err(ln_row[:M - 3, 0].cpu(), 1 / torch.sqrt(var + 1e-6)) < 1e-5
    # consumer: gain folded into W (rounded to bf16), shift into the bias
    gain, shift = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    w_in, b_in = torch.randn(N2, D, generator=g) * 0.05, torch.randn(N2, generator=g)
    wf = (w_in * gain[None, :]).to(torch.bfloat16)
    bf, cs = b_in + w_in @ shift, wf.float().sum(1)
    for epi in (0, 1):
        out = ops.gemm_bf16_ln(xb, wf.cuda(), bf.cuda(), cs.cuda(), ln_row, epilogue=epi, tile=tile, m_valid=M - 3)[:M - 3].cpu().double()
        xn = (xb[:M - 3].cpu().double() - mu[:, None]) / torch.sqrt(var + 1e-6)[:, None]     # the same bf16 operand, normalised exactly
        ref = xn @ wf.double().T + bf.double()
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        assert float((out - ref).abs().max()) < 1.5 * 2 ** -8 * float(ref.abs().max()), (epi, tile)


@pytest.mark.parametrize("tile,M,D", [(128, 256, 256), (256, 512, 1024), (128, 384, 384)])
def test_residual_gemm_on_the_hi_lo_stream(tile, M, D):
    """Epilogue 8: the residual stream as two bf16 arrays (hi, lo).  Against epilogue 7 run on the fp32 stream x = hi + lo: the new high
    halves ARE epilogue 7's bf16 copy (both round the same fp32 x') and the row sums agree to fp32 rounding; hi' + lo' reproduces x' to
    16 mantissa bits; rows past M_valid are untouched."""
    from foundpose_amd import ops
    g = torch.Generator().manual_seed(M + D)
    h = torch.randn(M, 2 * D, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(D, 2 * D, generator=g) * 0.05).to(torch.bfloat16).cuda()
    b = torch.randn(D, generator=g).cuda()
    x0 = torch.randn(M, D, generator=g) * 3 + 0.7
    x0[:, 5] *= 200.0                                              # a massive-activation channel
    hi = x0.to(torch.bfloat16)
    lo = (x0 - hi.float()).to(torch.bfloat16)
    x32 = (hi.float() + lo.float()).cuda()                        # the stream the pair represents (exact in fp32)
    xb7, st7 = ops.gemm_bf16_resid_ln(h, w, b, x32, tile=tile, m_valid=M - 3)
    xb8, xl8 = hi.clone().cuda(), lo.clone().cuda()
    st8 = ops.gemm_bf16_resid_hilo(h, w, b, xb8, xl8, tile=tile, m_valid=M - 3)
    assert torch.equal(xb8[:M - 3], xb7[:M - 3])
    torch.testing.assert_close(st8[:, :M - 3], st7[:, :M - 3], rtol=2e-6, atol=1e-4)    # same addends, another tree (16 lanes x 8 columns vs 32 x 4)
    got = xb8.float() + xl8.float()
    err = (got[:M - 3] - x32[:M - 3]).abs() / x32[:M - 3].abs().clamp_min(1e-3)
    assert float(err.max()) < 2 ** -15                            # hi: 8 bits, lo: 8 more (round to nearest twice)
    assert torch.equal(xb8[M - 3:].cpu(), hi[M - 3:]) and torch.equal(xl8[M - 3:].cpu(), lo[M - 3:])


@pytest.mark.parametrize("K", [1024, 4096])
def test_residual_gemm_hi_lo_320_row_tile_equals_256_row_tile(K):
    """Epilogue 8 on the 320-row block tile (the residual GEMMs of large batches): hi', lo' and the row sums equal the 256-row tile's
    bit for bit; rows past M_valid untouched."""
    from foundpose_amd import ops
    M, N, mv = 2560, 1024, 2200
    g = torch.Generator().manual_seed(K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    xb0 = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    xl0 = (torch.randn(M, N, generator=g) * 0.003).to(torch.bfloat16).cuda()
    outs = []
    for tile in (256, 320):
        xb, xl = xb0.clone(), xl0.clone()
        st = ops.gemm_bf16_resid_hilo(a, w, bias, xb, xl, tile=tile, m_valid=mv)
        outs.append((xb, xl, st))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2][:, :mv], outs[1][2][:, :mv])
    assert torch.equal(outs[1][0][mv:], xb0[mv:]) and torch.equal(outs[1][1][mv:], xl0[mv:]) and not torch.equal(outs[1][0][:mv], xb0[:mv])


def test_tall_gemm_tiles_leave_the_features_bit_identical():
    """The whole bf16 forward with the 320-row tiles allowed (default) and forbidden (tall_tiles=False -> fp_vit_model.flags & FP_VIT_NO_TALL_TILES): the
    same feature maps bit for bit, at a batch whose residual GEMMs take the taller tile too -- and likewise in the f16 mode."""
    from foundpose_amd import feature_util
    arch = ARCHS["vitl14-reg"]
    sd = synthetic.make_vit_state_dict(arch, seed=4)
    imgs = synthetic.make_crops(24, 518, seed=2).cuda()      # 24 x 1374 tokens = 32 976 rows: qkv, fc1 AND the residual GEMMs (516 tiles = 3 rounds -> 416 = 2) take the taller tile
    for prec in ("bf16", "f16"):
        fms = []
        for tall in (True, False):
            ex = feature_util.make_feature_extractor("dinov2_version=vitl14-reg_stride=14_facet=token_layer=2_norm=1", state_dict=sd, precision=prec, tall_tiles=tall).to("cuda")
            assert ex.padded_rows(24 * 1374) == (24 * 1374 + 1279) // 1280 * 1280
            fms.append(ex(imgs)["feature_maps"].clone())
            del ex
        assert torch.equal(fms[0], fms[1]), prec


@pytest.mark.parametrize("hilo", ["0", "1"])
def test_hi_lo_stream_end_to_end_switch(hilo):
    """resid_hilo=False keeps the fp32 residual stream in every block; True (default) holds it as (hi, lo) bf16 pairs in front of the hooked
    block.  Both stay within the bf16 mode's distance from oracle B, and the engine's token-selected form == the full form bit for bit
    in either setting (the hooked block always runs on an fp32 stream)."""
    from foundpose_amd import feature_util
    arch = ARCHS["vits14-reg"]
    name = "dinov2_version=vits14-reg_stride=14_facet=token_layer=5_norm=1"
    sd = synthetic.make_vit_state_dict(arch, seed=4)
    imgs = synthetic.make_crops(3, 224, seed=1)
    ex = feature_util.make_feature_extractor(name, state_dict=sd, precision="bf16", resid_hilo=hilo == "1").to("cuda")
    fm = ex(imgs.cuda())["feature_maps"].cpu()
    ref_b = ov.extractor_forward(sd, arch, imgs, 5, True, quant="bf16")["feature_maps"]
    check_bar(f"hilo_{hilo}_vits14reg_224_l5/bf16/vs_oracle_b", rel_err(fm, ref_b), 1.5e-2)
    one = ex(imgs[1:2].cuda())["feature_maps"].cpu()
    assert torch.equal(one[0], fm[1])                              # batch invariance


@pytest.mark.parametrize("version,size,precision", [("vits14-reg", 224, "fp32"), ("vits14-reg", 224, "bf16"), ("vitl14-reg", 518, "bf16")])
def test_fused_norm_and_sampling_is_bit_identical(version, size, precision):
    """fp_vit_sample_features (final LayerNorm + bilinear sampling at the query points only, straight from the residual
    stream) == forward()["feature_maps"] sampled with fp_sample_bilinear, bit for bit -- also at points off the cell centres
    and outside the image, where the four taps and the zero padding matter."""
    from foundpose_amd import feature_util, ops
    layer = 3
    ex = feature_util.make_feature_extractor(f"dinov2_version={version}_stride=14_facet=token_layer={layer}_norm=1", random_init_seed=8, precision=precision).to("cuda")
    B = 3
    imgs = synthetic.make_crops(B, size, seed=2).cuda()
    g = torch.Generator().manual_seed(0)
    grid = feature_util.generate_grid_points((size, size), 14.0)
    pts = torch.cat([grid[torch.randperm(grid.shape[0], generator=g)[:200]], torch.rand(150, 2, generator=g) * (size + 20) - 10]).cuda()
    img_of = torch.randint(0, B, (pts.shape[0],), generator=g).to(torch.int32).cuda()
    fmap = ex(imgs)["feature_maps"]
    want = ops.sample_bilinear(fmap, pts, img_of, (size, size))
    ex.forward_hidden(imgs)
    got = ex.sample_patch_features(pts, img_of)
    assert torch.equal(got, want)


@pytest.mark.parametrize("version,size,layer,precision", [("vits14-reg", 224, 3, "bf16"), ("vits14-reg", 224, 0, "bf16"), ("vitl14-reg", 518, 2, "bf16"),
                                                         ("vits14-reg", 224, 3, "f16x3"), ("vits14-reg", 224, 0, "f16x3"), ("vitl14-reg", 518, 1, "f16x3"),
                                                         ("vitl14-reg", 518, 1, "fp8"), ("vitl14-reg", 224, 0, "fp8"), ("vitg14-reg", 224, 1, "fp8")])
def test_selected_tokens_in_hooked_block_bit_identical(version, size, layer, precision):
    """fp_vit_forward_prefix + fp_vit_block_selected + fp_vit_sample_features_selected: the hooked block computed only for
    the patch tokens the sampling reads (attention queries, proj, fc1, fc2 on the selected rows; keys / values all tokens)
    gives the SAME sampled features, bit for bit, as the full forward -- per-image selections of different sizes (a disc, a
    thin bar, a single cell, everything), points on and off the cell centres."""
    from foundpose_amd import feature_util
    from foundpose_amd.engine import FoundPoseEngine
    ex = feature_util.make_feature_extractor(f"dinov2_version={version}_stride=14_facet=token_layer={layer}_norm=1", random_init_seed=8, precision=precision).to("cuda")
    assert ex.supports_token_selection
    B = 4
    imgs = synthetic.make_crops(B, size, seed=5).cuda()
    if precision == "fp8":
        ex.calibrate_fp8(imgs)   # the static activation scales are part of the fp8 model
    masks = torch.zeros(B, size, size, dtype=torch.uint8)
    masks[0] = synthetic.make_disc_mask(size)
    masks[1, size // 3: size // 3 + 20, 10: size - 30] = 1
    masks[2, size // 2, size // 2] = 1
    masks[2, 7, 7] = 1                       # a corner cell: taps outside the map
    masks[3] = 1
    masks = masks.cuda()
    eng = FoundPoseEngine(ex, None)
    # full forward, then sampling at the engine's query points
    ex.forward_hidden(imgs)
    q_pts, q_img, counts = eng.query_points(masks)
    want = ex.sample_patch_features(q_pts, q_img)
    # selected: prefix, block on the selected tokens, sampling through the row map
    pending = eng._query_points_begin(masks, select_tokens=True)
    ex.forward_hidden(imgs, prefix_only=True)
    q_pts2, q_img2, counts2, (sel_rows, sel_off, row_map, num_sel, max_sel) = eng._query_points_end(*pending)
    assert counts2 == counts and torch.equal(q_pts2, q_pts)
    gh = size // 14
    assert 0 < num_sel < B * gh * gh and max_sel == gh * gh   # image 3 selects every cell, image 2 a handful
    ex.forward_selected_block(sel_rows, sel_off, num_sel, max_sel)
    got = ex.sample_patch_features(q_pts2, q_img2, row_map=row_map)
    assert not bool(torch.isnan(got).any())
    assert torch.equal(got, want)
    # points off the cell centres inside the selected region sample the same values too (image 3: every cell is selected)
    g = torch.Generator().manual_seed(1)
    off = (torch.rand(300, 2, generator=g) * (size + 16) - 8).cuda()
    img3 = torch.full((300,), 3, dtype=torch.int32, device="cuda")
    got_off = ex.sample_patch_features(off, img3, row_map=row_map)
    ex.forward_hidden(imgs)
    assert torch.equal(got_off, ex.sample_patch_features(off, img3))


@pytest.mark.parametrize("dtype", [torch.uint8, torch.bool, torch.float32])
def test_query_select_equals_filter_points_by_mask(dtype):
    """fp_query_select (the engine's batched mask test + point lists) == generate_grid_points + filter_points_by_mask of the
    reference (feature_util.py:19-41) per detection, in its order; mask dtypes as callers hand them over; random masks,
    an empty one, a full one, pixels on the canvas border."""
    from foundpose_amd import feature_util
    from foundpose_amd.engine import FoundPoseEngine
    ex = feature_util.make_feature_extractor("dinov2_version=vits14-reg_stride=14_facet=token_layer=1_norm=1", random_init_seed=8, precision="bf16").to("cuda")
    g = torch.Generator().manual_seed(3)
    for size, cell in ((224, 14.0), (210, 10.0), (126, 7.0), ((280, 168), 14.0)):   # the last one: width != height
        B = 5
        size_wh = size if isinstance(size, tuple) else (size, size)
        m = (torch.rand(B, size_wh[1], size_wh[0], generator=g) > 0.6)
        m[1] = False
        m[2] = True
        m[3] = False
        m[3, 0, :] = True          # only border pixels: strictly-inside test
        m[3, :, 0] = True
        masks = m.to(dtype).cuda()
        eng = FoundPoseEngine(ex, None, grid_cell_size=cell)
        pts, img, counts = eng.query_points(masks)
        grid = feature_util.generate_grid_points(size_wh, cell)
        rnd = torch.rand(300, 2, generator=g) * torch.tensor([size_wh[0] + 20.0, size_wh[1] + 20.0]) - 10.0   # arbitrary points, some outside the canvas
        off = 0
        for b in range(B):
            ref = feature_util.filter_points_by_mask(grid, m[b].to(dtype)).cuda()     # CPU tensors: the reference's tensor-indexing form
            assert torch.equal(feature_util.filter_points_by_mask(grid.cuda(), masks[b]), ref)    # device tensors: the drop-in's fp_query_select form
            assert torch.equal(feature_util.filter_points_by_mask(rnd.cuda(), masks[b]), feature_util.filter_points_by_mask(rnd, m[b].to(dtype)).cuda())
            assert counts[b] == ref.shape[0], (size, b)
            assert torch.equal(pts[off:off + counts[b]], ref)
            assert bool((img[off:off + counts[b]] == b).all())
            off += counts[b]
        assert off == pts.shape[0]
