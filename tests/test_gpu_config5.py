"""BASELINE config 5 at one GPU's share: ViT-g/14-reg in the fp8 mode (real dims: D = 1536, 24 heads, SwiGLU hidden 4096),
a 50 000-template bank, 128 detections (= batch 1024 over 8 GPUs).

  * the fp8 extractor at batch 32 against "oracle C" (oracle/vit.py fp8_act=: the CPU model of the device's quantisation
    points) on the real ViT-g geometry -- the first blocks only, so the CPU side stays at seconds;
  * the whole path on the planted workload: every detection must retrieve its five planted templates in order, in the
    fp8 mode as in the library's fp32 mode, through the 32-detection chunks of the retrieval kernel and the strict
    (torch.topk) tie order on 50 000-element rows."""
import pytest
import torch

from foundpose_amd import engine as fe
from foundpose_amd import feature_util, synthetic, workload
from foundpose_amd.bank import DeviceBank
from foundpose_amd.vit_config import ARCHS
from oracle import vit as ov
from tests.helpers import check_bar

pytestmark = pytest.mark.gpu


_SD = {}


@pytest.fixture(params=["vitg14-reg", "vitg14"])   # BASELINE config 5 says "ViT-g/14": with the register tokens and, as literally named, without
def version(request):
    return request.param


@pytest.fixture
def vitg_sd(version):
    if version not in _SD:
        _SD.clear()            # one 1.1 B-parameter state dict in host memory at a time
        _SD[version] = synthetic.make_vit_state_dict(ARCHS[version], seed=3)
    return _SD[version]


def test_vitg14_fp8_batch32_vs_oracle_c(vitg_sd, version):
    arch, layer = ARCHS[version], 4
    name = f"dinov2_version={version}_stride=14_facet=token_layer={layer}_norm=1"
    ex = feature_util.make_feature_extractor(name, state_dict=vitg_sd, precision="fp8").to("cuda")
    imgs = synthetic.make_crops(32, 518, seed=0)
    scales = ex.calibrate_fp8(imgs.cuda())
    fm = ex(imgs.cuda())["feature_maps"]
    assert fm.shape == (32, 1536, 37, 37) and bool(torch.isfinite(fm).all()) and ex.arch.registers == (4 if version.endswith("-reg") else 0)
    b = 7
    ref_c = ov.extractor_forward(vitg_sd, arch, imgs[b:b + 1], layer, True, fp8_act=scales)["feature_maps"][0]
    ref_32 = ov.extractor_forward(vitg_sd, arch, imgs[b:b + 1], layer, True)["feature_maps"][0]
    got = fm[b].cpu()
    scale = float(ref_32.abs().max())
    e_c, e_32 = float((got - ref_c).abs().max()) / scale, float((got - ref_32).abs().max()) / scale
    rms_32 = float((got - ref_32).pow(2).mean().sqrt()) / scale
    print(f"\n{version} fp8, layer {layer}, crop {b} of 32: vs oracle C {e_c:.4f}, vs fp32 oracle max {e_32:.4f} rms {rms_32:.4f} (of the feature scale)")
    # same quantisation points: bf16-level agreement up to the elements that land on the other side of an e4m3 rounding boundary; then the
    # fp8 noise against the exact model -- both held to 2.5 x the measured values (tests/golden/measured_bars.json)
    check_bar(f"config5_{version}_l4_b32/fp8/vs_oracle_c_max", e_c, 8e-2)
    check_bar(f"config5_{version}_l4_b32/fp8/vs_fp32_max", e_32, 0.3)
    check_bar(f"config5_{version}_l4_b32/fp8/vs_fp32_rms", rms_32, 4e-2)
    # batch invariance with static scales: the crop alone == the crop inside the batch
    ex1 = feature_util.make_feature_extractor(name, state_dict=vitg_sd, precision="fp8", act_scales=scales).to("cuda")
    assert torch.equal(ex1(imgs[b:b + 1].cuda())["feature_maps"][0].cpu(), got)


def test_config5_share_fp8_engine_on_planted_bank(vitg_sd, version):
    name = f"dinov2_version={version}_stride=14_facet=token_layer=39_norm=1"
    B, T = 128, 50000
    ex32 = feature_util.make_feature_extractor(name, state_dict=vitg_sd, precision="fp32").to("cuda")
    wl = workload.build_planted_workload(ex32, B, 518, 1, T, seed=5, crop_seed=9)
    bank = DeviceBank(wl.repres)
    assert bank.max_templates == T and bank.feats.shape[0] > 15_000_000
    got32 = []
    eng32 = fe.FoundPoseEngine(ex32, bank, 14.0, 5, 300, tie_order="torch")
    for b0 in range(0, B, 32):
        r = eng32.infer_batch(wl.crops[b0:b0 + 32], wl.masks[b0:b0 + 32], wl.det_obj[b0:b0 + 32])
        got32 += [r.corresp_list(b) for b in range(32)]
    del eng32, ex32
    torch.cuda.empty_cache()
    ex8 = feature_util.make_feature_extractor(name, state_dict=vitg_sd, precision="fp8").to("cuda")
    scales = ex8.calibrate_fp8(wl.crops[:32])
    wl.repres[0].extractor_fp8_act_scales = scales.tolist()   # the scales travel with the bank
    res8 = fe.FoundPoseEngine(ex8, bank, 14.0, 5, 300, tie_order="torch").infer_batch(wl.crops, wl.masks, wl.det_obj)  # 128 detections, one call
    got8 = [res8.corresp_list(b) for b in range(B)]
    p32, p8 = workload.planted_stats(got32, wl.targets.tolist()), workload.planted_stats(got8, wl.targets.tolist())
    agree = workload.parity_stats(got8, got32)
    print(f"\n[config5 share, {version}] planted: fp32 {p32} fp8 {p8}\n[config5 share] fp8 vs fp32 mode: {agree}")
    assert p32["planted_top5_in_order"] == B and p8["planted_top5_in_order"] == B
    assert agree["templates_equal"] == B and agree["corresp_overlap"] >= 0.85
    # static scales from 32 calibration crops (with the default head room over the sample maxima) serving 128: clamped values must be rare.
    # The counter counts reporting THREADS; the LayerNorm launches alone (one wave per token row, two per block) are a lower bound of the
    # threads that could report, so the bound below is < 0.1 % of all producer threads a fortiori.
    n16, n8 = ex8.saturation_counts()
    ln_threads = 2 * 40 * B * (1 + ex8.arch.registers + 37 * 37) * 64
    print(f"[config5 share] fp8 clamped threads: {n8} of > {ln_threads} producer threads ({100.0 * n8 / ln_threads:.5f} %)")
    assert n16 == 0 and n8 < 1e-3 * ln_threads
