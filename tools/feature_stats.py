"""Statistics of the random-init ViT features that decide how a planted workload must be scaled: spread of the patch
features of one crop (common component vs distinct part), distance between crops, and the bf16 mode's error on them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundpose_amd import feature_util, synthetic, workload
ver, layer, size, B = (sys.argv[1:] + ["vitl14-reg", "18", "518", "8"])[:4]
layer, size, B = int(layer), int(size), int(B)
name = f"dinov2_version={ver}_stride=14_facet=token_layer={layer}_norm=1"
m = synthetic.make_disc_mask(size)
if "--noise" in sys.argv:
    crops, tex = synthetic.make_crops(B, size, seed=0).cuda(), None
else:  # the planted workload's crops: 682 textures, distinct inside the mask
    crops, tex = synthetic.make_dictionary_crops(B, size, m, 682, 14, seed=0)
    crops = crops.cuda()
masks = m.unsqueeze(0).repeat(B, 1, 1).cuda()
f = {}
for prec in ("fp32", "bf16"):
    ex = feature_util.make_feature_extractor(name, random_init_seed=1234, precision=prec).to("cuda")
    f[prec], pts, counts = workload.query_features(ex, crops, masks)
    del ex
x = f["fp32"]
Q = counts[0]
print(f"raw features: dim {x.shape[1]}, overall std {float(x.std()):.4f}, |x| mean {float(x.norm(dim=1).mean()):.3f}")
mu_all = x.mean(0, keepdim=True)
print(f"  |global mean vector| {float(mu_all.norm()):.3f};  after removing it: |x - mu| mean {float((x - mu_all).norm(dim=1).mean()):.3f}")
a = x[:Q]
mu = a.mean(0, keepdim=True)
print(f"  crop 0: |crop mean| {float(mu.norm()):.3f}, |x - crop mean| mean {float((a - mu).norm(dim=1).mean()):.3f}")
d = torch.cdist(a, a)
d.fill_diagonal_(float('inf'))
print(f"  crop 0: nearest other patch at {float(d.min(1).values.mean()):.3f} (mean), {float(d.min()):.3f} (min); mean pairwise {float(d[d.isfinite()].mean()):.3f}")
b = x[Q:2 * Q]
print(f"  crop 0 vs crop 1: same grid cell distance {float((a - b).norm(dim=1).mean()):.3f}, nearest in other crop {float(torch.cdist(a, b).min(1).values.mean()):.3f}")
e = (f["bf16"] - x).norm(dim=1)
print(f"  bf16 - fp32: |err| mean {float(e.mean()):.4f} max {float(e.max()):.4f}  (relative to nearest-other-patch distance: {float(e.mean() / d.min(1).values.mean()):.3f})")
if tex is not None:
    n = size // 14
    cell = (pts[:, 1] / 14).long() * n + (pts[:, 0] / 14).long()
    det = torch.repeat_interleave(torch.arange(B), torch.tensor(counts)).cuda()
    qt = tex.reshape(B, -1).cuda()[det, cell]
    ta, tb = qt[:Q], qt[Q:2 * Q]
    same = (ta[:, None] == tb[None, :])
    dd = torch.cdist(a, b)
    print(f"  dictionary crops: same texture in crop 0 and crop 1 ({int(same.sum())} pairs): distance {float(dd[same].mean()):.3f} mean / {float(dd[same].max()):.3f} max;"
          f" different textures: {float(dd[~same].min()):.3f} min / {float(dd[~same].mean()):.3f} mean")
