"""Bank-builder tier timings on the MI355X: PCA fit, k-means, tf-idf descriptors at LM-O-object scale
(798 templates x ~375 patches = 3e5 features, 1024 -> 256 dims, 2048 words, 50 k-means iterations)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundpose_amd import bank_builder, cluster_util, projector_util


def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0):9.1f} ms", flush=True)
    return out


T, P, D = 798, 375, 1024
g = torch.Generator(device="cuda").manual_seed(0)
raw = torch.randn(T * P, D, generator=g, device="cuda") * (torch.arange(1, D + 1, device="cuda") ** -0.5)
f2t = torch.repeat_interleave(torch.arange(T, dtype=torch.int32, device="cuda"), P)
proj = projector_util.PCAProjector(n_components=256)
timed("warm-up PCA fit (100k samples)", lambda: proj.fit(raw, max_samples=100000))
timed("PCA fit (100k x 1024 -> 256)", lambda: proj.fit(raw, max_samples=100000))
feats = timed("PCA transform (3e5 x 1024)", lambda: proj.transform(raw))
cent = timed("k-means 2048 words, 50 iters, 3e5 x 256", lambda: cluster_util.kmeans(feats, 2048, 50, verbose=False))[0]
timed("tf-idf descriptors (1-NN + 3-NN + histograms)", lambda: bank_builder.calc_tfidf_descriptors(feats, f2t, cent, T))
