"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU restatement of the reference's per-detection hot path
(/root/reference/scripts/infer.py:468-542 and the utils it calls), used ONLY as the
checker by tests/, `__graft_entry__.smoke()` and the `cpu_baseline` leg of bench.py.
Nothing under foundpose_amd/ imports this package; the product path fails loudly
when its HIP library is missing rather than falling back to anything here.

Pinning status (DESIGN.md "Oracle"):
  * logic around the k-NN primitive, tf-idf, cyclic matching, topk tie behaviour,
    extractor wrapper, PCA projection, point utilities: PINNED against the
    reference's own Python modules imported in the build container
    (oracle/ref_shim.py + oracle/make_golden.py -> tests/golden/*.npz).
  * faiss 1.8.0 distance arithmetic (not in the mount) and the DINOv2 backbone
    (external/dinov2 is an empty submodule): PARITY UNPINNED -- restated from the
    published algorithms; the backbone is cross-checked against the independent
    `transformers` Dinov2WithRegisters implementation.
"""
