#!/usr/bin/env python
"""Timeline of the role-split split-fp16 attention kernel: s_memtime stamps of one workgroup's steady-state iterations (waves 0 and 4 =
the two halves of SIMD 0).  Needs the measurement build:  bash tools/build_variant.sh spp_tl attn.hip -DSPP_TIMELINE=1024 [-DSPP_NO_...]
    cp foundpose_amd/lib/spp_tl.so foundpose_amd/lib/libfoundpose_amd.so; python tools/spp_timeline.py
Stamps per iteration -- wave 0: top | matrix phase done | after mid barrier | softmax done | after vmcnt(4) | (end barrier) ;
wave 4: top | softmax done | after mid barrier | matrix phase done | after vmcnt(4)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundpose_amd import _lib, ops  # noqa: E402

B, N, D, H = 32, 1374, 1024, 16
x = torch.randn(B * N, 3 * D, device="cuda")
packed = torch.cat([ops.split16_pack(x[:, i * D:(i + 1) * D].contiguous(), 16.0) for i in range(3)], dim=1)
for _ in range(3):
    ops.attention_split(packed, B, N, D, H, 16.0, 16.0, variant=2)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 128)()
lib = _lib.lib()
rc = lib.fp_debug_spp_timeline(buf)
assert rc == 0, rc
names = [["top", "matrix done", "mid barrier", "softmax done", "vmcnt(4)"], ["top", "softmax done", "mid barrier", "matrix done", "vmcnt(4)"]]
for g in range(2):
    st = [buf[g * 64 + i] for i in range(64)]
    st = [v for v in st if v]
    print(f"wave {4 * g}: {len(st)} stamps; clock ticks between stamps (s_memtime), iterations as rows")
    t0 = st[0]
    for it in range(len(st) // 5):
        row = st[it * 5:(it + 1) * 5]
        nxt = st[(it + 1) * 5] if (it + 1) * 5 < len(st) else None
        d = [row[i + 1] - row[i] for i in range(4)] + ([nxt - row[4]] if nxt else [])
        print(f"  it {it:2d} @ {row[0] - t0:7d}: " + "  ".join(f"{n} {v:5d}" for n, v in zip(["phase1", "->bar", "phase2", "wait", "end bar"], d)))
