#!/usr/bin/env python3
"""Per-block timeline of ONE prefill attention launch (library built with -DATT_TRACE=1: scripts/build_variants.py a_trace):
every block's start / end (s_memrealtime, 10 ns ticks), XCC and CU - which CU ran which blocks, when, and how long a tile took.
    gpurun -- 'MISTRAL_HIP_LIB=mistral-inference_amd/lib/variants/libmistral_hip_a_trace.so python scripts/attn_prefill_trace.py 4096'"""
import ctypes
import os
import sys
from collections import defaultdict

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "mistral-inference_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mistral_inference import _hip  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H, KV = 32, 8
dev = torch.device("cuda:0")
torch.manual_seed(7)
qkv = (torch.randn(T, (H + 2 * KV) * 128, device=dev)).to(torch.bfloat16)
q_start = torch.tensor([0, T], dtype=torch.int32, device=dev)
kv_before = torch.tensor([0], dtype=torch.int32, device=dev)
for _ in range(3):
    _hip.attn_prefill(qkv, H, KV, 128, None, None, T, q_start, kv_before, 1, T)
torch.cuda.synchronize()
nblk = ((T + 255) // 256) * H
buf = np.zeros(16384 * 4, dtype=np.uint64)
lib = _hip.lib()
lib.mi_probe_attn_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.mi_probe_attn_trace(buf.ctypes.data, buf.nbytes) == 0
tr = buf.reshape(-1, 4)[:nblk]
t0 = int(tr[:, 0].min())
start = (tr[:, 0].astype(np.int64) - t0) / 100.0  # us
end = (tr[:, 1].astype(np.int64) - t0) / 100.0
hw = tr[:, 2] & np.uint64(0xffffffff)
xcc = (tr[:, 2] >> np.uint64(32)) & np.uint64(0xf)
cu = ((hw >> np.uint64(8)) & np.uint64(0xf)) | (((hw >> np.uint64(12)) & np.uint64(0x1)) << np.uint64(4)) | (((hw >> np.uint64(13)) & np.uint64(0x7)) << np.uint64(5))
tiles = tr[:, 3].astype(np.int64)
print(f"T {T}: {nblk} blocks, launch span {end.max():.1f} us; sum of tiles {tiles.sum()} -> {tiles.sum() / 256:.1f} per CU")
per_cu = defaultdict(list)
for b in range(nblk):
    per_cu[(int(xcc[b]), int(cu[b]))].append(b)
print(f"distinct (xcc, cu): {len(per_cu)}; blocks per CU: min {min(len(v) for v in per_cu.values())} max {max(len(v) for v in per_cu.values())}")
dur = end - start
us_per_tile = dur / tiles
for lo, hi in [(1, 8), (9, 16), (17, 32), (33, 48), (49, 64), (65, 1 << 30)]:
    m = (tiles >= lo) & (tiles <= hi)
    if m.any():
        print(f"  blocks with {lo:3d}..{min(hi, int(tiles.max())):3d} tiles: n {int(m.sum()):4d}  us/tile mean {us_per_tile[m].mean():6.3f} min {us_per_tile[m].min():6.3f} max {us_per_tile[m].max():6.3f}"
              f"  start mean {start[m].mean():7.1f}  end mean {end[m].mean():7.1f} max {end[m].max():7.1f}")
cu_end = sorted((max(end[b] for b in v), sum(int(tiles[b]) for b in v), k) for k, v in per_cu.items())
print("CU finishing times (us): first 5", [f"{e:.1f}/{t}t" for e, t, _ in cu_end[:5]], " last 5", [f"{e:.1f}/{t}t" for e, t, _ in cu_end[-5:]])
tsum = np.array([t for _, t, _ in cu_end])
print(f"tiles per CU: min {tsum.min()} max {tsum.max()} mean {tsum.mean():.1f}")
# the busiest CU's blocks
e, t, k = cu_end[-1]
print("last CU", k, [(b, int(tiles[b]), round(float(start[b]), 1), round(float(end[b]), 1)) for b in per_cu[k]])
e, t, k = cu_end[0]
print("first CU", k, [(b, int(tiles[b]), round(float(start[b]), 1), round(float(end[b]), 1)) for b in per_cu[k]])
# per XCD
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  xcc {x}: blocks {int(m.sum())} tiles {int(tiles[m].sum())} last end {end[m].max():.1f} CUs {len({k for k in per_cu if k[0] == x})}")
# ---- ATT_TRACE=2 builds: cycles (s_memtime) waves 0 and 7 of a block spent, summed over its tiles, per phase
ph = buf[4096 * 4: 4096 * 4 + 1024 * 2 * 6].reshape(-1, 2, 6).astype(np.int64)
if ph.any():
    names = ["issue next tile's loads", "S^T = K.Q^T (16 MFMA)", "mask, row max, rescale", "exp + P.V (16 MFMA)", "stage next tile to LDS", "barrier"]
    for bid in (0, 1, 255, nblk - 1):
        nt = int(tiles[bid])
        print(f"block {bid}: {nt} tiles, {dur[bid]:.1f} us = {dur[bid] / nt:.3f} us per tile")
        for w, wn in ((0, "wave 0"), (1, "wave 7")):
            tot = ph[bid, w].sum()
            print(f"  {wn}: cycles per tile {tot / nt:7.0f} (-> {tot / nt / (dur[bid] / nt):.0f} MHz)  " + "  ".join(f"{n}: {c / nt:.0f}" for n, c in zip(names, ph[bid, w])))
