#!/usr/bin/env python3
"""Itemise the HBM traffic of bench.py's headline step (VonMises3D mixed, sparse trial history, two alternating
Newton iterates) from the plastic masks of the workload itself, at several access granularities of the
memory side, next to the algorithmic count -- VERDICT r1 item 6: where do the 5-6 % between
the algorithmic 487 B/pt and the measured bytes go?

    python tools/traffic_itemise.py [n]

Per 64-point tile the kernel (fcamd_kernels.hip: tile_von_mises) touches the eps_n rows of `need = plastic now |
plastic at the previous launch`; with <= 20 such rows it reads / writes exactly those rows (48 B each, three
16-B chunks), otherwise the whole 3072-B tile image; a tile with need != 0 stores its 64 alpha values (512 B);
the mask word is read always and written when it changed."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
wl = bench.Workload("von_mises_mixed", n, 1234, dev, 0)
wl.warmup(2)
masks = []
for i in (0, 1):
    wl.launch(i)
    torch.cuda.synchronize()
    masks.append(wl.hmask.clone())
ntiles = n // 64
shifts = torch.arange(64, device=dev, dtype=torch.int64)


def bits(words):  # [tiles] int64 -> [tiles, 64] 0/1 float
    return ((words[:ntiles, None] >> shifts[None, :]) & 1).to(torch.float32)


b0, b1 = bits(masks[0]), bits(masks[1])
need = torch.clamp(b0 + b1, max=1.0)  # alternating iterates: plastic now | plastic at the previous launch
rows = need.sum(dim=1)
masked = rows <= 20  # kMaskedRowMaxVonMises
touched = rows > 0
plastic = 0.5 * (b0.sum() + b1.sum()).item() / n
changed = (masks[0][:ntiles] != masks[1][:ntiles]).float().mean().item()
out = {"n": n, "plastic_fraction": round(plastic, 4), "rows_touched_fraction": round(need.mean().item(), 4),
       "tiles_masked_path": round(masked.float().mean().item(), 4), "tiles_touched": round(touched.float().mean().item(), 4),
       "mask_words_rewritten": round(changed, 4)}
fixed_read = 72 + 48 + 8 + 8 / 64  # grad, committed stress, alpha, mask word
fixed_write = 48 + 288  # trial stress, tangent
alpha_w = 512 * touched.float().sum().item() / n
maskw_w = 8 * changed * ntiles / n
for gran in (16, 32, 64, 128):
    S = 3072 // gran
    r = torch.arange(64, device=dev)
    s0, s1 = (48 * r) // gran, (48 * r + 47) // gran
    M = torch.zeros(64, S, device=dev)
    for k in range(int((s1 - s0).max().item()) + 1):
        idx = torch.clamp(s0 + k, max=S - 1)
        M[r[(s0 + k) <= s1], idx[(s0 + k) <= s1]] = 1.0
    sect = ((need @ M) > 0).float().sum(dim=1) * gran  # bytes of distinct sectors holding touched rows, per tile
    eps = torch.where(masked, sect, torch.full_like(sect, 3072.0))
    eps = torch.where(touched, eps, torch.zeros_like(eps)).sum().item() / n
    out[f"granularity_{gran}B"] = {"eps_n_read_B_per_pt": round(eps, 2), "eps_n_write_B_per_pt": round(eps, 2),
                                   "read_B_per_pt": round(fixed_read + eps, 2),
                                   "write_B_per_pt": round(fixed_write + eps + alpha_w + maskw_w, 2),
                                   "total_B_per_pt": round(fixed_read + fixed_write + 2 * eps + alpha_w + maskw_w, 2)}
out["alpha_store_B_per_pt"] = round(alpha_w, 2)
out["mask_word_write_B_per_pt"] = round(maskw_w, 3)
out["algorithmic_B_per_pt"] = round(464 * (1 - plastic) + 568 * plastic, 2)
out["exact_rows_B_per_pt"] = {"eps_n_read": round(48 * need.mean().item(), 2), "alpha_write_if_masked_per_lane": round(8 * need.mean().item(), 2)}
# what a packed [alpha, eps_n(6)] row per point (the comfe-rs layout) would move: alpha enters the yield function, so the
# 56-byte row of EVERY point is read; rows of touched points are written
out["packed_rows_7"] = {"read_B_per_pt": round(72 + 48 + 56 + 8 / 64, 2), "write_B_per_pt_exact_rows": round(fixed_write + 56 * need.mean().item() + maskw_w, 2)}
print(json.dumps(out, indent=1))
