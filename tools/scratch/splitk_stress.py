import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch.nn.functional as F
from vits_amd import native, packing
from vits_amd.engine import choose_tile
DEV = "cuda:0"
SLICED = len(sys.argv) > 1 and sys.argv[1] == "sliced"      # the sliced-fp32 kernel instead of the fp32 one
g = torch.Generator().manual_seed(5)
bad = 0
for it, (cin, cout, k, dil, B, L) in enumerate([(256, 256, 3, 1, 1, 2064), (256, 256, 11, 5, 1, 2064), (192, 384, 5, 1, 1, 258),
                                                 (512, 256, 7, 1, 2, 33), (128, 128, 7, 3, 1, 1500), (64, 64, 11, 1, 1, 3000)] * 3):
    x = torch.randn(B, cin, L, generator=g).to(DEV)
    w = torch.randn(cout, cin, k, generator=g) / (cin * k) ** 0.5
    tile = choose_tile(cout); ti = native.tile_info(tile)
    wp = (packing.pack_conv_weight_split(w.to(DEV), ti.ck_split, ti.rows_per_block) if SLICED
          else packing.pack_conv_weight(w, ti.ck, ti.rows_per_block).to(DEV))
    pad = (k - 1) * dil // 2
    def run(split, out):
        native.SPLIT_SMALL_LAUNCHES = split
        native.conv1d(x, wp, out, c_in=cin, c_out=cout, taps=k, dilation=dil, pad=pad, tile=tile, epilogue=native.EPI_STORE)
    ref = torch.empty(B, cout, L, device=DEV); run(False, ref)
    first = torch.empty_like(ref); run(True, first)
    outs = [torch.empty_like(ref) for _ in range(200)]
    for o in outs: run(True, o)
    torch.cuda.synchronize()
    nd = sum(int(not torch.equal(o, first)) for o in outs)
    err = (first - ref).abs().max().item()
    bad += nd + (err > 1e-4)
    print(f"{cin}->{cout} k{k} L{L}: split vs unsplit {err:.2e}, {nd} of 200 repeats differ from the first")
print("BAD" if bad else "OK")
