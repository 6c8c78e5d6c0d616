import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_uq_amd import _native as nat
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
idx = nat.NativeIndex(768, 0); idx.reserve(n)
for c in range(n // 125_000):
    g = torch.Generator(device=dev); g.manual_seed(1235 + c)
    x = torch.nn.functional.normalize(torch.randn((125_000, 768), device=dev, generator=g), dim=1).half().contiguous()
    idx.add_f16_device(x, 125_000); del x
q = torch.randn((64, 768), device=dev)
sc = torch.empty((64, 10), device=dev); rw = torch.empty((64, 10), device=dev, dtype=torch.int64); st = torch.zeros((64,), device=dev, dtype=torch.int32)
print("scan8", idx.get_option("scan8"), "level", idx.get_option("scan8_level"))
idx.search_device(q, 64, 10, 0, sc, rw, None, st, 0); torch.cuda.synchronize()
print("after first search: level", idx.get_option("scan8_level"), "calibrated rows", idx.get_option("scan8_calibrated_rows"), "used", idx.get_option("scan8_used"), "row err", idx.get_option("scan8_row_err"))
for c in (0, 1):
    print("class", c, [(round(idx.get_option(f"scan8_calib_ms_{c}{l}"), 4), int(idx.get_option(f"scan8_calib_unc_{c}{l}"))) for l in range(3)])
idx.close()
