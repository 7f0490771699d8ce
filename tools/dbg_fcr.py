import sys
sys.path.insert(0, '/root/repo')
import torch
from tests.test_gpu_dwfwd import _stored_vs_rebuilt
import os
print('LIB', os.environ.get('DWN_LIB_PATH', 'product'))
for case in 2 * [(130, 9, 16, 448, 1), (33, 18, 32, 448, 1), (131, 5, 8, 448, 1), (16, 9, 16, 448, 1), (130, 9, 16, 64, 1), (129, 18, 32, 448, 2), (40, 36, 64, 448, 2), (3, 36, 64, 64, 2)]:
    planes, Hin, Win, Cc, stride = case
    (y0, s0), (y1, s1) = _stored_vs_rebuilt(*case)
    Hout, Wout = (Hin - 1) // stride + 1, (Win - 1) // stride + 1
    d = (y0.view(torch.int16) != y1.view(torch.int16)).view(planes, Hout, Wout, Cc)
    n = int(d.sum())
    print(case, "mismatches", n, "of", d.numel())
    if n:
        idx = d.nonzero()
        print("  planes", sorted(set(idx[:, 0].tolist()))[:20], "rows", sorted(set(idx[:, 1].tolist())), "cols", sorted(set(idx[:, 2].tolist())),
              "chan slices", sorted(set((idx[:, 3] // 64).tolist())), "chan%64", sorted(set((idx[:, 3] % 64).tolist()))[:20])
        i = idx[0].tolist()
        print("  first", i, float(y0.view(planes, Hout, Wout, Cc)[tuple(i)]), float(y1.view(planes, Hout, Wout, Cc)[tuple(i)]))
