#!/usr/bin/env python3
"""Errors of the stored-y1 and rebuilt-y1 stencil forms against float64 (tests/dw_reference.py) on the tests' data: the numbers quoted in
DESIGN.md section 5 and in the bounds of tests/test_gpu_dwfwd.py / test_gpu_dwbwd.py.  python3 tools/rebuilt_parity_report.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tests.dw_reference import rel_l2  # noqa: E402
from tests import test_gpu_dwbwd as B, test_gpu_dwfwd as F  # noqa: E402

for case, cin in (((130, 18, 32, 448, 1), 64), ((40, 36, 64, 448, 2), 64), ((130, 9, 16, 896, 1), 128), ((129, 18, 32, 896, 2), 128)):
    (y0, _), (y1, _), ref = F._stored_and_rebuilt(*case, cin=cin)
    print(f"fwd {case} cin={cin}: rel L2 vs float64  stored {rel_l2(y0, ref):.3e}  rebuilt {rel_l2(y1, ref):.3e}   max abs / max|ref|  "
          f"stored {float((y0.double() - ref).abs().max() / ref.abs().max()):.3e}  rebuilt {float((y1.double() - ref).abs().max() / ref.abs().max()):.3e}", flush=True)
    (d0, w0, _), (d1, w1, _), r, _ = B._stored_and_rebuilt(*case, cin=cin)
    print(f"bwd {case} cin={cin}: dh1 rel L2  stored {rel_l2(d0, r[0]):.3e}  rebuilt {rel_l2(d1, r[0]):.3e}   dW rel L2  stored {rel_l2(w0, r[1]):.3e}  rebuilt {rel_l2(w1, r[1]):.3e}", flush=True)
