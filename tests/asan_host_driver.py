"""Run under the AddressSanitizer runtime by tests/test_host_asan.py (LD_PRELOAD + DWN_LIB_PATH = the host-ASAN build of the
library): drives every host-only path of the C-ABI that works without a GPU — layout checks, workspace carving for all block /
stem / cortex / readout shapes of the benchmarked architecture and for ragged ones, the support / geometry predicates over a sweep
of plane sizes (the y1-rebuilding stencil's register geometries, the fused conv_pw backward), and the error path of every entry point
(no device here: each call must come back with an error code and a message, not a crash).  Prints ASAN_HOST_OK when done."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import sensorium_amd._lib as L

lib = L.lib
assert lib.dwn_abi_version() == 7
for cname, struct in L._STRUCTS.items():
    assert lib.dwn_sizeof(cname.encode()) == C.sizeof(struct), cname

features = (64, 64, 64, 64, 128, 128, 128, 256, 256)
strides = (2, 1, 1, 1, 2, 1, 1, 2, 1)
calls = 0
for dtype in (L.DWN_F32, L.DWN_BF16):
    for (B, T, H0, W0) in ((32, 32, 36, 64), (2, 8, 36, 64), (3, 5, 7, 9), (1, 1, 1, 1), (90, 16, 64, 64), (2, 4, 130, 260)):
        h, w = H0, W0
        s = L.StemArgs(); s.dtype = dtype; s.training = 1; s.B = B; s.Cin = 5; s.C0 = 64; s.S = T * h * w
        assert lib.dwn_stem_workspace_bytes(C.byref(s)) > 0
        for i, (cin, st) in enumerate(zip(features, strides)):
            cout = features[i + 1] if i + 1 < len(features) else cin
            for exp in (7, 6, 3):
                for training in (0, 1):
                    a = L.BlockArgs()
                    a.dtype = dtype; a.training = training; a.B = B; a.T = T; a.Hin = h; a.Win = w
                    a.Hout = (h - 1) // st + 1; a.Wout = (w - 1) // st + 1
                    a.Cin = cin; a.Cmid = cin * exp; a.Cout = cout; a.stride = st; a.ks = 3; a.kt = 5
                    a.se_r = max(1, cin * exp // 32)
                    for bwd in (0, 1):
                        n = lib.dwn_block_workspace_bytes(C.byref(a), bwd)
                        assert n > 0
                    assert lib.dwn_block_forward_writes(C.byref(a)) in (0, 1, 2, 3)
                    for mode in (0, 1, 2):       # conv_pwl backward path: by shape / per-sample products / materialised du
                        a.pwl_bwd = mode
                        assert lib.dwn_block_workspace_bytes(C.byref(a), 1) > 0
                    for y1m in (0, 1):           # y1 left unmaterialised where both stencils rebuild it / always stored
                        a.y1_mode = y1m
                        wr = lib.dwn_block_forward_writes(C.byref(a))
                        assert wr in (0, 1, 2, 3) and (wr & 1 or y1m == 0 or not training)
                        assert lib.dwn_block_workspace_bytes(C.byref(a), 0) > 0 and lib.dwn_block_workspace_bytes(C.byref(a), 1) > 0
                    lib.dwn_pw_bwd_fused_supported(dtype, B * T * h * w, a.Cmid, a.Cin)
                    calls += 8
            h, w = (h - 1) // st + 1, (w - 1) // st + 1
        for groups in (1, 2, 4):
            c = L.CortexArgs(); c.dtype = dtype; c.training = 1; c.B = B; c.T = T; c.Cin = 256; c.C = 1024; c.groups = groups
            r = L.ReadoutArgs(); r.dtype = dtype; r.B = B; r.T = T; r.Cin = 4096; r.groups = groups
            for n_out in (1, 7, 7863, 8285):
                r.n_out = n_out
                for bwd in (0, 1):
                    assert lib.dwn_cortex_workspace_bytes(C.byref(c), bwd) > 0
                    assert lib.dwn_readout_workspace_bytes(C.byref(r), bwd) > 0
                assert lib.dwn_readout_wt_bytes(C.byref(r)) > 0
                calls += 5
# the rebuilt-y1 stencil: LDS / register-geometry predicate over a sweep of plane sizes, both strides, both channel counts
for cin in (64, 128, 256, 72):
    assert lib.dwn_dw_spatial_rc_blob_bytes(cin * 7, cin) >= 0
    for st in (1, 2):
        for hin in (1, 2, 5, 9, 18, 36, 64, 130):
            for win in list(range(1, 40)) + [63, 64, 65, 126, 130, 158, 167, 168, 255, 256, 257, 512]:
                for dtype in (L.DWN_F32, L.DWN_BF16):
                    assert lib.dwn_dw_spatial_rc_supported(dtype, cin, cin * 7, 3, st, hin, win) in (0, 1)
                    calls += 1
# the rebuilt-y1 backward stencils: support predicate over plane sizes / channel counts
for cin in (64, 128, 72):
    for cmid in (64, 448, 72, 896):
        for st in (1, 2):
            for hin in (1, 9, 18, 36):
                for win in (8, 16, 32, 64, 30, 130):
                    for dtype in (L.DWN_F32, L.DWN_BF16):
                        d = L.DwSpatialBwdArgs()
                        d.planes = 3; d.Hin = hin; d.Win = win; d.Hout = (hin - 1) // st + 1; d.Wout = (win - 1) // st + 1
                        d.C = cmid; d.stride = st; d.ks = 3; d.dy.ld = cmid; d.y1.ld = cmid; d.a0_ld = cin; d.Cin = cin
                        assert lib.dwn_dw_spatial_bwd_rc_supported(C.byref(d), dtype) in (0, 1)
                        f = L.DwSpatialFwdArgs()
                        f.planes = 3; f.Hin = hin; f.Win = win; f.Hout = (hin - 1) // st + 1; f.Wout = (win - 1) // st + 1
                        f.C = cmid; f.stride = st; f.ks = 3; f.inp.ld = cmid; f.a0_ld = cin; f.Cin = cin
                        assert lib.dwn_dw_spatial_fwd_rc_supported(C.byref(f), dtype) in (0, 1)
                        calls += 2
assert lib.dwn_conv_pw_bn_stats_workspace_bytes(64) > 0
# every entry point's error path: no device, null / zero arguments — an error code and a message, never a crash
def expect_error(rc):
    assert rc != 0
    assert lib.dwn_last_error()
a = L.BlockArgs(); a.dtype = L.DWN_BF16; a.B = 2; a.T = 4; a.Hin = 8; a.Win = 16; a.Hout = 8; a.Wout = 16
a.Cin = 64; a.Cmid = 448; a.Cout = 64; a.stride = 1; a.ks = 3; a.kt = 5; a.se_r = 14; a.training = 1
expect_error(lib.dwn_block_forward(C.byref(a), 0, None))
expect_error(lib.dwn_block_backward(C.byref(a), 0, None))
s = L.StemArgs(); s.dtype = L.DWN_BF16; s.training = 1; s.B = 2; s.Cin = 5; s.C0 = 64; s.S = 128
expect_error(lib.dwn_stem_forward(C.byref(s), 0, None))
expect_error(lib.dwn_stem_backward(C.byref(s), 0, None))
c = L.CortexArgs(); c.dtype = L.DWN_BF16; c.training = 1; c.B = 2; c.T = 4; c.Cin = 64; c.C = 128; c.groups = 2
expect_error(lib.dwn_cortex_forward(C.byref(c), 0, None))
expect_error(lib.dwn_cortex_backward(C.byref(c), 0, None))
r = L.ReadoutArgs(); r.dtype = L.DWN_BF16; r.B = 2; r.T = 4; r.Cin = 128; r.groups = 2; r.n_out = 7
expect_error(lib.dwn_readout_forward(C.byref(r), 0, None))
expect_error(lib.dwn_readout_backward(C.byref(r), 0, None))
p = L.PoolArgs(); p.dtype = L.DWN_BF16; p.BT = 8; p.HW = 16; p.C = 64
expect_error(lib.dwn_pool_forward(C.byref(p), 0, None))
expect_error(lib.dwn_pool_backward(C.byref(p), 0, None))
g = L.GemmNNArgs(); g.M = 128; g.N = 64; g.K = 64; g.groups = 1
expect_error(lib.dwn_gemm_nn(C.byref(g), L.DWN_BF16, 0, None))
t = L.GemmTNArgs(); t.M = 128; t.R = 64; t.Cc = 64; t.groups = 1
expect_error(lib.dwn_gemm_tn(C.byref(t), L.DWN_BF16, 0, None))
# conv_pw backward without y1: workspace sizes over the architecture's widths and ragged ones, then the error path
for dtype in (L.DWN_F32, L.DWN_BF16):
    for (e_, cin_) in ((448, 64), (384, 64), (896, 128), (1792, 256), (24, 8), (200, 72)):
        assert lib.dwn_pw_backward_workspace_bytes(e_, cin_, dtype) > 0
        calls += 1
pw = L.PwBwdArgs(); pw.M = 1024; pw.E = 448; pw.Cin = 64
expect_error(lib.dwn_pw_backward(C.byref(pw), L.DWN_BF16, 0, None))
expect_error(lib.dwn_adamw_ema_multi(None, 1, 16, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, 0.999, 1.0, 0, None))
expect_error(lib.dwn_ema_lerp_multi(None, 1, 16, 0.999, 0, None))
expect_error(lib.dwn_poisson_loss_forward(None, None, None, 1, 1, 1e-8, None, 0, None))
print(f"ASAN_HOST_OK host calls={calls}", flush=True)
