import ctypes as C, os, sys
sys.path.insert(0, '/root/repo')
import torch
import sensorium_amd._lib as L
from tests.gpu_helpers import dev
BF = torch.bfloat16
d = dev(); s = torch.cuda.current_stream().cuda_stream
planes, Hin, Win, Cc, stride, cin = 130, 9, 16, 448, 1, 64
g = torch.Generator(device=d); g.manual_seed(0)
a0 = torch.randn(planes * Hin * Win, cin, device=d, generator=g).to(BF)
w1 = (torch.randn(Cc, cin, device=d, generator=g) / cin ** 0.5).to(BF)
y1 = torch.empty(planes * Hin * Win, Cc, dtype=BF, device=d)
gm = L.GemmNNArgs(); da = L.LoadDesc(); da.p = a0.data_ptr(); da.ld = cin; da.rows_per_sample = 1
gm.a = da; gm.a_kind = L.LD_PLAIN; gm.b = w1.data_ptr(); gm.ldb = cin; gm.c = y1.data_ptr(); gm.ldc = Cc
gm.M, gm.N, gm.K, gm.groups = planes * Hin * Win, Cc, cin, 1; gm.epi = L.EPI_STORE
L.check(L.lib.dwn_gemm_nn(C.byref(gm), L.DWN_BF16, 0, s), "gemm")
coef = torch.cat([torch.ones(Cc, device=d), torch.zeros(Cc, device=d)])
w = torch.zeros(9, Cc, device=d); w[4] = 1.0          # centre tap only: y2 = z1 = SiLU(y1)
def run(mode):
    y2 = torch.full((planes * Hin * Win, Cc), float("nan"), dtype=BF, device=d)
    a = L.DwSpatialFwdArgs(); di = L.LoadDesc()
    di.p = y1.data_ptr() if mode == "stored" else None
    di.ld = Cc; di.rows_per_sample = 1; di.v1 = coef.data_ptr(); di.v2 = coef[Cc:].data_ptr(); di.act = 1
    a.inp = di; a.w = w.data_ptr(); a.out = y2.data_ptr(); a.planes = planes; a.Hin = Hin; a.Win = Win; a.Hout = Hin
    a.Wout = Win; a.C = Cc; a.stride = 1; a.ks = 3
    if mode == "rebuilt":
        a.a0 = a0.data_ptr(); a.a0_ld = cin; a.w1 = w1.data_ptr(); a.Cin = cin
    L.check(L.lib.dwn_dw_spatial_fwd(C.byref(a), L.DWN_BF16, 0, s), "fwd")
    torch.cuda.synchronize()
    return y2.view(planes, Hin, Win, Cc)
ref = run("stored")
z = torch.nn.functional.silu(y1.float()).to(BF).view(planes, Hin, Win, Cc)
print("stored path == silu(y1):", bool(torch.equal(ref, z)))
for it in range(int(os.environ.get('DBG_ITERS', '40'))):
    out = run("rebuilt")
    dm = out.view(torch.int16) != ref.view(torch.int16)
    n = int(dm.sum())
    print("iter", it, "mismatches", n)
    if n:
        idx = dm.nonzero()
        pl, r, c, ch = idx[0].tolist()
        print("  first", idx[0].tolist(), "stored", float(ref[pl, r, c, ch]), "rebuilt", float(out[pl, r, c, ch]))
        print("  rows", sorted(set(idx[:, 1].tolist())), "planes", sorted(set(idx[:, 0].tolist()))[:8], "chans", sorted(set(idx[:, 3].tolist()))[:8])
        # is the rebuilt value some other element of the reference?  same pixel other channel / same channel other pixel
        v = out[pl, r, c, ch]
        same_px = (ref[pl, r, c] == v).nonzero().flatten().tolist()
        print("  equals ref at same pixel, channels:", same_px[:8])
        # y1 value it would correspond to
        yy = y1.view(planes, Hin, Win, Cc)[pl, r, c]
        print("  y1[ch]", float(yy[ch]), "silu", float(torch.nn.functional.silu(yy[ch].float())), " neighbours ch-1, ch+1:", float(ref[pl, r, c, ch - 1]), float(ref[pl, r, c, ch + 1]))
        print("  rebuilt values along the row, this channel:", [round(float(x), 4) for x in out[pl, r, :, ch]])
        print("  stored  values along the row, this channel:", [round(float(x), 4) for x in ref[pl, r, :, ch]])
        break
import numpy as np
try:
    fn = L.lib.dwn_wf_dbg_read
    buf = np.zeros(16, dtype=np.uint32)
    fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
    print("dbg rc", fn(buf.ctypes.data), buf.tolist())
except AttributeError:
    pass
