"""Where mc_classify's time goes: the count phase alone (events), on volumes with no / few / many active cells."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from alignsdf_amd import _native
from alignsdf_amd.marching_cubes import _workspace
L = _native.lib()
n = 256
ax = torch.linspace(-1, 1, n, device="cuda")
zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
sphere = (torch.sqrt(zz * zz + yy * yy + xx * xx) - 0.63).contiguous()
vols = {"all positive (no active cell)": torch.ones_like(sphere) + 0.001 * sphere, "sphere (1 % active)": sphere,
        "noise (all cells active)": torch.rand_like(sphere) - 0.5}
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, vol in vols.items():
    ws, res = _workspace(vol.shape, vol.device, 0)
    best = 1e9
    for it in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _native.check(L.asdf_mc_count_enqueue(vol.data_ptr(), n, n, n, ctypes.c_double(0.0), ws.data_ptr(), ws.numel(), res.data_ptr(), st), "count")
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print("%-34s count phase %.1f us  V=%d F=%d" % (name, best * 1e3, res[0].item(), res[1].item()))
