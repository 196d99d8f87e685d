"""Count-phase time (classify + finalize) of stand-alone builds of mc33.hip side by side - how the round-2 variants of
mc_classify were compared:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared [-D...] mc33.hip glue.cpp
-o tools/bin/mcx_<name>.so  (glue.cpp defines asdf::g_last_hip_error), then this script on the GPU box."""
import ctypes, glob, sys, torch
n = 256
ax = torch.linspace(-1, 1, n, device="cuda")
zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
sphere = (torch.sqrt(zz * zz + yy * yy + xx * xx) - 0.63).contiguous()
vols = {"no active": torch.ones_like(sphere) + 0.001 * sphere, "sphere": sphere}
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for path in sorted(glob.glob("tools/bin/mcx_*.so")):
    L = ctypes.CDLL(path)
    nb = ctypes.c_size_t()
    L.asdf_mc_workspace_bytes(n, n, n, ctypes.byref(nb))
    ws = torch.zeros(nb.value, dtype=torch.uint8, device="cuda")
    res = torch.zeros(4, dtype=torch.int32).pin_memory()
    L.asdf_mc_count_enqueue.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_void_p,
                                        ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    out = []
    for name, vol in vols.items():
        best = 1e9
        for it in range(15):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = L.asdf_mc_count_enqueue(vol.data_ptr(), n, n, n, 0.0, ws.data_ptr(), nb.value, None, st)
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out.append("%s %.1f us" % (name, best * 1e3))
    print(path, rc, " | ".join(out), flush=True)
