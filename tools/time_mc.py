import sys, time, torch
sys.path.insert(0, '.')
from alignsdf_amd.marching_cubes import marching_cubes_device
for n in (64, 128, 256):
    ax = torch.linspace(-1, 1, n, device="cuda")
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    vol = torch.sqrt(zz * zz + yy * yy + xx * xx) - 0.63
    for it in range(3):
        torch.cuda.synchronize(); t = time.time()
        v, f = marching_cubes_device(vol, 0.0)
        torch.cuda.synchronize(); dt = time.time() - t
        print("MC N=%d %.3f ms V=%d F=%d  (%.1f GB/s on 4N^3 bytes)" % (n, dt * 1e3, v.shape[0], f.shape[0], 4 * n ** 3 / dt / 1e9), flush=True)
