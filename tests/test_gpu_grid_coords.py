"""The lattice the decoder kernels generate in registers (grid_point, csrc/sdf_mlp_common.h) against the reference's own
grid construction (utils/mesh.py:27-40,82-96) BIT FOR BIT, through the debug hook asdf_debug_grid_coords (the kernels call
the same device function).  ref_grid.npz holds the reference's columns; for N = 128 / 256 and for zoom-cube lattices the
checker is the oracle's grid_coords, which tests/test_oracle_decoder.py pins bit-exactly to that fixture."""
import ctypes

import numpy as np
import pytest
import torch

from alignsdf_amd import _native

pytestmark = pytest.mark.gpu


def device_coords(N, origin, voxel_size, mode, first=0, count=None):
    L = _native.lib()
    count = N ** 3 - first if count is None else count
    out = torch.empty((count, 3), dtype=torch.float32, device="cuda")
    org = (ctypes.c_float * 3)(*[float(np.float32(o)) for o in origin])
    _native.check(L.asdf_debug_grid_coords(N, org, ctypes.c_float(float(np.float32(voxel_size))), mode, first, count, out.data_ptr(),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "asdf_debug_grid_coords")
    return out.cpu().numpy()


@pytest.mark.parametrize("N", [4, 5, 7, 16, 31, 32, 64, 100])
def test_reference_lattice_bit_for_bit(N, golden_dir):
    g = np.load(golden_dir + "/ref_grid.npz")
    got = device_coords(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1), _native.GRID_REFERENCE)
    want = g["coord_%d" % N]
    sel = g["sel_%d" % N]
    assert np.array_equal(got[sel].view(np.uint32), want.view(np.uint32)), N


@pytest.mark.parametrize("N", [128, 256])
def test_full_size_lattices_bit_for_bit(N):
    """Every one of the N^3 points of pass 1, and of a pass-2 lattice with a zoom cube's fp32 voxel size and origin."""
    from oracle import sdf_oracle as orc
    got = device_coords(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1), _native.GRID_REFERENCE)
    want = orc.grid_coords(N, 2.0 / (N - 1), [-1, -1, -1]).numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    nvs = torch.tensor(0.0047244094, dtype=torch.float32) * 1.0          # 0-dim fp32, like new_voxel_size (utils/mesh.py:252)
    norg = torch.tensor([-0.62204725, -0.35433072, -0.37007874], dtype=torch.float32)
    got = device_coords(N, norg.tolist(), nvs.item(), _native.GRID_REFERENCE)
    want = orc.grid_coords(N, nvs, norg).numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("N", [5, 32, 100])
def test_integer_mode_and_windows(N):
    from oracle import sdf_oracle as orc
    want = orc.grid_coords(N, 0.013, [-0.5, 0.25, -1.0], integer_mode=True).numpy()
    got = device_coords(N, [-0.5, 0.25, -1.0], 0.013, _native.GRID_INTEGER)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    first, count = N ** 3 // 3, min(1000, N ** 3 - N ** 3 // 3)
    win = device_coords(N, [-0.5, 0.25, -1.0], 0.013, _native.GRID_INTEGER, first, count)
    assert np.array_equal(win, got[first:first + count])


@pytest.mark.parametrize("N", [17, 255, 257, 511, 512, 777, 1023, 1024])
def test_windows_of_large_and_odd_lattices_bit_for_bit(N):
    """grid_point's 32-bit index arithmetic and its one-FMA fmod (csrc/sdf_mlp_common.h: fmod_small) against the reference's operations
    on windows of lattices up to the largest the C ABI accepts: the start, the end, around 2^24 (where int64 -> fp32 starts to round)
    and pseudo-random places, in both lattice modes."""
    from oracle import sdf_oracle as orc
    P = N ** 3
    rng = np.random.RandomState(N)
    firsts = [0, max(0, P - 70000), max(0, min(P - 70000, (1 << 24) - 35000))] + [int(f) for f in rng.randint(0, max(1, P - 70000), 5)]
    for first in firsts:
        count = min(70000, P - first)
        for mode, integer in ((_native.GRID_REFERENCE, False), (_native.GRID_INTEGER, True)):
            got = device_coords(N, [-1.0, -0.37, 0.21], 2.0 / (N - 1), mode, first, count)
            want = orc.grid_coords_window(N, 2.0 / (N - 1), [-1.0, -0.37, 0.21], first, count, integer_mode=integer).numpy()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (N, first, mode)


def test_argument_checks(native_lib):
    org = (ctypes.c_float * 3)(0, 0, 0)
    buf = torch.empty(30, device="cuda")
    assert native_lib.asdf_debug_grid_coords(1, org, ctypes.c_float(1.0), 0, 0, 1, buf.data_ptr(), None) == -1
    assert native_lib.asdf_debug_grid_coords(4, org, ctypes.c_float(1.0), 5, 0, 1, buf.data_ptr(), None) == -1
    assert native_lib.asdf_debug_grid_coords(4, org, ctypes.c_float(1.0), 0, 60, 10, buf.data_ptr(), None) == -1
    assert native_lib.asdf_debug_grid_coords(4, org, ctypes.c_float(1.0), 0, 0, 0, buf.data_ptr(), None) == 0
