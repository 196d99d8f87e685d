"""Legacy DeepSDF entry points (deep_sdf/mesh.py:14-116) on the HIP kernels: one pass on [-1,1]^3, the
hand head as the single output, marching cubes, binary PLY."""
import logging
import time

import numpy as np
import torch

from .. import _native
from ..marching_cubes import marching_cubes_device
from ..ply import write_ply
from .utils import legacy_evaluator


def create_mesh(decoder, latent_vec, filename, N=256, max_batch=32 ** 3, grid_mode="reference"):
    """deep_sdf.mesh.create_mesh (deep_sdf/mesh.py:14-61). `max_batch` is accepted and ignored.  `decoder` is any module with the
    legacy contract decoder(cat(latent, xyz)) -> [M, 1]; `latent_vec` None = decoder(xyz) (deep_sdf/utils.py:64-75)."""
    start = time.time()
    hip = legacy_evaluator(decoder, latent_vec)      # the fused kernels, or any single-output module on PyTorch-ROCm
    hip.set_sample(latent_vec)
    mode = {"reference": _native.GRID_REFERENCE, "integer": _native.GRID_INTEGER}[grid_mode]
    voxel_origin = [-1, -1, -1]
    voxel_size = 2.0 / (N - 1)
    # no bbox buffer on this path: decode_grid reads the decoder's fp16 range status behind the sweep and repeats it on the
    # fp32 kernel if the split-half planes overflowed
    sdf_values, _, _ = hip.decode_grid(N, voxel_origin, voxel_size, mode, want_bbox=False, obj=False)      # the single output only
    torch.cuda.synchronize(sdf_values.device)
    print("sampling takes: %f" % (time.time() - start))
    return convert_sdf_samples_to_ply(sdf_values, voxel_origin, voxel_size, filename + ".ply")


def convert_sdf_samples_to_ply(pytorch_3d_sdf_tensor, voxel_grid_origin, voxel_size, ply_filename_out):
    """deep_sdf.mesh.convert_sdf_samples_to_ply (deep_sdf/mesh.py:64-116); raises on MC failure like the original."""
    start_time = time.time()
    vol = pytorch_3d_sdf_tensor if pytorch_3d_sdf_tensor.is_cuda else pytorch_3d_sdf_tensor.cuda()
    verts_d, faces_d = marching_cubes_device(vol, 0.0)
    verts, faces = verts_d.cpu().numpy(), faces_d.cpu().numpy()
    verts = verts * np.r_[[voxel_size] * 3]
    mesh_points = np.zeros_like(verts)
    for a in range(3):
        mesh_points[:, a] = voxel_grid_origin[a] + verts[:, a]
    write_ply(ply_filename_out, mesh_points, faces)
    logging.debug("converting to ply format and writing to file took {} s".format(time.time() - start_time))
    return mesh_points, faces
