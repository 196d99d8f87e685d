"""Drop-in counterpart of the reference's utils/mesh.py hot path (create_mesh_combined_decoder,
get_higher_res_cube, convert_sdf_samples_to_ply) on the HIP kernels.

Signatures follow utils/mesh.py:17, :198 and :331.  Volumes stay on the device; only the extracted mesh
(verts / faces) is copied to the host for export.  There is no CPU path: without the HIP library or an
MI355X the functions raise.
"""
import ctypes
import logging
import os

import numpy as np
import torch

from .. import _native
from ..marching_cubes import marching_cubes_device
from ..mesh_post import keep_largest_component_device
from ..ply import write_ply
from .utils import bind_sample, decoder_for

GRID_MODES = {"reference": _native.GRID_REFERENCE, "integer": _native.GRID_INTEGER}


def neg_bbox(volume):
    """int32[7] host array (min0,min1,min2,max0,max1,max2,count) of the voxels with value < 0 of a device volume."""
    vol = volume.detach().to(torch.float32).contiguous()
    if not vol.is_cuda:
        raise TypeError("neg_bbox needs a CUDA tensor (there is no CPU fallback)")
    out = torch.empty(16, dtype=torch.int32, device=vol.device)
    with torch.cuda.device(vol.device):
        _native.check(_native.lib().asdf_neg_bbox(vol.data_ptr(), vol.shape[0], vol.shape[1], vol.shape[2], out.data_ptr(),
                                                   ctypes.c_void_p(torch.cuda.current_stream(vol.device).cuda_stream)),
                      "asdf_neg_bbox")
    return out[:7].cpu().numpy()


def zoom_cube_from_bboxes(bboxes, N, voxel_size):
    """The fp32 arithmetic of get_higher_res_cube (utils/mesh.py:239-254) on per-branch bounding boxes
    [(min3, max3, count)]; an empty branch contributes zeros (utils/mesh.py:209-211,225-227).
    Returns (new_voxel_size 0-dim fp32 tensor, new_origin [3] fp32 tensor) on the CPU, like the reference."""
    los, his = [], []
    for lo, hi, count in bboxes:
        if count == 0:
            los.append(torch.zeros(3)); his.append(torch.zeros(3))
        else:
            los.append(torch.tensor([float(v) for v in lo])); his.append(torch.tensor([float(v) for v in hi]))
    min_index = los[0] if len(los) == 1 else torch.min(los[0], los[1])
    max_index = his[0] if len(his) == 1 else torch.max(his[0], his[1])
    new_cube_size = (torch.max(max_index - min_index) + 4) * voxel_size
    new_voxel_size = new_cube_size / (N - 1)
    new_origin = (min_index - 2) * voxel_size - 1.0
    return new_voxel_size, new_origin


def get_higher_res_cube(hand_branch, obj_branch, sdf_values_hand, sdf_values_obj, N, voxel_origin, voxel_size):
    """Zoom cube around the negative voxels of the enabled branches (utils/mesh.py:198-256)."""
    boxes = []
    for on, vol in ((hand_branch, sdf_values_hand), (obj_branch, sdf_values_obj)):
        if on:
            b = neg_bbox(vol)
            boxes.append((b[0:3], b[3:6], int(b[6])))
    return zoom_cube_from_bboxes(boxes, N, voxel_size)


def place_vertices(verts_d, faces_d, voxel_grid_origin, voxel_size, offset=None, scale=None):
    """MC output (device or already-copied host tensors) -> host arrays with the vertex arithmetic of
    utils/mesh.py:354-369 (spacing, origin, optional scale / offset).  Returns (verts, faces, mesh_points)."""
    verts, faces = verts_d.cpu().numpy(), faces_d.cpu().numpy()
    vs = voxel_size.item() if isinstance(voxel_size, torch.Tensor) else voxel_size
    spacing = [np.float32(vs)] * 3 if isinstance(voxel_size, torch.Tensor) else [vs] * 3
    if not np.array_equal(spacing, (1, 1, 1)):
        verts = verts * np.r_[spacing]
    mesh_points = np.zeros_like(verts)
    for a in range(3):
        mesh_points[:, a] = voxel_grid_origin[a] + verts[:, a]
    if scale is not None:
        mesh_points = mesh_points * scale
    if offset is not None:
        mesh_points = mesh_points + offset
    return verts, faces, mesh_points


def extract_surface(sdf, voxel_grid_origin, voxel_size, offset=None, scale=None):
    """MC on the device + place_vertices.  Raises like skimage on failure."""
    vol = sdf if isinstance(sdf, torch.Tensor) else torch.as_tensor(np.asarray(sdf))
    if not vol.is_cuda:
        vol = vol.cuda()
    verts_d, faces_d = marching_cubes_device(vol, 0.0)
    return place_vertices(verts_d, faces_d, voxel_grid_origin, voxel_size, offset, scale)


def ground_truth_mesh_path(ply_filename_out, task, data_root="data"):
    """Where the reference looks for the ground-truth mesh of an output file (utils/mesh.py:386-388):
    <data_root>/<task>/test/mesh_<hand|obj>/<sample id>.obj."""
    mesh_dir = "mesh_" + ply_filename_out.split("_")[-1].split(".")[0]
    gt_mesh_name = ply_filename_out.split("/")[-1].split("_")[0] + ".obj"
    return os.path.join(data_root, task, "test", mesh_dir, gt_mesh_name)


def begin_mesh(mesh_points, faces, ply_filename_out, eval_mode=False, task="obman", data_root="data", allow_missing_gt=False):
    """First half of utils/mesh.py:383-397 for the placed vertices of the (already filtered) surface: in eval mode sample
    both meshes and ENQUEUE the translate+scale ICP (K7) against the ground-truth mesh without waiting for it.
    Returns a ticket for end_mesh.  A missing ground-truth file aborts like the reference (its trimesh.load raises at
    utils/mesh.py:389) - a wrong data_root must not produce a full run of silently unaligned meshes; with
    allow_missing_gt the mesh is written unaligned and the caller records `icp_skipped`."""
    out_v, out_f = mesh_points, faces
    job = None
    if eval_mode:
        gt_path = ground_truth_mesh_path(ply_filename_out, task, data_root)
        if os.path.exists(gt_path):
            from ..icp import load_obj, start_alignment
            gt_v, gt_f = load_obj(gt_path)
            job = start_alignment(out_v, out_f, gt_v, gt_f)                         # 30 000 samples, <= 100 iterations
        elif allow_missing_gt:
            logging.warning("eval_mode: ground-truth mesh %s not found; writing the unaligned mesh (allow_missing_gt)" % gt_path)
        else:
            raise FileNotFoundError("eval_mode: ground-truth mesh %s not found (data_root=%r); pass allow_missing_gt to write "
                                    "unaligned meshes instead" % (gt_path, data_root))
    return out_v, out_f, job, ply_filename_out


def end_mesh(ticket):
    """Second half: wait for the ICP (if any), apply it, export.  Returns (trans [3], scale [1]) like the reference
    (zeros / one outside eval mode)."""
    out_v, out_f, job, ply_filename_out = ticket
    trans, scale = np.array([0, 0, 0]), np.array([1])
    if job is not None:
        from ..icp import finish_icp
        r = finish_icp(job, out_v)
        out_v = r["vertices"]
        trans, scale = np.asarray(r["all_trans"]).reshape(1, 3), np.asarray(r["all_scale"]).reshape(1)
    if ply_filename_out:
        os.makedirs(os.path.dirname(os.path.abspath(ply_filename_out)), exist_ok=True)
        write_ply(ply_filename_out, out_v, out_f)
    return trans, scale


def finish_mesh(mesh_points, faces, ply_filename_out, eval_mode=False, task="obman", data_root="data", allow_missing_gt=False):
    """utils/mesh.py:383-397 for the placed vertices of the (already filtered) surface: in eval mode align it to the
    ground-truth mesh with the translate+scale ICP (K7); export.  Returns (trans [3], scale [1])."""
    return end_mesh(begin_mesh(mesh_points, faces, ply_filename_out, eval_mode, task, data_root, allow_missing_gt))


def filter_surface_device(verts_d, faces_d, voxel_grid_origin, voxel_size):
    """K8 on a device surface: (kept lattice verts, kept faces) as device tensors.  Synchronises (it reads the counts)."""
    kv, kf, counts = keep_largest_component_device(verts_d, faces_d, voxel_size, voxel_grid_origin)
    c = counts.cpu().numpy()
    return kv[:c[0]], kf[:c[1]]


def begin_export_surface(verts_d, faces_d, voxel_grid_origin, voxel_size, ply_filename_out, offset=None, scale=None,
                         eval_mode=False, task="obman", largest_component=True, data_root="data", kept=None, allow_missing_gt=False):
    """place_vertices + the largest-component filter + begin_mesh: everything of the host tail up to (and including)
    enqueuing the eval-mode ICP.  `kept` = (verts, faces) of the largest component in lattice units when the caller has
    already run the device filter (the sample pipeline does, right behind marching cubes); otherwise the surface is
    filtered here on the device (K8; host arrays are uploaded for it - there is no host implementation in the product)."""
    if kept is None and largest_component:
        vd = verts_d if isinstance(verts_d, torch.Tensor) else torch.as_tensor(np.asarray(verts_d))
        fd = faces_d if isinstance(faces_d, torch.Tensor) else torch.as_tensor(np.asarray(faces_d))
        kept = filter_surface_device(vd.cuda().float(), fd.cuda().int(), voxel_grid_origin, voxel_size)
    verts, faces, mesh_points = place_vertices(verts_d, faces_d, voxel_grid_origin, voxel_size, offset, scale)
    if kept is not None:
        _, kept_faces, kept_points = place_vertices(kept[0], kept[1], voxel_grid_origin, voxel_size, offset, scale)
        return verts, faces, begin_mesh(kept_points, kept_faces, ply_filename_out, eval_mode, task, data_root, allow_missing_gt)
    return verts, faces, begin_mesh(mesh_points, faces, ply_filename_out, eval_mode, task, data_root, allow_missing_gt)


def end_export_surface(pending):
    """Wait for the ICP of begin_export_surface (if any), write the file.  Returns (verts, faces, trans, scale)."""
    verts, faces, ticket = pending
    trans, sc = end_mesh(ticket)
    return verts, faces, trans, sc


def export_surface(verts_d, faces_d, voxel_grid_origin, voxel_size, ply_filename_out, offset=None, scale=None, eval_mode=False,
                   task="obman", largest_component=True, data_root="data", kept=None, allow_missing_gt=False):
    """The host tail of convert_sdf_samples_to_ply for an already extracted surface (utils/mesh.py:360-397).
    Returns (verts, faces, trans, scale)."""
    return end_export_surface(begin_export_surface(verts_d, faces_d, voxel_grid_origin, voxel_size, ply_filename_out, offset,
                                                   scale, eval_mode, task, largest_component, data_root, kept, allow_missing_gt))


def convert_sdf_samples_to_ply(pytorch_3d_sdf_tensor, voxel_grid_origin, voxel_size, ply_filename_out, offset=None,
                               scale=None, eval_mode=False, task="obman", largest_component=True, data_root="data",
                               allow_missing_gt=False):
    """Iso-surface of one SDF volume -> .ply (utils/mesh.py:331-399).  Returns (verts, faces, trans, scale) with
    verts / faces the raw marching-cubes output like the reference.  MC failures are logged and skipped exactly
    like the reference (utils/mesh.py:353-358).  The written file holds the largest watertight component when the
    surface splits into several (utils/mesh.py:371-381, K8 / alignsdf_amd.mesh_post); in eval mode it is first aligned to
    the ground-truth mesh by the translate+scale ICP (utils/mesh.py:385-395, alignsdf_amd.icp) and `trans`, `scale`
    are the ICP's; otherwise they are zeros / one."""
    vol = pytorch_3d_sdf_tensor if isinstance(pytorch_3d_sdf_tensor, torch.Tensor) else torch.as_tensor(np.asarray(pytorch_3d_sdf_tensor))
    if not vol.is_cuda:
        vol = vol.cuda()
    try:
        verts_d, faces_d = marching_cubes_device(vol, 0.0)
    except (ValueError, RuntimeError) as e:
        logging.warning("Cannot reconstruct mesh from '{}'".format(ply_filename_out))
        print(e)
        return None, None, np.array([0, 0, 0]), np.array([1])
    return export_surface(verts_d, faces_d, voxel_grid_origin, voxel_size, ply_filename_out, offset, scale, eval_mode, task,
                          largest_component, data_root, allow_missing_gt=allow_missing_gt)


# colour per part label of the `--viz` output (the table of utils/mesh.py:305-310)
PART_COLORS = np.array([[13, 212, 128], [250, 70, 42], [131, 66, 37], [78, 137, 54], [187, 246, 163], [67, 220, 74]], dtype=np.uint8)


def _host(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def _scaled(points, offset, scale):
    pts = _host(points)
    if scale is not None:
        pts = pts * scale
    if offset is not None:
        pts = pts + offset
    return pts


def write_verts_label_to_obj(xyz, labels, obj_filename_out, offset=None, scale=None):
    """`v x y z g g g` lines with grey level 45 * label (utils/mesh.py:259-277)."""
    pts, lab = _scaled(xyz, offset, scale), _host(labels)
    with open(obj_filename_out, "w") as fp:
        fp.write("".join("v %.4f %.4f %.4f %.2f %.2f %.2f\n" % (v[0], v[1], v[2], c, c, c)
                         for v, c in zip(pts.tolist(), (lab * 45.0).tolist())))


def write_verts_label_to_npz(xyz, labels, npz_filename_out, offset=None, scale=None):
    """points / labels arrays (utils/mesh.py:280-296)."""
    np.savez(npz_filename_out, points=_scaled(xyz, offset, scale), labels=_host(labels))


def write_color_labeled_ply(xyz, faces, labels, ply_filename_out, offset=None, scale=None):
    """ASCII PLY coloured by part label (utils/mesh.py:299-326)."""
    from ..ply import write_ply_ascii
    pts, lab = _scaled(xyz, offset, scale), _host(labels)
    write_ply_ascii(ply_filename_out, pts, faces, PART_COLORS[lab.astype(np.int32)])


def label_points(decoder, latent_vec, mano_results, obj_results, specs, points):
    """Part label of every point [V,3] (normalised coordinates, any device): the label pass of utils/mesh.py:137-157
    in one launch (the reference chunks by max_batch).  Returns a float32 CPU tensor like `out_labels`."""
    hip = decoder_for(decoder, specs, mano_results)
    bind_sample(hip, specs, latent_vec, mano_results, obj_results)
    return hip.classify_points(points, want_sdf=False)[3].float().cpu()


def write_label_outputs(vertices, faces, labels, ply_filename_hand, offset, scale, viz):
    """The label files of one hand mesh (utils/mesh.py:160-184)."""
    if viz:
        write_verts_label_to_obj(vertices, labels, ply_filename_hand + "_label.obj", offset, scale)
        write_color_labeled_ply(vertices, faces, labels, ply_filename_hand + "_color.ply", offset, scale)
    write_verts_label_to_npz(vertices, labels, ply_filename_hand + "_label.npz", offset, scale)


def decode_two_pass(hand_branch, obj_branch, decoder, latent_vec, mano_results, obj_results, specs, N, grid_mode="reference",
                    cam_intr=None, mc_only=False):
    """Pass 1 on [-1,1]^3, zoom cube, pass 2 (utils/mesh.py:21-121) entirely on the device.
    Returns dict(vol_hand, vol_obj device tensors of pass 2, voxel_size 0-dim fp32 tensor, origin list,
    bbox int32[16] of pass 1).  mc_only=True declares that the volumes are handed to marching cubes and nothing else: a
    decoder set to the narrow-band fine sweep (ASDF_FINE=band) may then deliver them exact next to the surface only.""" 
    hip = decoder_for(decoder, specs, mano_results)
    bind_sample(hip, specs, latent_vec, mano_results, obj_results, cam_intr)
    mode = GRID_MODES[grid_mode]
    voxel_size = 2.0 / (N - 1)
    # a branch that is switched off is neither meshed nor used for the zoom cube (utils/mesh.py:239-247), so its
    # head is not evaluated at all (the reference computes and discards it)
    # coarse pass: consumed only through its boxes (ordinary sweep, or the box-only sweep when the decoder is set to it);
    # a sweep whose range / error guards fired is repeated inside coarse_finish
    b = hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], voxel_size, mode, hand=hand_branch, obj=obj_branch))
    boxes = []
    if hand_branch:
        boxes.append((b[0:3], b[3:6], int(b[6])))
    if obj_branch:
        boxes.append((b[8:11], b[11:14], int(b[14])))
    new_voxel_size, new_origin = zoom_cube_from_bboxes(boxes, N, voxel_size)
    # fine pass: an ordinary sweep (range report through its bbox record), or - when the decoder is set to it and the caller
    # declares that the volumes go to marching cubes only - the narrow-band sweep; either is repeated if its guards fired
    for _ in range(6):
        vol_hand, vol_obj, ticket = hip.fine_begin(N, new_origin.tolist(), new_voxel_size.item(), mode, hand=hand_branch, obj=obj_branch,
                                                   mc_only=mc_only)
        if not hip.fine_needs_repeat(ticket):
            break
    return {"vol_hand": vol_hand, "vol_obj": vol_obj, "voxel_size": new_voxel_size, "origin": new_origin.tolist(), "bbox": b}


def create_mesh_combined_decoder(hand_branch, obj_branch, cls_branch, decoder, latent_vec, mano_results, obj_results, cam_intr,
                                 specs, filename, N=256, max_batch=32 ** 3, offset=None, scale=None, device="cpu",
                                 label_out=False, viz=False, eval_mode=False, task="obman", grid_mode="reference", return_stats=False):
    """Hand + object meshes of one sample (utils/mesh.py:17-195): writes <filename>_hand.ply / _obj.ply.
    `max_batch` and `device` are accepted for signature compatibility; chunking is internal to the kernel and
    the decoder's device is used.  `grid_mode="reference"` reproduces the true-division lattice of
    utils/mesh.py:33-34 bit for bit; "integer" is the axis-aligned lattice.  Returns None like the reference; with
    `return_stats=True` a dict of per-surface (V, F) counts (and the labels of the label pass).

    `cls_branch` only makes the reference store a per-voxel class column that nothing reads (utils/mesh.py:59-60,
    111-112); it is accepted and has no effect.  `label_out` runs the label pass over the hand mesh vertices
    (utils/mesh.py:137-184) and needs a decoder with a classifier head.  As in the reference, the object mesh is
    written with the hand mesh's ICP translation / scale as its offset / scale (utils/mesh.py:123-133,186-195)."""
    decoder.eval() if hasattr(decoder, "eval") else None
    # (the volumes go to marching cubes only: a decoder set to the narrow-band fine sweep may use it)
    r = decode_two_pass(hand_branch, obj_branch, decoder, latent_vec, mano_results, obj_results, specs, N, grid_mode, cam_intr,
                        mc_only=True)
    stats = {}
    if hand_branch:
        v, f, offset, scale = convert_sdf_samples_to_ply(r["vol_hand"], r["origin"], r["voxel_size"], filename + "_hand.ply", None,
                                                         None, eval_mode, task)
        stats["hand"] = (0, 0) if v is None else (len(v), len(f))
        if label_out and v is not None:
            vertices = np.array(v, copy=True)
            for a in range(3):
                vertices[:, a] = r["origin"][a] + vertices[:, a]
            vertices = torch.from_numpy(vertices)
            labels = label_points(decoder, latent_vec, mano_results, obj_results, specs, vertices)
            write_label_outputs(vertices, f, labels, filename + "_hand", offset, scale, viz)
            stats["labels"] = labels
    if obj_branch:
        v, f, _, _ = convert_sdf_samples_to_ply(r["vol_obj"], r["origin"], r["voxel_size"], filename + "_obj.ply", offset, scale,
                                                False)
        stats["obj"] = (0, 0) if v is None else (len(v), len(f))
    return stats if return_stats else None
