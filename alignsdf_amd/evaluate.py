"""Mesh-level evaluation: the Chamfer branch of the reference's evaluate.py (evaluate.py:19-66, summary :233-310) with
the nearest-neighbour searches on the GPU (alignsdf_amd.deep_sdf.metrics.chamfer).

Covers the default and `--obj` modes: every `<name>_hand.ply` (`_obj.ply`) under <experiment>/Eval_<task>/meshes is
compared with <data_dir>/mesh_hand/<name>.obj (mesh_obj), and `chamfer_hand.txt` (`chamfer_obj.txt`) is written in the
reference's layout - one `name, chamfer, joints error, verts error` line per mesh sorted by decreasing Chamfer
distance, then mean / median / failure count.  The joint / vertex / object-pose errors of the reference come from the
MANO and pose predictions of the image encoder, which is outside this build: they are written as `nan` ("not computed";
round 1 wrote 0.0, which a reader could not tell from a perfect score).  The `--mano`,
`--fit` and `--rot` modes and the best / worst example folders are not reproduced.
"""
import argparse
import logging
import os

import numpy as np

from .deep_sdf.metrics.chamfer import compute_trimesh_chamfer


def evaluate(experiment_directory, data_dir, task="obman", obj=False, optim=False, start_point=0, end_point=None, seed=0):
    """[(name, chamfer_dist, nan, nan)] for the predicted meshes [start_point, end_point) (evaluate.py:19-112).
    Meshes that cannot be evaluated are skipped like the reference's bare `except: continue`."""
    suffix = "_obj.ply" if obj else "_hand.ply"
    pred_mesh_path = os.path.join(experiment_directory, "Eval_" + task, "meshes")
    names = sorted(f.split("_")[0] for f in os.listdir(pred_mesh_path) if suffix in f)
    out = []
    for name in names[start_point:end_point]:
        pred = os.path.join(pred_mesh_path, name + suffix)
        gt = os.path.join(data_dir, "mesh_obj" if obj else "mesh_hand", name + ".obj")
        if not (os.path.exists(gt) and os.path.exists(pred)):
            continue
        try:
            out.append((name, compute_trimesh_chamfer(gt, pred, optim, False, seed=seed), float("nan"), float("nan")))
        except Exception as e:      # noqa: BLE001
            logging.warning("skipping %s: %s", name, e)
    return out, len(names)


def evaluate_queue(queue, experiment_directory, data_dir, start_point, end_point, optim, mano, optim_mano, fit, rot, obj, task):
    """The reference's worker signature (evaluate.py:19): one `queue.put([(name, chamfer_dist, joints_dist, verts_dist)])` per
    evaluated mesh of [start_point, end_point), so that a caller written against the reference's multiprocess driver
    (evaluate.py:200-228) can use this worker.  Names are taken in sorted order (the reference iterates os.listdir's); the `mano`,
    `optim_mano`, `fit` and `rot` modes read the image encoder's outputs and are not reproduced (NotImplementedError)."""
    if mano or optim_mano or fit or rot:
        raise NotImplementedError("the --mano / --optim_mano / --fit / --rot modes evaluate encoder outputs, which are outside this build")
    results, _ = evaluate(experiment_directory, data_dir, task, obj, optim, start_point, end_point)
    for rec in results:
        queue.put([rec])


def write_summary(experiment_directory, task, summary, n_pred, obj=False):
    """chamfer_hand.txt / chamfer_obj.txt (evaluate.py:254-310)."""
    summary = sorted(summary, reverse=True, key=lambda r: r[1])
    path = os.path.join(experiment_directory, "Eval_" + task, "chamfer_obj.txt" if obj else "chamfer_hand.txt")
    chamfer = [r[1] for r in summary]
    with open(path, "w") as f:
        f.write("summary of chamfer_dist\n")
        for r in summary:
            f.write("{}, {}, {}, {}\n".format(r[0], r[1], r[2] * 1000, r[3] * 1000))
        f.write("mean chamfer distance:{}\n".format(np.mean(chamfer)))
        f.write("median chamfer distance:{}\n".format(np.median(chamfer)))
        if obj:
            f.write("mean obj center error:{}\n".format(float("nan")))       # not computed (encoder outputs)
            f.write("mean obj corners error:{}\n".format(float("nan")))
        else:
            f.write("mean joints error:{}\n".format(float("nan")))
            f.write("mean verts error:{}\n".format(float("nan")))
        f.write("failure count:{}\n".format(n_pred - len(summary)))
    return path


def main(argv=None):
    p = argparse.ArgumentParser(description="Chamfer evaluation of reconstructed meshes (GPU nearest neighbours)")
    p.add_argument("--experiment", "-e", dest="experiment_directory", required=True)
    p.add_argument("--task", "-t", dest="task", default="obman")
    p.add_argument("--optim", dest="optim", action="store_true", help="align each mesh to its ground truth by the translate+scale ICP first")
    p.add_argument("--obj", dest="obj", action="store_true")
    p.add_argument("--data_root", default="data")
    args = p.parse_args(argv)
    data_source = os.path.join(args.data_root, args.task, "test")
    summary, n_pred = evaluate(args.experiment_directory, data_source, args.task, args.obj, args.optim)
    path = write_summary(args.experiment_directory, args.task, summary, n_pred, args.obj)
    print("".join(l for l in open(path) if l.startswith(("mean", "median", "failure"))))


if __name__ == "__main__":
    main()
