"""CPU restatement of the reference's eval-mode translate+scale ICP.  TEST INFRASTRUCTURE ONLY.

Follows deep_sdf/metrics/icp_trans_scale.py step by step (ICP_T_S.sample_mesh :19-31 without the random sampling,
run_icp_f :33-113, get_trans_scale :188-191, export_source_mesh :193-196) in float64, with scipy's cKDTree in place of
sklearn's KDTree (both return the exact nearest neighbour) and the 4-unknown least-squares system solved in closed
form instead of np.linalg.lstsq (same minimiser).  Pinned by tests/test_oracle_icp.py against
tests/golden/ref_icp.npz, which is produced by the reference class itself (tests/golden/make_icp_goldens.py).
"""
import numpy as np
from scipy.spatial import cKDTree


def normalise_source(points_source, points_target):
    """sample_mesh's normalisation (:25-31): move / scale the source samples onto the target's centroid and RMS radius."""
    ps, pt = np.asarray(points_source, np.float64), np.asarray(points_target, np.float64)
    offset_s = ps.mean(0)
    scale_s = np.sqrt(((ps - offset_s) ** 2).sum() / len(ps))
    offset_t = pt.mean(0)
    scale_t = np.sqrt(((pt - offset_t) ** 2).sum() / len(pt))
    return (ps - offset_s) / scale_s * scale_t + offset_t, dict(offset_source=offset_s, scale_source=scale_s,
                                                                 offset_target=offset_t, scale_target=scale_t)


def solve_scale_trans(X, Y):
    """argmin_{s,t} sum |s X_i + t - Y_i|^2  (the system of :76-107)."""
    n = len(X)
    xm, ym = X.mean(0), Y.mean(0)
    s = ((X * Y).sum() - n * (xm * ym).sum()) / ((X * X).sum() - n * (xm * xm).sum())
    return s, ym - s * xm


def run_icp_f(points_source, points_target, max_iter=100, stop_error=1e-3, stop_improvement=1e-5):
    """ICP_T_S.run_icp_f on already normalised source samples.  Returns (scale, trans[3], iterations, errors)."""
    ps, pt = np.asarray(points_source, np.float64), np.asarray(points_target, np.float64)
    tree_t, tree_s = cKDTree(pt), cKDTree(ps)
    scale, trans = 1.0, np.zeros(3)
    previous, errors, it = 1e8, [], 0
    for it in range(max_iter):
        q_s = ps * scale + trans
        idx_t = tree_t.query(q_s)[1]
        ct = pt[idx_t]
        idx_s = tree_s.query((pt - trans) / scale)[1]
        cs = ps[idx_s] * scale + trans
        error = np.sqrt((((q_s - ct) ** 2).sum() + ((pt - cs) ** 2).sum()) / (len(ps) + len(pt)))
        errors.append(error)
        if previous - error < stop_improvement:
            break
        previous = error
        if error < stop_error:
            break
        scale, trans = solve_scale_trans(np.concatenate([ps, ps[idx_s]]), np.concatenate([ct, pt]))
    return scale, trans, it + 1, errors


def icp_trans_scale(points_source, points_target, vertices, max_iter=100):
    """sample_mesh normalisation + run_icp_f + get_trans_scale + the vertex transform of export_source_mesh."""
    ps, n = normalise_source(points_source, points_target)
    scale, trans, iters, errors = run_icp_f(ps, points_target, max_iter)
    all_scale = n["scale_target"] * scale / n["scale_source"]
    all_trans = trans + n["offset_target"] * scale - n["offset_source"] * n["scale_target"] * scale / n["scale_source"]
    v = (np.asarray(vertices, np.float64) - n["offset_source"]) / n["scale_source"] * n["scale_target"] + n["offset_target"]
    return dict(scale=scale, trans=trans, iterations=iters, errors=errors, all_scale=all_scale, all_trans=all_trans,
                vertices=v * scale + trans)


def chamfer_sum(points_source, points_target):
    """The distance part of compute_trimesh_chamfer (deep_sdf/metrics/chamfer.py:217-229) with the same scipy cKDTree
    the reference imports: (gt_to_gen, gen_to_gt) mean squared nearest-neighbour distances; the metric is their sum."""
    ps, pt = np.asarray(points_source, np.float64), np.asarray(points_target, np.float64)
    one_distances, _ = cKDTree(ps).query(pt)
    gt_to_gen = np.mean(np.square(one_distances))
    two_distances, _ = cKDTree(pt).query(ps)
    gen_to_gt = np.mean(np.square(two_distances))
    return gt_to_gen, gen_to_gt
