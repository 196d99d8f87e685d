"""Train the decoders of the synthetic GRASP family (alignsdf_amd/synthetic.py: grasp_scene / grasp_sdf) on the CPU and commit
them as fixtures:  tests/golden/grasp_decoder_<tag>.npz  (state dict of a full-size SeparateDecoder + the 16 per-sample latent codes).

Why: every decoder the one-plane sweeps, their certificate and the list capacities had been tried
on was a random network with a ridge-fitted last layer approximating a sphere and a box far apart.  Here EVERY layer is fitted, the
way the reference trains (train.py:392-396, :512-536, :625):

    * the module of networks/model.py:192-350 in TRAINING mode - weight-norm on layers 0-3, dropout p = 0.2 behind them;
    * Adam on   sum |clamp(pred, +-0.05) - clamp(sdf, +-0.05)| / points   per head (ClampingDistance 0.05, enforce_minmax);
    * the latent code of a sample is a free parameter (the reference's comes from its image encoder, which is out of scope:
      auto-decoding the 16 scenes is the stand-in), initialised 0.1 N(0, 1) like synthetic.latent_code;
    * points: 70 % within the clamping distance of one of the two surfaces (DeepSDF-style surface samples), 30 % uniform in the cube;
    * "grasp9": the point features are utils.utils.kinematic_embedding's ((9, "both"): world, wrist-frame and object-frame
      coordinates, utils/utils.py:376-430) under each scene's own pose (synthetic.grasp_pose_inputs).

The training is NOT bit-reproducible across machines (multi-threaded fp32 reductions); the committed fixture is the pin, this
script is its provenance.  It runs on the CPU (~0.7 s per step of 12 k points on 8 cores: 3000 steps in 35 min per tag) or, with
--device cuda, on PyTorch-ROCm (stock torch modules and optimiser: plumbing, none of the product's kernels).  The committed
fixtures were produced on an MI355X box with

    python tests/golden/train_grasp_decoders.py grasp3 grasp9 --device cuda --steps 20000 --per-scene 2048 --out gpurun_out

(a few thousand steps on the CPU give blobs with the right statistics; the thin fingers need the longer run).

Usage:  python tests/golden/train_grasp_decoders.py [grasp3] [grasp9] [--steps 3000] [--threads 8] [--device cpu|cuda] [--out DIR]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from alignsdf_amd import synthetic as syn  # noqa: E402
from alignsdf_amd.hip_decoder import kinematic_affine  # noqa: E402   (float64 numpy; no device involved)
from alignsdf_amd.networks.model import SeparateDecoder  # noqa: E402

POOL_NEAR, POOL_UNIFORM = 160000, 60000


_POOLS = {}


def scene_pool(sample):
    """Training points of one scene with both targets: uniform inside a band of +-0.06 around either surface, plus the cube."""
    if sample not in _POOLS:
        _POOLS[sample] = _scene_pool(sample)
    return _POOLS[sample]


def _scene_pool(sample):
    box = syn.uniform((int(os.environ.get("ASDF_GRASP_POOL", 3600000)), 3), 41000 + sample, -0.7, 0.7)
    h, o = (np.concatenate(c) for c in zip(*(syn.grasp_sdf(box[k:k + 600000], sample) for k in range(0, len(box), 600000))))
    near_h = np.flatnonzero(np.abs(h) < 0.06)[:POOL_NEAR]
    near_o = np.flatnonzero(np.abs(o) < 0.06)[:POOL_NEAR * 2 // 3]
    cube = syn.uniform((POOL_UNIFORM, 3), 42000 + sample, -1.0, 1.0)
    hc, oc = syn.grasp_sdf(cube, sample)
    pts = np.concatenate([box[near_h], box[near_o], cube])
    return pts.astype(np.float32), np.concatenate([h[near_h], h[near_o], hc]).astype(np.float32), \
        np.concatenate([o[near_h], o[near_o], oc]).astype(np.float32), (len(near_h), len(near_o), len(cube))


def train(tag, steps, per_scene, lr, seed, device, out_dir):
    torch.manual_seed(seed)
    pf, style = syn.GRASP_TAGS[tag]
    specs = syn.specs_for(tag)
    dec = SeparateDecoder(256, pf, style, **specs["NetworkSpecs"], use_classifier=False).train().to(device)
    latents = torch.nn.Parameter((0.1 * torch.randn(syn.GRASP_SAMPLES, 256)).to(device))
    pools, affine = [], []
    t0 = time.time()
    for s in range(syn.GRASP_SAMPLES):
        pts, h, o, counts = scene_pool(s)
        pools.append((torch.from_numpy(pts).to(device), torch.from_numpy(h).to(device), torch.from_numpy(o).to(device), counts))
        if style == "both":
            mano, obj = syn.grasp_pose_inputs(s)
            eh, eo = kinematic_affine(pf, style, specs["SdfScaleFactor"], {k: torch.from_numpy(v) for k, v in mano.items()},
                                      {k: torch.from_numpy(v) for k, v in obj.items()})
            # decoder input = [latent | m sf/2 (3) | q0 sf/2 (3) | o sf/2 (3)]: hand rows 0..5, object rows 3..5 (networks/model.py:297-299)
            e = np.concatenate([eh, eo[3:]], 0)
            affine.append(torch.from_numpy(e.astype(np.float32)).to(device))
        print("scene %2d: pool %s  (%.0f s)" % (s, counts, time.time() - t0), flush=True)
    opt = torch.optim.Adam([{"params": dec.parameters(), "lr": lr}, {"params": [latents], "lr": 4 * lr}])
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda k: 0.5 ** (k / (steps / 4.0)))          # the reference halves on a step schedule
    gen = torch.Generator(device=device).manual_seed(seed + 1)
    clamp = syn.GRASP_CLAMP
    t0 = time.time()
    for step in range(steps):
        xs, th, to, rows = [], [], [], []
        for s, (pts, h, o, counts) in enumerate(pools):
            nh, no, nc = counts
            k = per_scene
            idx = torch.cat([torch.randint(0, nh, (int(0.42 * k),), generator=gen, device=device),
                             nh + torch.randint(0, no, (int(0.28 * k),), generator=gen, device=device),
                             nh + no + torch.randint(0, nc, (k - int(0.42 * k) - int(0.28 * k),), generator=gen, device=device)])
            p = pts[idx]
            feats = p if style == "nerf" else p @ affine[s][:, :3].T + affine[s][:, 3]
            xs.append(feats)
            th.append(h[idx])
            to.append(o[idx])
            rows.append(torch.full((k,), s, dtype=torch.long, device=device))
        rows = torch.cat(rows)
        inputs = torch.cat([latents[rows], torch.cat(xs)], 1)
        ph, po, _ = dec(inputs)
        th, to = torch.cat(th).clamp(-clamp, clamp), torch.cat(to).clamp(-clamp, clamp)
        n = float(len(rows))
        # enforce_minmax clamps the PREDICTION too (train.py:520-524): no gradient wherever the network is beyond the clamp, on
        # either side.  The reference lives with that over its 1600 epochs; a 3000-step fit does not, so the prediction's clamp is
        # opened to 4 x the distance for the first 40 % of the steps
        cp = clamp if step >= 0.4 * steps else 4 * clamp
        loss_h = (ph[:, 0].clamp(-cp, cp) - th).abs().sum() / n
        loss_o = (po[:, 0].clamp(-cp, cp) - to).abs().sum() / n
        loss = loss_h + loss_o
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        sched.step()
        if step % max(steps // 60, 1) == 0 or step == steps - 1:
            print("%s step %5d  L1 hand %.5f obj %.5f  |latent| %.3f  lr %.2e  %.2f s/step" % (
                tag, step, loss_h.item(), loss_o.item(), latents.norm(dim=1).mean().item(), sched.get_last_lr()[0],
                (time.time() - t0) / (step + 1)), flush=True)
    dec.eval()
    out = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in dec.state_dict().items()}
    out["latents"] = latents.detach().cpu().numpy().astype(np.float32)
    # evaluation-mode fit on fresh points, per scene (what the surfaces will look like)
    with torch.no_grad():
        for s, (pts, h, o, counts) in enumerate(pools):
            p = pts[:40000]
            feats = p if style == "nerf" else p @ affine[s][:, :3].T + affine[s][:, 3]
            ph, po, _ = dec(torch.cat([latents[s:s + 1].expand(len(p), -1), feats], 1))
            eh = (ph[:, 0] - h[:40000]).abs()[h[:40000].abs() < 0.03]
            eo = (po[:, 0] - o[:40000]).abs()[o[:40000].abs() < 0.03]
            print("scene %2d eval: |pred - sdf| within 0.03 of the surface: hand mean %.4f max %.4f, obj mean %.4f max %.4f" % (
                s, eh.mean(), eh.max(), eo.mean() if len(eo) else 0, eo.max() if len(eo) else 0), flush=True)
    os.makedirs(out_dir, exist_ok=True)
    np.savez(os.path.join(out_dir, "grasp_decoder_%s.npz" % tag), **out)
    print("wrote", os.path.join(out_dir, "grasp_decoder_%s.npz" % tag), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("tags", nargs="*", default=["grasp3", "grasp9"])
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--per-scene", type=int, default=768)
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    for t in args.tags:
        train(t, args.steps, args.per_scene, args.lr, 20260929 + (t == "grasp9"), torch.device(args.device), args.out)
