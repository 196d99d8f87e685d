"""How many voxels a narrow-band fine pass would have to re-evaluate: corners of the cells whose one-plane signs are mixed or
uncertain (|value| < tau), on the zoom-cube lattice of a few synthetic samples."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignsdf_amd import synthetic as syn
from alignsdf_amd.hip_decoder import HipSdfDecoder
from alignsdf_amd.utils.mesh import zoom_cube_from_bboxes
N = 256
for tag in ("nerf3", "both9"):
    specs = syn.specs_for(tag)
    hip = HipSdfDecoder(syn.full_state_dict(tag), 256, specs["PointFeatSize"], specs["EncodeStyle"])
    from alignsdf_amd.utils.utils import sample_embedding
    for sample in (0, 1, 2):
        mano = obj = None
        if specs["EncodeStyle"] != "nerf":
            m, o = syn.pose_inputs(sample)
            mano = {k: torch.from_numpy(v).cuda() for k, v in m.items()}
            obj = {k: torch.from_numpy(v).cuda() for k, v in o.items()}
        hip.set_sample(torch.from_numpy(syn.latent_code(sample)).cuda(), sample_embedding(specs, mano, obj, hip.combined))
        vs = 2.0 / (N - 1)
        b = hip.decode_grid(N, [-1.0, -1.0, -1.0], vs)[2].cpu().numpy()
        nvs, norg = zoom_cube_from_bboxes([(b[0:3], b[3:6], int(b[6])), (b[8:11], b[11:14], int(b[14]))], N, vs)
        eh, eo, _ = hip.decode_grid(N, norg.tolist(), nvs.item())
        rec, fh, fo = hip._box_launch(N, norg.tolist(), nvs.item(), 0, True, True, 1e-7)
        for name, e, f in (("hand", eh, fh), ("obj", eo, fo)):
            err = (e - f).abs().max().item()
            for mult in (2.0, 4.0):
                tau = mult * 4.5e-4
                pos = f >= tau
                neg = f < -tau
                unc = ~(pos | neg)
                def cells(x):     # reduce over the 8 corners of every cell
                    a = x[:-1, :-1, :-1]
                    outs = [x[i:N - 1 + i, j:N - 1 + j, k:N - 1 + k] for i in (0, 1) for j in (0, 1) for k in (0, 1)]
                    return outs
                anypos = torch.stack(cells(pos)).any(0); anyneg = torch.stack(cells(neg)).any(0); anyunc = torch.stack(cells(unc)).any(0)
                cand = (anypos & anyneg) | anyunc
                mark = torch.zeros((N, N, N), dtype=torch.bool, device=f.device)
                for i in (0, 1):
                    for j in (0, 1):
                        for k in (0, 1):
                            mark[i:N - 1 + i, j:N - 1 + j, k:N - 1 + k] |= cand
                active = torch.stack(cells(e >= 0)).any(0) & torch.stack(cells(e < 0)).any(0)
                print("%s s%d %s: err %.2e tau %.1e | uncertain voxels %d | candidate cells %d (exact active %d) | voxels to re-evaluate %d = %.2f %% | value range [%.3f %.3f]" % (
                    tag, sample, name, err, tau, int(unc.sum()), int(cand.sum()), int(active.sum()), int(mark.sum()), 100.0 * int(mark.sum()) / N ** 3, e.min().item(), e.max().item()), flush=True)
    hip.close()
