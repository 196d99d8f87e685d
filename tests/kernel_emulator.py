"""Lane-accurate numpy emulation of the K0 (fold) and K1 (fused MLP) kernels' data flow.

Consumes the REAL packed images produced by the C++ packer (asdf_debug_pack_host) and walks the
same indices as alignsdf_amd/csrc/sdf_mlp_kernel.h, with v_mfma_f32_32x32x2_f32 modelled by its
operand maps:  A[i][k] = a[lane = i + 32 k],  B[k][j] = b[lane = j + 32 k],
               D[row(r, lane >> 5)][lane & 31] = acc[r][lane].
This is how the layouts are validated without a GPU (CPU test suite); it is test infrastructure.
"""
import ctypes

import numpy as np
import torch

from alignsdf_amd import _native
from alignsdf_amd.hip_decoder import _effective, head_point_feats

K_HIDDEN, K_LATENT = 512, 256
STAGE = 4096


def offsets(kp):
    """cst_offsets(kp) of sdf_layout.h."""
    a2 = 16 * kp * 64
    c0 = 2 * a2
    return dict(A0=0, A2=a2, C0=c0, B1=c0 + 512, C2=c0 + 768, B3=c0 + 1280, W4=c0 + 1792, W4B=c0 + 2304, B4=c0 + 2816,
                FLOATS=c0 + 2824)
LANE = np.arange(64)
HALF = LANE >> 5
ROW = np.array([[(r & 3) + 8 * (r >> 2) + 4 * h for h in range(2)] for r in range(16)])   # [r][half]


def pack_host(sd, point_feat_size, encode_style):
    """Run the C++ packer on a state dict; returns dict of numpy images."""
    L = _native.lib()
    pf = head_point_feats(point_feat_size, encode_style)
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    combined = "lin0.bias" in sd
    nerf = point_feat_size > 3 and encode_style == "nerf"
    mode = _native.FEATURES_NERF if nerf else _native.FEATURES_AFFINE
    kp = (point_feat_size + 1) // 2 if nerf else 2
    if combined:
        pf, prefixes = (point_feat_size,), ("lin",)
        spec = _native.DecoderSpec(256, 512, 1, (ctypes.c_int32 * 2)(pf[0], 0), (ctypes.c_int32 * 2)(2, 0), mode)
    else:
        prefixes = ("linh", "lino")
        spec = _native.DecoderSpec(256, 512, 2, (ctypes.c_int32 * 2)(*pf), (ctypes.c_int32 * 2)(1, 1), mode)
    heads = (_native.HeadParams * 2)()
    keep = []
    for hi, prefix in enumerate(prefixes):
        for layer in range(5):
            name = "%s%d" % (prefix, layer)
            w = _effective(sd, name)
            b = sd[name + ".bias"].float().contiguous()
            keep += [w, b]
            heads[hi].w[layer] = w.data_ptr()
            heads[hi].b[layer] = b.data_ptr()
    out = {
        "stream": np.zeros(256 * STAGE, np.float32), "wlat": np.zeros(2 * 2 * 512 * 256, np.float32),
        "wpt": np.zeros(2 * 2 * 512 * _native.MAX_POINT_FEATS, np.float32), "b02": np.zeros(2 * 2 * 512, np.float32),
        "cst": np.zeros(2 * offsets(kp)["FLOATS"], np.float32), "emb": np.zeros(2 * _native.MAX_POINT_FEATS * 4, np.float32),
    }
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _native.check(L.asdf_debug_pack_host(ctypes.byref(spec), heads, ptr(out["stream"]), ptr(out["wlat"]), ptr(out["wpt"]),
                                         ptr(out["b02"]), ptr(out["cst"]), ptr(out["emb"])), "asdf_debug_pack_host")
    out["pf"] = pf
    out["combined"] = combined
    out["kp"] = kp
    out["nerf"] = nerf
    if True:          # split-half image (sdf_mlp_f16_kernel.h)
        out["stream16"] = np.zeros(256 * STAGE * 2, np.uint16)
        out["cst16"] = np.zeros(2 * offsets(kp)["FLOATS"], np.float32)
        out["s2"] = np.zeros(2, np.float32)
        _native.check(L.asdf_debug_pack_host_f16(ctypes.byref(spec), heads, ptr(out["stream16"]), ptr(out["cst16"]), ptr(out["s2"])),
                      "asdf_debug_pack_host_f16")
        # ... and the W form's image of the same weights (sdf_mlp_f16w_kernel.h)
        out["stream16w"] = np.zeros(256 * STAGE * 2, np.uint16)
        _native.check(L.asdf_debug_pack_host_f16w(ctypes.byref(spec), heads, ptr(out["stream16w"])), "asdf_debug_pack_host_f16w")
    return out


def fold(pk, latent, embed=None):
    """K0 emulation: fills the per-sample parts of pk['cst'] (returns a new cst array)."""
    OFF = offsets(pk["kp"])
    cst = pk["cst"].copy().reshape(2, OFF["FLOATS"])
    wlat = pk["wlat"].reshape(2, 2, 512, 256)
    wpt = pk["wpt"].reshape(2, 2, 512, _native.MAX_POINT_FEATS)
    b02 = pk["b02"].reshape(2, 2, 512)
    emb = pk["emb"].reshape(2, _native.MAX_POINT_FEATS, 4).copy()
    if embed is not None:
        emb[:] = 0
        for h in range(len(pk["pf"])):
            emb[h, :pk["pf"][h]] = np.asarray(embed[h], np.float32)
    lat = np.asarray(latent, np.float32).reshape(256)
    for head in range(len(pk["pf"])):
        for layer in range(2):
            dot = (wlat[head, layer].astype(np.float32) @ lat).astype(np.float32)
            a = (wpt[head, layer, :, :pk["pf"][head]] @ emb[head, :pk["pf"][head]]).astype(np.float32)   # [512,4]
            c = (dot + b02[head, layer]).astype(np.float32)
            if not pk["nerf"]:
                c = (c + a[:, 3]).astype(np.float32)
            for row in range(512):
                t, rr = row >> 5, row & 31
                hh, r = (rr >> 2) & 1, (rr & 3) + 4 * (rr >> 3)
                cst[head, (OFF["C2"] if layer else OFF["C0"]) + (t * 2 + hh) * 16 + r] = c[row]
                for d in range(0 if pk["nerf"] else 4):
                    step, h2 = d >> 1, d & 1
                    cst[head, (OFF["A2"] if layer else OFF["A0"]) + (t * 2 + step) * 64 + h2 * 32 + rr] = a[row, d] if d < 3 else 0.0
    return cst


def mfma(a, b, acc):
    """acc[16][64] += A.B with the 32x32x2 operand maps (fp32 products, fp64 accumulate is avoided on purpose)."""
    A = a.reshape(2, 32)            # A[k][i]
    B = b.reshape(2, 32)            # B[k][j]
    D = (A[0][:, None] * B[0][None, :]).astype(np.float32)
    D = (D + (A[1][:, None] * B[1][None, :]).astype(np.float32)).astype(np.float32)
    rows = ROW[:, HALF]             # [16][64]
    cols = (LANE & 31)[None, :].repeat(16, 0)
    return (acc + D[rows, cols]).astype(np.float32)


def bias16(c, off, t):
    """load_bias16 for all lanes: [16][64]."""
    out = np.zeros((16, 64), np.float32)
    for h in range(2):
        out[:, HALF == h] = c[off + (t * 2 + h) * 16: off + (t * 2 + h) * 16 + 16][:, None]
    return out


def nerf_feature(f, x):
    """nerf_feature() of sdf_mlp_kernel.h for all points x [n,3]."""
    if f < 3:
        return x[:, f]
    j = f - 3
    r, d = j % 6, (j % 6) % 3
    arg = (x[:, d] * np.float32(1 << (j // 6))).astype(np.float32)
    return (np.sin(arg) if r < 3 else np.cos(arg)).astype(np.float32)


def run_wave(pk, cst, xyz32):
    """K1 emulation for one wave: xyz32 [32,3] -> (hand [32], obj [32])."""
    OFF = offsets(pk["kp"])
    KP = pk["kp"]
    x = np.asarray(xyz32, np.float32)
    pt = LANE & 31
    if pk["nerf"]:
        feats = [nerf_feature(f, x) if f < pk["pf"][0] else np.zeros(32, np.float32) for f in range(2 * KP)]
        bp = [np.where(HALF == 1, feats[2 * s + 1][pt], feats[2 * s][pt]).astype(np.float32) for s in range(KP)]
    else:
        bp = [np.where(HALF == 1, x[pt, 1], x[pt, 0]).astype(np.float32), np.where(HALF == 1, 0.0, x[pt, 2]).astype(np.float32)]
    stream = pk["stream"].reshape(256, 16, 64, 4)
    outs = []
    for head in range(len(pk["pf"])):
        c = cst[head]
        sbase = head * 128

        def layer(ntiles, stages_per_tile, s0, hin, bias_off, extra=None):
            res = []
            for t in range(ntiles):
                acc = bias16(c, bias_off, t)
                if extra is not None:
                    for s in range(KP):
                        acc = mfma(c[extra + (t * KP + s) * 64: extra + (t * KP + s) * 64 + 64], bp[s], acc)
                for q in range(stages_per_tile):
                    st = stream[sbase + s0 + t * stages_per_tile + q]
                    for g in range(16):
                        for j in range(4):
                            s = q * 64 + g * 4 + j
                            acc = mfma(st[g, :, j], hin[s >> 4][s & 15], acc)
                res.append(acc)
            return res

        h0 = [np.maximum(a, 0) for a in layer(16, 0, 0, None, OFF["C0"], OFF["A0"])]
        h1 = [np.maximum(a, 0) for a in layer(8, 4, 0, h0, OFF["B1"])]
        h2 = [np.maximum(a, 0) for a in layer(16, 2, 32, h1, OFF["C2"], OFF["A2"])]
        h3 = layer(16, 4, 64, h2, OFF["B3"])
        part = np.zeros(64, np.float32)
        for t in range(16):
            w = bias16(c, OFF["W4"], t)
            for r in range(16):
                part = (np.maximum(h3[t][r], 0) * w[r] + part).astype(np.float32)
        tot = part + part[LANE ^ 32]
        outs.append(np.tanh(tot + c[OFF["B4"]])[:32].astype(np.float32))
        if pk["combined"]:
            partb = np.zeros(64, np.float32)
            for t in range(16):
                w = bias16(c, OFF["W4B"], t)
                for r in range(16):
                    partb = (np.maximum(h3[t][r], 0) * w[r] + partb).astype(np.float32)
            totb = partb + partb[LANE ^ 32]
            outs.append(np.tanh(totb + c[OFF["B4"] + 1])[:32].astype(np.float32))
    return outs[0], outs[1]


# ---------------------------------------------------------------- split-half kernel (sdf_mlp_f16_kernel.h)
ACT_SCALE = np.float32(8.0)


def fold16(pk, latent, embed=None):
    """K0 into the split-half constants image: the fp32 fold with the layer-2 constants (c2, A2) times s2[head]."""
    OFF = offsets(pk["kp"])
    plain = fold(pk, latent, embed)
    cst = pk["cst16"].copy().reshape(2, OFF["FLOATS"])
    for head in range(len(pk["pf"])):
        cst[head, OFF["A0"]:OFF["A2"]] = plain[head, OFF["A0"]:OFF["A2"]]
        cst[head, OFF["C0"]:OFF["C0"] + 512] = plain[head, OFF["C0"]:OFF["C0"] + 512]
        cst[head, OFF["A2"]:OFF["C0"]] = plain[head, OFF["A2"]:OFF["C0"]] * pk["s2"][head]
        cst[head, OFF["C2"]:OFF["C2"] + 512] = plain[head, OFF["C2"]:OFF["C2"] + 512] * pk["s2"][head]
    return cst


def split_tile(acc, mul):
    """split_tile(): relu(acc) * mul -> (hi, lo) fp16 planes of the two K-blocks of a tile, each [8][64]."""
    t = (np.maximum(acc, 0) * np.float32(mul)).astype(np.float32)
    hi = t.astype(np.float16)
    lo = (t - hi.astype(np.float32)).astype(np.float16)
    return (hi[:8], lo[:8]), (hi[8:], lo[8:])


def mfma16(a, b, acc):
    """v_mfma_f32_32x32x16_f16: a, b [64][8] fp16 (lane, element): A[i = lane & 31][k = 8 (lane >> 5) + e],
    B[k = 8 (lane >> 5) + e][j = lane & 31]; products exact in fp32, fp32 accumulation."""
    A = np.zeros((32, 16), np.float32)
    B = np.zeros((16, 32), np.float32)
    for h in range(2):
        A[:, 8 * h:8 * h + 8] = a[32 * h:32 * h + 32].astype(np.float32)
        B[8 * h:8 * h + 8, :] = b[32 * h:32 * h + 32].astype(np.float32).T
    D = (A @ B).astype(np.float32)
    rows = ROW[:, HALF]
    cols = (LANE & 31)[None, :].repeat(16, 0)
    return (acc + D[rows, cols]).astype(np.float32)


def run_wave16(pk, cst, xyz32):
    """K1h emulation for one wave: xyz32 [32,3] -> per-output SDF values (hand, obj)."""
    OFF = offsets(2)
    x = np.asarray(xyz32, np.float32)
    pt = LANE & 31
    bp = [np.where(HALF == 1, x[pt, 1], x[pt, 0]).astype(np.float32), np.where(HALF == 1, 0.0, x[pt, 2]).astype(np.float32)]
    stream = pk["stream16"].view(np.float16).reshape(256, 8, 2, 64, 8)       # [stage][kblock][plane][lane][e]
    outs = []
    for head in range(len(pk["pf"])):
        c = cst[head]
        sbase = head * 128
        mul1, mul2, mul0 = c[OFF["B4"] + 2], c[OFF["B4"] + 3], c[OFF["B4"] + 4]

        def planes(accs, mul):
            """list of per-K-block (hi, lo) operand planes [64][8] from a layer's output tiles"""
            out = []
            for acc in accs:
                for hi, lo in split_tile(acc, mul):
                    out.append((hi.T.copy(), lo.T.copy()))          # [e][lane] -> [lane][e]
            return out

        def layer(ntiles, stages_per_tile, s0, xin, bias_off, extra=None):
            res = []
            for t in range(ntiles):
                acc = bias16(c, bias_off, t)
                if extra is not None:
                    for s in range(2):
                        acc = mfma(c[extra + (t * 2 + s) * 64: extra + (t * 2 + s) * 64 + 64], bp[s], acc)
                for q in range(stages_per_tile):
                    st = stream[sbase + s0 + t * stages_per_tile + q]
                    for kb in range(8):
                        xh, xl = xin[q * 8 + kb]
                        acc = mfma16(st[kb, 0], xl, acc)
                        acc = mfma16(st[kb, 1], xh, acc)
                        acc = mfma16(st[kb, 0], xh, acc)
                res.append(acc)
            return res

        l0 = []
        for t in range(16):
            acc = bias16(c, OFF["C0"], t)
            for s in range(2):
                acc = mfma(c[OFF["A0"] + (t * 2 + s) * 64: OFF["A0"] + (t * 2 + s) * 64 + 64], bp[s], acc)
            l0.append(acc)
        h0 = planes(l0, mul0)
        h1 = planes(layer(8, 4, 0, h0, OFF["B1"]), mul1)
        h2 = planes(layer(16, 2, 32, h1, OFF["C2"], OFF["A2"]), mul2)
        h3 = layer(16, 4, 64, h2, OFF["B3"])
        for woff, boff in ((OFF["W4"], OFF["B4"]),) + (((OFF["W4B"], OFF["B4"] + 1),) if pk["combined"] else ()):
            part = np.zeros(64, np.float32)
            for t in range(16):
                w = bias16(c, woff, t)
                for r in range(16):
                    part = (np.maximum(h3[t][r], 0) * w[r] + part).astype(np.float32)
            tot = part + part[LANE ^ 32]
            outs.append(np.tanh(tot + c[boff])[:32].astype(np.float32))
    return outs[0], outs[1]
