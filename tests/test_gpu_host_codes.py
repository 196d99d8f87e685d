"""Per-sample codes that live on the HOST (round 6): the reference's codes come out of the encoder once per sample
(reconstruct.py:83-84); a run from saved codes has them in host memory.  HipSdfDecoder.set_sample stages a host-side latent (and the
affine embedding) in a pinned slot and asdf_decoder_set_sample_host's one-workgroup staging launch reads it over the link in stream
order - no copy engine, no runtime blit kernel, no side stream, and the call does not wait for the queue."""
import time

import numpy as np
import pytest
import torch

from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _decoder(tag):
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    specs = syn.specs_for(tag)
    return HipSdfDecoder(syn.full_state_dict(tag), 256, specs["PointFeatSize"], specs["EncodeStyle"]), specs


@pytest.mark.parametrize("tag", ["nerf3", "both9"])
def test_host_codes_give_the_device_codes_volumes_bit_for_bit(tag):
    """The same samples bound from device tensors and from host tensors: identical volumes and boxes, through more binds than the staging
    ring has slots, with the samples interleaved (a slot must not be rewritten while its staging launch is still queued)."""
    from alignsdf_amd import hip_decoder as hd
    from alignsdf_amd.utils.utils import sample_embedding
    hip, specs = _decoder(tag)
    N = 32
    origin, vs = [-1.0, -1.0, -1.0], 2.0 / (N - 1)
    want = {}
    for s in range(6):
        lat, m, o = syn.sample_inputs(tag, s)
        mano = {k: torch.from_numpy(v).cuda() for k, v in m.items()} if m is not None else None
        obj = {k: torch.from_numpy(v).cuda() for k, v in o.items()} if o is not None else None
        hip.set_sample(torch.from_numpy(lat).cuda(), sample_embedding(specs, mano, obj, hip.combined))
        want[s] = hip.decode_grid(N, origin, vs)
    torch.cuda.synchronize()
    # a long queue in front, so that the binds below are all ENQUEUED before the first of them runs
    a = torch.randn(4096, 4096, device="cuda")
    for _ in range(20):
        a = (a @ a) * 1e-3
    got = []
    for k in range(3 * hd.CODE_SLOTS):
        s = k % 6
        lat, m, o = syn.sample_inputs(tag, s)
        mano = {k2: torch.from_numpy(v) for k2, v in m.items()} if m is not None else None            # host tensors throughout
        obj = {k2: torch.from_numpy(v) for k2, v in o.items()} if o is not None else None
        hip.set_sample(torch.from_numpy(lat), sample_embedding(specs, mano, obj, hip.combined))
        got.append((s, hip.decode_grid(N, origin, vs, check_range=False)))
    torch.cuda.synchronize()
    for s, g in got:
        for k in (0, 1, 2):
            assert torch.equal(want[s][k], g[k]), (tag, s, k)
    hip.close()


def test_binding_host_codes_does_not_wait_for_the_queue():
    """set_sample of a host-side latent returns while a long kernel queue is still running (a `.to(device)` of pageable memory is a
    synchronous copy in stream order: it waited for everything queued - 50 ms per sample in round 5's eval-mode flow)."""
    hip, specs = _decoder("nerf3")
    lat = torch.from_numpy(syn.sample_inputs("nerf3", 0)[0])
    hip.set_sample(lat)
    hip.decode_grid(32, [-1.0] * 3, 2.0 / 31)
    torch.cuda.synchronize()
    a = torch.randn(4096, 4096, device="cuda")
    a = (a @ a) * 1e-3                                     # (the first GEMM of a process loads its library: not part of the measurement)
    torch.cuda.synchronize()
    for _ in range(60):
        a = (a @ a) * 1e-3
    t0 = time.perf_counter()
    for _ in range(8):
        hip.set_sample(lat)
    call = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    assert total > 3 * call, "binding host codes waited for the queue: %.1f ms of calls, queue drained after %.1f ms" % (1e3 * call, 1e3 * total)
    hip.close()


def test_code_sources_keep_host_codes_on_the_host_and_the_pipeline_takes_them(tmp_path):
    """synthetic_code_source / npz_code_source hand out CPU tensors by default; the sample pipeline's meshes are those of device-side
    codes, vertex for vertex."""
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.reconstruct import pipelined_two_pass, synthetic_code_source
    tag, N = "both9", 48
    specs = syn.specs_for(tag)
    dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()})
    out = {}
    for on_host in (True, False):
        src = synthetic_code_source(tag, "cuda", on_host=on_host)
        items = [(s,) + src("s%d" % s, s) for s in range(5)]
        assert all((it[1].device.type == "cpu") == on_host and (it[2]["rot_center"].device.type == "cpu") == on_host for it in items)
        out[on_host] = {s: r for s, r in pipelined_two_pass(dec, specs, iter(items), N)}
    for s in range(5):
        a, b = out[True][s], out[False][s]
        assert a["origin"] == b["origin"] and float(a["voxel_size"]) == float(b["voxel_size"])
        for part in ("hand", "obj"):
            assert torch.equal(a["verts_" + part], b["verts_" + part]) and torch.equal(a["faces_" + part], b["faces_" + part])
