"""Sample pipeline with a per-image front end of the reference encoder's size in the loop:
    python tools/time_frontend_overlap.py [N] [samples]
prints ms per sample for codes resident on the device, and for codes produced per image by a ResNet-18-sized encoder whose
kernels are enqueued between the decoder passes (alignsdf_amd/frontend.py)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from alignsdf_amd import synthetic as syn
from alignsdf_amd.frontend import ResNet18Like, encoder_code_source
from alignsdf_amd.hip_decoder import HipSdfDecoder
from alignsdf_amd.reconstruct import pipelined_two_pass

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
specs = syn.specs_for("nerf3")
dec = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, 3, "nerf")
torch.manual_seed(0)
enc = ResNet18Like().cuda().eval()
images = [torch.rand(1, 3, 256, 256).pin_memory() for _ in range(4)]
src = encoder_code_source(enc, lambda name, i: images[i % 4])
resident = [torch.from_numpy(syn.latent_code(s)).cuda() for s in range(K + 2)]
with torch.no_grad():           # the encoder alone
    for _ in range(3):
        enc(images[0].cuda())
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(20):
        enc(images[i % 4].cuda(non_blocking=True))
    torch.cuda.synchronize(); t_enc = (time.perf_counter() - t) / 20


def run(stream, reps=5):
    """median of `reps` runs (shared hosts: a worker thread gets descheduled now and then)"""
    import numpy as np
    list(pipelined_two_pass(dec, specs, stream(0, 2), N))          # warm-up
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = list(pipelined_two_pass(dec, specs, stream(2, K), N))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K)
    return float(np.median(ts)), out


def resident_stream(first, count):
    for i in range(first, first + count):
        yield i, resident[i % len(resident)], None, None


def encoder_stream(first, count):
    for i in range(first, first + count):
        lat, mano, obj = src("img%d" % i, i)
        yield i, lat, mano, obj


# round 3: the images come from JPEG FILES through the prefetching loader (decode + crop + normalise on a worker thread, pinned
# upload on a side stream, two samples ahead)
import os, tempfile
import numpy as np
from PIL import Image
from alignsdf_amd.frontend import ImageFilePrefetcher
root = tempfile.mkdtemp()
names = ["%08d" % i for i in range(K + 2)]
rng = np.random.default_rng(0)
for n in names:
    Image.fromarray((rng.random((480, 640, 3)) * 255).astype(np.uint8)).save(os.path.join(root, n + ".jpg"), "JPEG", quality=90)


def file_stream_factory():
    pre = ImageFilePrefetcher(root, names, image_size=(256, 256))
    fsrc = encoder_code_source(enc, pre)

    def file_stream(first, count):
        for i in range(first, first + count):
            lat, mano, obj = fsrc(names[i], i)
            yield i, lat, mano, obj
    return pre, file_stream


t_res, _ = run(resident_stream)
t_encp, out = run(encoder_stream)
pre, file_stream = file_stream_factory()
t_file, out_f = run(file_stream)
pre.close()
decoded = (K + 2) + 5 * K
print("N=%d, %d samples: images decoded from 640 x 480 JPEG files by the prefetching loader -> encoder in the loop %.2f ms/sample "
      "(%.2f ms over codes resident; %.1f ms of decode / crop / normalise per image, on the worker thread)" % (
          N, K, 1e3 * t_file, 1e3 * (t_file - t_res), 1e3 * pre.decode_seconds / decoded))
print("N=%d, %d samples: codes resident %.2f ms/sample; ResNet-18-sized encoder per image in the loop %.2f ms/sample "
      "(encoder alone %.2f ms/image, so %.2f ms of it is exposed); F_hand of the last sample %d" % (
          N, K, 1e3 * t_res, 1e3 * t_encp, 1e3 * t_enc, 1e3 * (t_encp - t_res), out[-1][1]["F_hand"]))
