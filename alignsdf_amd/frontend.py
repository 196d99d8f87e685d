"""Front-end hooks of the reconstruction driver: how per-image codes reach the hot path.

The reference computes them per image with `utils.decode_model_output(model, input, meta_input, specs)`
(utils/utils.py:575-625: ResNet-18 encoder, MANO branch, object-pose branch) and hands
`(latent, mano_results, obj_results)` to create_mesh_combined_decoder (reconstruct.py:84-93).  The encoder itself is outside
this build (it needs the datasets and MANO assets); what is here:

* `model_output_code_source(decode_fn)` - adapter from anything with decode_model_output's return contract to the
  `code_source(name, index)` the drivers consume: picks the keys the hot path reads (utils/utils.py:387,393,414), keeps
  everything on the device, never synchronises.
* `ResNet18Like` + `encoder_code_source` - a stand-in workload of the reference encoder's size (ResNet-18 trunk on a
  256 x 256 image -> 256-d latent, random weights) to measure how a per-image front end shares the GPU with the decoder
  passes.  A decoder pass owns every SIMD's whole register file, so nothing else runs WHILE it runs: the front end's kernels
  are enqueued on the compute stream where the sample pipeline fetches sample k+2 - between pass 1 and pass 2 of sample k+1 -
  and execute in that gap (tools/time_frontend_overlap.py, profiles/r02_frontend_overlap.txt).
"""
import os

import torch
import torch.nn as nn


def model_output_code_source(decode_fn, device="cuda"):
    """decode_fn(name, index) -> (latent [1, L] (or a [1, C, H, W] feature map under PixelAlign), mano_results or None,
    obj_results or None) exactly as utils.decode_model_output returns them (utils/utils.py:620-625)."""
    def on_device(t):
        # codes that are still on the HOST stay there (round 6): the HIP decoder's set_sample reads them from pinned memory in stream
        # order and the module path uploads what it needs itself - a plain .to(device) of pageable memory would be a synchronous copy
        # behind every pass already queued, and the runtime's blit of 1 KB cannot get a wave slot under a persistent sweep
        t = t.detach()
        if t.device.type == "cpu":
            return t.to(torch.float32)
        return t.to(device=device, dtype=torch.float32)

    def source(name, index):
        latent, mano_results, obj_results = decode_fn(name, index)
        lat = on_device(latent)
        mano = obj = None
        if mano_results is not None:
            # the decoder path reads global_trans / rot_center (kinematic_embedding) and, under PixelAlign, joints
            mano = {k: on_device(mano_results[k]) for k in ("global_trans", "rot_center", "joints") if k in mano_results}
        if obj_results is not None:
            obj = {"obj_trans": on_device(obj_results["obj_trans"])}
        return lat, mano, obj
    return source


class _Block(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.c1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.b1 = nn.BatchNorm2d(cout)
        self.c2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.b2 = nn.BatchNorm2d(cout)
        self.down = None if stride == 1 and cin == cout else nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = torch.relu(self.b1(self.c1(x)))
        y = self.b2(self.c2(y))
        return torch.relu(y + (x if self.down is None else self.down(x)))


class ResNet18Like(nn.Module):
    """The layer shapes of a ResNet-18 trunk (7x7/2 stem, 3x3/2 max-pool, 4 stages of 2 basic blocks at 64 / 128 / 256 / 512
    channels, global average pool) with a linear head to the latent size: 1.8 GFLOP per 256 x 256 image, the size of the
    reference's encoder (networks/resnet.py:155-194).  Random weights - a workload, not a model."""

    def __init__(self, latent_size=256):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1))
        blocks, cin = [], 64
        for cout, stride in ((64, 1), (128, 2), (256, 2), (512, 2)):
            blocks += [_Block(cin, cout, stride), _Block(cout, cout, 1)]
            cin = cout
        self.layers = nn.Sequential(*blocks)
        self.head = nn.Linear(512, latent_size)

    def forward(self, x):
        x = self.layers(self.stem(x))
        return 0.1 * torch.tanh(self.head(x.mean((2, 3))))


def encoder_code_source(encoder, image_of, device="cuda"):
    """code_source that runs `encoder` on `image_of(name, index)` ([1, 3, H, W], host or device) on the current stream,
    without synchronising: the latent stays on the device and K0 reads it from there."""
    encoder = encoder.to(device).eval()

    def source(name, index):
        with torch.no_grad():
            img = image_of(name, index).to(device, non_blocking=True)
            return encoder(img), None, None
    return source


def quick_gil_handover(interval=float(os.environ.get("ASDF_GIL_INTERVAL", 5e-4))):
    """Worker threads (image decode, ground-truth parsing, PLY writes) hold the interpreter lock between their C calls; with
    CPython's default 5 ms switch interval the main thread - whose job is to keep the GPU's queue full - can wait that long
    for it every time.  0.5 ms makes a worker hand it back promptly."""
    import sys
    previous = sys.getswitchinterval()
    if previous > interval:
        sys.setswitchinterval(interval)
    return previous


def restore_gil_handover(previous):
    """Undo quick_gil_handover (the switch interval is a PROCESS-wide setting: whoever shortened it for its worker threads puts the
    caller's value back when the threads are gone)."""
    import sys
    if previous is not None and previous > sys.getswitchinterval():
        sys.setswitchinterval(previous)


IMAGENET_MEAN = (0.485, 0.456, 0.406)      # transforms.Normalize of utils/data.py:220
IMAGENET_STD = (0.229, 0.224, 0.225)


def load_image_tensor(path, image_size, out=None):
    """One encoder input as utils.data.ImagesInput.__getitem__ prepares it (utils/data.py:235-244): decode the file (RGB), take the
    centre crop of `image_size` = (H, W) (generate_patch_image with a centre box, no scale / rotation; outside the picture is
    black), scale to [0, 1], normalise with the ImageNet statistics.  Returns (or fills `out`, a [1, 3, H, W] fp32 tensor)."""
    import numpy as np
    from PIL import Image
    H, W = int(image_size[0]), int(image_size[1])
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    h, w = rgb.shape[:2]
    top, left = (h - H) // 2, (w - W) // 2
    crop = np.zeros((H, W, 3), dtype=np.uint8)
    ys, xs = max(top, 0), max(left, 0)
    ye, xe = min(top + H, h), min(left + W, w)
    crop[ys - top:ye - top, xs - left:xe - left] = rgb[ys:ye, xs:xe]
    x = torch.from_numpy(crop).permute(2, 0, 1).to(torch.float32).div_(255.0)
    x = (x - torch.tensor(IMAGENET_MEAN).view(3, 1, 1)) / torch.tensor(IMAGENET_STD).view(3, 1, 1)
    if out is None:
        return x.unsqueeze(0)
    out[0].copy_(x)
    return out


class ImageFilePrefetcher:
    """`image_of(name, index)` for encoder_code_source, from image FILES: the counterpart of the reference's
    ImagesInput + DataLoader(num_workers=1) (reconstruct.py:54-66).  `ahead` samples in advance a worker thread decodes
    <image_root>/<name><ext> (PIL), crops / normalises it into a pinned [1, 3, H, W] buffer and uploads it on a side stream;
    the call for sample k makes the compute stream wait for that upload's event and hands the device tensor over - the host
    never waits for the device, and the decode of image k+2 runs while the GPU is inside the passes of sample k."""

    def __init__(self, image_root, names, image_size=(256, 256), ext=".jpg", ahead=2, device="cuda"):
        from concurrent.futures import ThreadPoolExecutor
        quick_gil_handover()
        self.root, self.names, self.size, self.ext, self.ahead = image_root, list(names), tuple(image_size), ext, max(1, int(ahead))
        self.device = torch.device(device)
        self.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="asdf-img")
        self.side = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.order = {n: i for i, n in enumerate(self.names)}
        self.jobs = {}
        self.decode_seconds = 0.0
        # a ring of pinned staging buffers, allocated ONCE: pinning memory per image costs a millisecond and takes a runtime lock
        # the main thread's kernel launches queue behind
        self.ring = []
        if self.side is not None:
            self.ring = [[torch.empty((1, 3) + self.size, dtype=torch.float32).pin_memory(), None] for _ in range(self.ahead + 3)]
        self.turn = 0

    def _load(self, name):
        import os
        import time
        t0 = time.perf_counter()
        if self.side is None:
            host = load_image_tensor(os.path.join(self.root, name + self.ext), self.size)
            self.decode_seconds += time.perf_counter() - t0
            return host, None
        slot = self.ring[self.turn % len(self.ring)]
        self.turn += 1
        if slot[1] is not None:
            slot[1].synchronize()         # the upload that last used this buffer (ahead + 3 images ago) is long done
        load_image_tensor(os.path.join(self.root, name + self.ext), self.size, out=slot[0])
        self.decode_seconds += time.perf_counter() - t0
        with torch.cuda.stream(self.side):
            dev = slot[0].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.side)
        slot[1] = ev
        return dev, ev

    def _schedule(self, name):
        if name not in self.jobs:
            self.jobs[name] = self.pool.submit(self._load, name)

    def __call__(self, name, index):
        k = self.order.get(name)
        self._schedule(name)
        if k is not None:
            for nxt in self.names[k + 1:k + 1 + self.ahead]:
                self._schedule(nxt)
        dev, ev = self.jobs.pop(name).result()
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            dev.record_stream(torch.cuda.current_stream(self.device))
        return dev

    def close(self):
        self.pool.shutdown(wait=True)
