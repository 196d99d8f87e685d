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
import torch
import torch.nn as nn


def model_output_code_source(decode_fn, device="cuda"):
    """decode_fn(name, index) -> (latent [1, L] (or a [1, C, H, W] feature map under PixelAlign), mano_results or None,
    obj_results or None) exactly as utils.decode_model_output returns them (utils/utils.py:620-625)."""
    def source(name, index):
        latent, mano_results, obj_results = decode_fn(name, index)
        lat = latent.detach().to(device=device, dtype=torch.float32)
        mano = obj = None
        if mano_results is not None:
            # the decoder path reads global_trans / rot_center (kinematic_embedding) and, under PixelAlign, joints
            mano = {k: mano_results[k].detach().to(device=device, dtype=torch.float32)
                    for k in ("global_trans", "rot_center", "joints") if k in mano_results}
        if obj_results is not None:
            obj = {"obj_trans": obj_results["obj_trans"].detach().to(device=device, dtype=torch.float32)}
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
