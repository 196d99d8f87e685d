"""Checkpoint-compatible containers for AlignSDF's SDF decoders.

`SeparateDecoder` has the constructor signature, attribute names and state-dict keys of the reference
class (networks/model.py:191-282: `lin{h,o}{0..4}.weight_g|weight_v|bias`, plain `weight` on the last
layer), so `module.decoder.*` tensors of a reference `latest.pth` load unchanged.  It exists to carry
parameters to the HIP path (alignsdf_amd.hip_decoder.HipSdfDecoder).  Its `forward` is a plain PyTorch
evaluation of the same network: the mesh-extraction path calls it (on PyTorch-ROCm device tensors, through
alignsdf_amd.torch_decoder) only for the variants the HIP kernels do not cover - `use_tanh`, the LayerNorm
form (`weight_norm` false), `xyz_in_all` - like the reference calls its module.
"""
import warnings

import torch
import torch.nn as nn


def _linear(n_in, n_out, normed):
    lin = nn.Linear(n_in, n_out)
    if normed:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lin = nn.utils.weight_norm(lin)     # parameters weight_g [out,1], weight_v [out,in]
    return lin


class SeparateDecoder(nn.Module):
    """Two independent MLP heads (hand, object) over [latent | point features]."""

    def __init__(self, latent_size, point_feat_size, encode_style, dims, num_class=6, dropout=None, dropout_prob=0.0,
                 norm_layers=(), latent_in=(), weight_norm=False, xyz_in_all=None, use_tanh=False, latent_dropout=False,
                 use_classifier=False):
        super().__init__()
        self.latent_size, self.point_feat_size, self.encode_style = latent_size, point_feat_size, encode_style
        self.norm_layers, self.latent_in, self.weight_norm = tuple(norm_layers or ()), tuple(latent_in), weight_norm
        self.dropout, self.dropout_prob = dropout, dropout_prob
        self.use_classifier, self.use_tanh, self.xyz_in_all, self.latent_dropout = use_classifier, use_tanh, xyz_in_all, latent_dropout
        self.num_class = num_class
        widths = {"nerf": (point_feat_size, point_feat_size), "hand": (point_feat_size, 3), "obj": (3, point_feat_size),
                  "both": (point_feat_size - 3, 6)}[encode_style]          # networks/model.py:212-223
        self.head_point_feats = widths
        for prefix, pf in zip(("linh", "lino"), widths):
            sizes = [latent_size + pf] + list(dims) + [1]
            self.num_layers = len(sizes)
            for layer in range(len(sizes) - 1):
                n_out = sizes[layer + 1] - sizes[0] if (layer + 1) in self.latent_in else sizes[layer + 1]
                setattr(self, prefix + str(layer), _linear(sizes[layer], n_out, weight_norm and layer in self.norm_layers))
                if not weight_norm and layer in self.norm_layers:          # LayerNorm form: bnh* / bno* (networks/model.py:252-253)
                    setattr(self, "bn" + prefix[3] + str(layer), nn.LayerNorm(n_out))
        if use_classifier:      # on the hand head's last hidden activation (networks/model.py:257-259)
            self.classifier_head = nn.Linear(list(dims)[-1], num_class)

    def head_inputs(self, inputs):
        """Per-head input slices (networks/model.py:288-299)."""
        L = self.latent_size
        if self.encode_style == "nerf":
            return inputs, inputs
        if self.encode_style == "hand":
            return inputs, inputs[:, :L + 3]
        if self.encode_style == "obj":
            return inputs[:, :L + 3], inputs
        return inputs[:, :-3], torch.cat([inputs[:, :L + 3], inputs[:, -3:]], 1)

    def _head(self, prefix, x0, classify=False):
        x = x0
        last = self.num_layers - 2
        scores = None
        for layer in range(self.num_layers - 1):
            if classify and layer == last:
                scores = self.classifier_head(x)
            if layer in self.latent_in:
                x = torch.cat([x, x0], 1)
            x = getattr(self, prefix + str(layer))(x)
            if layer == last and self.use_tanh:               # networks/model.py:314-315: tanh twice with use_tanh
                x = torch.tanh(x)
            if layer < last:
                if not self.weight_norm and layer in self.norm_layers:
                    x = getattr(self, "bn" + prefix[3] + str(layer))(x)
                x = torch.relu(x)
                if self.training and self.dropout is not None and layer in self.dropout:
                    x = torch.nn.functional.dropout(x, p=self.dropout_prob, training=True)
        return torch.tanh(x), scores

    def forward(self, inputs):
        xh, xo = self.head_inputs(inputs)
        hand, scores = self._head("linh", xh, self.use_classifier)
        obj, _ = self._head("lino", xo)
        return hand[:, 0:1], obj[:, 0:1], scores if self.use_classifier else torch.zeros(1, device=inputs.device)


class CombinedDecoder(nn.Module):
    """One MLP over [latent | point features] with a two-row last layer: column 0 = hand SDF, column 1 = object
    SDF (networks/model.py:79-188; state-dict keys `lin{0..4}.*`)."""

    def __init__(self, latent_size, point_feat_size, encode_style, dims, num_class=6, dropout=None, dropout_prob=0.0,
                 norm_layers=(), latent_in=(), weight_norm=False, xyz_in_all=None, use_tanh=False, latent_dropout=False,
                 use_classifier=False):
        super().__init__()
        self.latent_size, self.point_feat_size, self.encode_style = latent_size, point_feat_size, encode_style
        self.norm_layers, self.latent_in, self.weight_norm = tuple(norm_layers or ()), tuple(latent_in), weight_norm
        self.dropout, self.dropout_prob = dropout, dropout_prob
        self.use_classifier, self.use_tanh, self.xyz_in_all, self.latent_dropout = use_classifier, use_tanh, xyz_in_all, latent_dropout
        self.num_class = num_class
        sizes = [latent_size + point_feat_size] + list(dims) + [2]
        self.num_layers = len(sizes)
        for layer in range(len(sizes) - 1):
            if (layer + 1) in self.latent_in:
                n_out = sizes[layer + 1] - sizes[0]
            else:
                n_out = sizes[layer + 1]
                if xyz_in_all and layer != self.num_layers - 2:      # room for the xyz re-injection (networks/model.py:118-119)
                    n_out -= point_feat_size
            setattr(self, "lin" + str(layer), _linear(sizes[layer], n_out, weight_norm and layer in self.norm_layers))
            if not weight_norm and layer in self.norm_layers:        # LayerNorm form (networks/model.py:131-132)
                setattr(self, "bn" + str(layer), nn.LayerNorm(n_out))
        if use_classifier:      # networks/model.py:134-137
            self.classifier_head = nn.Linear(list(dims)[-1], num_class)

    def forward(self, inputs):
        x = inputs
        xyz = inputs[:, -self.point_feat_size:]
        scores = torch.zeros(1, device=inputs.device)
        for layer in range(self.num_layers - 1):
            if self.use_classifier and layer == self.num_layers - 2:
                scores = self.classifier_head(x)
            if layer in self.latent_in:
                x = torch.cat([x, inputs], 1)
            elif layer != 0 and self.xyz_in_all:                     # networks/model.py:166-167
                x = torch.cat([x, xyz], 1)
            x = getattr(self, "lin" + str(layer))(x)
            if layer == self.num_layers - 2 and self.use_tanh:
                x = torch.tanh(x)
            if layer < self.num_layers - 2:
                if not self.weight_norm and layer in self.norm_layers:
                    x = getattr(self, "bn" + str(layer))(x)
                x = torch.relu(x)
        x = torch.tanh(x)
        return x[:, 0:1], x[:, 1:2], scores


def build_decoder(specs, state_dict=None):
    """Decoder module from a specs.json dict (networks/model_utils.py:14-31): SeparateDecoder for ModelType
    1encoder2decoder, CombinedDecoder for 1encoder1decoder."""
    cls = {"1encoder2decoder": SeparateDecoder, "1encoder1decoder": CombinedDecoder}.get(specs.get("ModelType", "1encoder2decoder"))
    if cls is None:
        raise NotImplementedError("unsupported ModelType %r" % specs.get("ModelType"))
    dec = cls(specs["LatentSize"], specs["PointFeatSize"], specs["EncodeStyle"], **specs["NetworkSpecs"],
              use_classifier=bool(specs.get("ClassifierBranch", False)))
    if state_dict is not None:
        sd = {}
        for k, v in state_dict.items():
            for pre in ("module.decoder.", "decoder."):
                if k.startswith(pre):
                    k = k[len(pre):]
            if k.startswith(("lin", "bn", "classifier_head")):
                sd[k] = torch.as_tensor(v)
        dec.load_state_dict(sd)
    return dec.eval()
