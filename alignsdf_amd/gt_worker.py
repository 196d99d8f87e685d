"""Ground-truth sampling worker process of eval mode (`python -m alignsdf_amd.gt_worker`, started by
reconstruct.GroundTruthPrefetcher).  The reference reads one ground-truth mesh per sample and samples 30 000 points from it
(utils/mesh.py:386-389, deep_sdf/metrics/icp_trans_scale.py:19-23); here that is numpy-only work in a process of its own, so that
15 ms of parsing per sample never holds the interpreter lock of the process that launches the decoder passes.

Protocol on stdin / stdout: length-prefixed pickles.  Request (path, samples, seed) -> reply None (no such file),
a [samples, 3] float64 array, or ("error", text).  An empty read ends the worker."""
import os
import pickle
import struct
import sys


def read_message(stream):
    head = stream.read(8)
    if len(head) < 8:
        return None
    (size,) = struct.unpack("<q", head)
    body = stream.read(size)
    if len(body) < size:
        return None
    return pickle.loads(body)


def write_message(stream, obj):
    body = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    stream.write(struct.pack("<q", len(body)))
    stream.write(body)
    stream.flush()


def load_samples(path, samples, seed):
    import numpy as np
    from .surface_sampling import load_obj, sample_surface
    if not os.path.exists(path):
        return None
    gv, gf = load_obj(path)
    return np.ascontiguousarray(sample_surface(gv, gf, samples, seed))


def main():
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    sys.stdout = sys.stderr                       # stray prints must not corrupt the reply stream
    while True:
        req = read_message(inp)
        if req is None:
            return
        try:
            reply = load_samples(*req)
        except Exception as e:                    # reported to the consumer, which raises it at get()
            reply = ("error", "%s: %s" % (type(e).__name__, e))
        write_message(out, reply)


if __name__ == "__main__":
    main()
