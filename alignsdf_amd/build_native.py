"""Build libalignsdf_hip.so (gfx950 only) in-tree with hipcc.

The shared library is the product: HIP kernels + the C ABI of include/alignsdf_hip.h.  It is
built next to its sources (alignsdf_amd/csrc/) so that it travels with the tree to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libalignsdf_hip.so")
SOURCES = ["decoder.hip", "k1_kernels.hip", "k1_cls_kernels.hip", "k1h_kernels.hip", "k1hw_kernels.hip", "k1h_nerf_kernels.hip", "k1s_kernels.hip", "k1s_nerf_kernels.hip", "mc33.hip", "icp.hip", "mesh_cc.hip", "surface_sample.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
# MFMA accumulators in VGPRs (the compiler's default heuristic puts them in AGPRs at this register pressure and the epilogues then read
# every accumulator back through v_accvgpr_read): measured per translation unit, same box, interleaved - the one-plane kernels -4.6 %
# time (1 862 -> 70 reads per tile body), the fp32 chain -0.4 % and no scratch left in its CombinedDecoder form; the split-half kernels
# (k1h_kernels.hip) +0.5 % and k1h_nerf_kernels.hip crashes this compiler in that form: both stay on the default.
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
# the W form's K-blocks carry more IR (sub-accumulator extracts / inserts around six MFMAs): at the default threshold the layer loops of
# k1hw_kernels.hip come out ROLLED, with indexed registers (s_set_gpr_idx_on) - 87 instead of 72 ms per sweep
# (-Wno-inline-asm: its LDS-DMA pieces name the reserved register M0 as clobbered - sdf_mlp_f16w_kernel.h: dma_piece_w - which the
# compiler reports once per instantiation)
UNROLL_ALL = ["-mllvm", "-pragma-unroll-threshold=200000", "-Wno-inline-asm"]
TU_FLAGS = {"k1hw_kernels.hip": UNROLL_ALL, "k1s_kernels.hip": VGPR_FORM, "k1s_nerf_kernels.hip": VGPR_FORM, "k1_kernels.hip": VGPR_FORM, "k1_cls_kernels.hip": VGPR_FORM}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP translation unit and link the shared library. Returns its path."""
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(HERE, "..", "include", "alignsdf_hip.h"))
    deps.append(os.path.abspath(__file__))          # (the per-unit flags live here)
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(o, deps):
            cmd = [HIPCC, *FLAGS, *TU_FLAGS.get(src, []), "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((src, subprocess.Popen(cmd, cwd=CSRC)))      # the translation units compile side by side
        objs.append(o)
    failed = [src for src, proc in jobs if proc.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc (%s)" % ", ".join(failed))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
