cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_icp.py tests/test_gpu_chamfer.py -q 2>&1 | tail -12
python - <<'PY'
import ctypes, time, numpy as np, torch
from alignsdf_amd import _native, icp
L = _native.lib()
rng = np.random.default_rng(1)
u = rng.normal(size=(30000, 3)); a = 0.35 * u / np.linalg.norm(u, axis=1, keepdims=True)
w = rng.normal(size=(30000, 3)); b = (0.35 * w / np.linalg.norm(w, axis=1, keepdims=True)) * 1.08 + np.array([0.03, -0.02, 0.015])
for mode, name in ((1, "brute force"), (2, "grid")):
    L.asdf_icp_set_search(mode)
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = icp.icp_trans_scale(a, b, a)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: ICP 30k x 30k, %d iterations, %.2f ms total, scale %.12f" % (name, out["iterations"], 1e3 * dt, out["scale"]))
L.asdf_icp_set_search(0)
PY
bash tools/r3_files.sh 2>&1 | grep -v "passed" | tail -24
python tools/time_frontend_overlap.py 256 8 2>/dev/null | tee gpurun_out/r3/frontend_overlap.txt
python -m pytest tests/test_module_path.py -q -m gpu 2>&1 | tail -3
