"""Build libsis3d_hip.so (gfx950) in-tree: 3d-sis_amd/sis3d/libsis3d_hip.so.

hipcc cross-compiles without a GPU.  The integer-exact kernels (NMS, RoI pooling,
projection, proposal decode) are compiled with -ffp-contract=off so that no FMA
contraction changes the reference's binary32 operation sequence; the conv kernels
may contract.  No torch dependency: the library is a plain C-ABI shared object
(include/sis3d.h).
"""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "sis3d", "libsis3d_hip.so")
OBJ = os.path.join(HERE, "build")

EXACT = ["nms.hip", "roi_pool.hip", "projection.hip", "frustum.hip", "proposal.hip", "pool_misc.hip", "api.hip", "topk.hip"]
FAST = ["conv3d.hip", "conv3d_wino.hip", "conv3d_t16.hip", "bottleneck.hip", "pointwise.hip", "mlp.hip", "mlp16.hip", "enet.hip", "proj_sparse.hip"]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "mfma16.h"), os.path.join(ROOT, "include", "sis3d.h")]
    objs, jobs = [], []
    # slowest translation units first so the pool drains evenly
    for name in FAST + EXACT:
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJ, name.replace(".hip", ".o"))
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs):
            jobs.append(base + (["-ffp-contract=off"] if name in EXACT else []) + (["-fno-slp-vectorize"] if name == "conv3d_wino.hip" else []) + ["-c", src, "-o", obj])
        objs.append(obj)
    width = max(1, min(len(jobs), int(os.environ.get("SIS3D_BUILD_JOBS", os.cpu_count() or 1))))
    running = []
    while jobs or running:
        while jobs and len(running) < width:
            cmd = jobs.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))
        # reap whichever job ends first (waiting on the head of the list leaves the pool idle behind one slow translation unit)
        done = [i for i, (_, p) in enumerate(running) if p.poll() is not None]
        if not done:
            time.sleep(0.2)
            continue
        cmd, proc = running.pop(done[0])
        if proc.returncode != 0:
            for _, other in running:
                other.kill()
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    if force or any(_newer(o, OUT) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
