"""Build libs2t_hip.so (all HIP kernels + the C-ABI) in-tree for gfx950 with hipcc.

The built library lives at ``s2t_amd/lib/libs2t_hip.so`` (git-ignored, shipped to the GPU box by gpurun).
Objects are cached by source mtime so that a rebuild after touching one kernel takes seconds.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libs2t_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# Per-file extras.  attention_fused.hip: hipcc puts the MFMA results of the fused attention kernels into AccVGPRs and copies
# every one of them out again for the softmax arithmetic (64 v_accvgpr moves per 16 x 64 step in kernels that are bound by
# vector issue); with the results in ordinary VGPRs the copies are gone and the kernels need FEWER registers (dQ 224 -> 208,
# the decoder's dK/dV 180 -> 168): dQ 44.1 -> 43.3 us, dK/dV 46.6 -> 44.4 us per launch on MI355X.
FILE_FLAGS = {"attention_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "s2t_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr = _deps_mtime()
    jobs = []
    objs = []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for done in ex.map(cc, jobs):
                if verbose:
                    print("[s2t_amd.build] compiled", os.path.basename(done), flush=True)
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print("[s2t_amd.build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
