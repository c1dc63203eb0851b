"""Build libvirconv_hip.so (gfx950) in-tree with hipcc.  ``python -m virconv_amd.build [--force]``.

The shared object lands next to this file (``virconv_amd/libvirconv_hip.so``): it is git-ignored but travels to the
GPU box with the gpurun snapshot.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvirconv_hip.so")
ARCH = "gfx950"

# per-source extra flags: the index kernels must not contract mul+add (bit-exact integer outputs vs the oracle)
SOURCES = {
    "index_kernels.hip": ["-ffp-contract=off"],
    # MFMA accumulators in VGPRs (gfx950 has a unified register file): hipcc's default keeps them in AGPRs and shuttles them with
    # v_accvgpr_read/write around every branch (48 moves in the hot loop of gather_gemm_v2<64,32>) -- 68 VGPR + 12 AGPR -> 68 VGPR,
    # 6 -> 7 waves per SIMD, and these kernels' throughput follows their occupancy.  (Round 5 checked whether this flag is
    # an ingredient of LOG.md A.17 -- VIRCONV_NO_VGPR_FORM=1 builds without it: it is not, the unguarded loop fails the same way.)
    "conv_kernels.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
    "bn_kernels.hip": [],
    "pool_kernels.hip": ["-ffp-contract=off"],
    "frontend_kernels.hip": ["-ffp-contract=off"],
    # rotated IoU / NMS: no mul+add contraction, so that the polygon arithmetic rounds like the reference's CPU build
    "nms_kernels.hip": ["-ffp-contract=off"],
    "group_kernels.hip": [],
    "optim_kernels.hip": [],
    "unit.hip": [],
    "pass.hip": [],
    "plan.hip": [],
}
if os.environ.get("VIRCONV_NO_VGPR_FORM") == "1":   # A/B builds only (LOG.md A.17): MFMA accumulators in AGPRs
    SOURCES["conv_kernels.hip"] = []
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "virconv_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    # A/B builds (developer only): VIRCONV_LIB_OUT=<path> VIRCONV_HIPCC_EXTRA="-DVC_V2_SCHED=0" python -m virconv_amd.build
    # writes a second library that tools/kbench.py can load through VIRCONV_LIB=<path>
    out, extra_env = os.environ.get("VIRCONV_LIB_OUT"), os.environ.get("VIRCONV_HIPCC_EXTRA", "").split()
    if out:
        return _build_to(out, os.path.join(HERE, "build_ab"), extra_env, verbose)
    if not force and not _stale():
        return LIB
    return _build_to(LIB, os.path.join(HERE, "build"), extra_env, verbose, force_all=force)


def _build_to(LIB: str, objdir: str, extra_env, verbose: bool, force_all: bool = False) -> str:
    hipcc = _hipcc()
    os.makedirs(objdir, exist_ok=True)

    def compile_one(item):
        src, extra = item
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *COMMON, *extra, *extra_env, "-c", os.path.join(CSRC, src), "-o", obj]
        # per-object staleness: the object is reused when it is newer than its source and the shared headers and was built
        # with the same command line (stamp file next to it)
        stamp, line = obj + ".cmd", " ".join(cmd)
        deps = [os.path.join(CSRC, src), os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "virconv_hip.h")]
        if (not force_all and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == line
                and all(os.path.getmtime(d) < os.path.getmtime(obj) for d in deps)):
            return obj
        if verbose:
            print("[virconv_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(stamp, "w") as f:
            f.write(line)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES.items()))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", LIB + ".tmp"]
    if verbose:
        print("[virconv_amd.build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
