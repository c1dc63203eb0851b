"""TEST INFRASTRUCTURE (oracle): build oracle/_ref/libiou3d_ref.so from the reference's own source file
/root/reference/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp (compiled where it lies, never copied) + oracle/ref_iou3d_wrap.cpp.

    python -m oracle.build_ref

The reference file includes <cuda.h> / <cuda_runtime_api.h> and marks its helpers __device__ although the CPU path needs no
CUDA: two empty stand-in headers (oracle/ref_stubs/) and -D__device__= let g++ compile it unmodified.  It links against the
libtorch of this image (the reference's own setup.py builds it as a torch extension too).  The output is git-ignored and
travels to the GPU box with the snapshot; when /root/reference is absent the prebuilt file is used as is.
"""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp"
OUT = os.path.join(HERE, "_ref", "libiou3d_ref.so")


def build(force: bool = False, verbose: bool = True) -> str | None:
    if not os.path.exists(REF_SRC):
        return OUT if os.path.exists(OUT) else None
    wrap = os.path.join(HERE, "ref_iou3d_wrap.cpp")
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) > max(os.path.getmtime(wrap), os.path.getmtime(REF_SRC)):
        return OUT
    from torch.utils import cpp_extension as ce
    import torch
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    inc = [os.path.join(HERE, "ref_stubs"), os.path.dirname(REF_SRC), sysconfig.get_paths()["include"], *ce.include_paths()]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__device__=", "-w",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", *[f"-I{i}" for i in inc], REF_SRC, wrap,
           f"-L{libdir}", "-ltorch", "-ltorch_cpu", "-lc10", f"-Wl,-rpath,{libdir}", "-o", OUT + ".tmp"]
    if verbose:
        print("[oracle.build_ref]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed on the reference's iou3d_cpu.cpp:\n" + r.stderr[-4000:])
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
