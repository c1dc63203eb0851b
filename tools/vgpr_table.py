#!/usr/bin/env python3
"""VGPR counts (and the wave occupancy they allow) of the gather-GEMM instantiations in libvirconv_hip.so.

    python tools/vgpr_table.py [--save FILE] [--against FILE]

Extracts the gfx950 code objects of the built library into a temporary directory (llvm-objdump --offloading), reads the kernel
descriptors' notes (llvm-readelf --notes) and prints, per `gather_gemm_v2_kernel<CK, CN, BWD, ..., EPI, NW, PK>`, the VGPR
count and private segment size.  --save writes the table as JSON; --against compares with a saved table and marks the
instantiations whose waves-per-SIMD changed (512 VGPRs per SIMD lane, allocation granule 8).
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def waves(v):
    return min(8, 512 // max(8, (v + 7) // 8 * 8))


def parse_notes(notes, out):
    for name, priv, vg in re.findall(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)",
                                     notes, flags=re.S):
        m = re.match(r"_ZN2vc21gather_gemm_v2_kernelILi(\d+)ELi(\d+)ELb([01])ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb([01])ELb([01])ELb([01])E",
                     name)
        if m:
            ck, cn, bwd, rt, ot, epi, nw, pk, dxs, il = (int(x) for x in m.groups())
            key = f"ck{ck} cn{cn} {'bwd' if bwd else 'fwd'} rt{rt} ot{ot} epi{epi} nw{nw} pk{pk} dxs{dxs} il{il}"
            out[key] = [int(vg), int(priv)]


def table(lib):
    out = {}
    if lib.endswith(".elf") or lib.endswith(".co"):   # a bare gfx950 code object
        parse_notes(subprocess.run([f"{LLVM}/llvm-readelf", "--notes", lib], capture_output=True, text=True).stdout, out)
        return out
    with tempfile.TemporaryDirectory() as d:
        local = os.path.join(d, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=d, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, check=False)
        for f in sorted(os.listdir(d)):
            if "gfx950" not in f:
                continue
            parse_notes(subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(d, f)], capture_output=True,
                                       text=True).stdout, out)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "virconv_amd", "libvirconv_hip.so"))
    ap.add_argument("--save")
    ap.add_argument("--against")
    ap.add_argument("--epi-only", action="store_true", help="only the STATS / BWD epilogue instantiations")
    args = ap.parse_args()
    t = table(args.lib)
    if args.save:
        with open(args.save, "w") as f:
            json.dump(t, f, indent=0, sort_keys=True)
    base = json.load(open(args.against)) if args.against else None
    for k in sorted(t):
        if args.epi_only and not (" epi1 " in k or " epi3 " in k):
            continue
        v, p = t[k]
        line = f"{k:58s} vgpr {v:3d} priv {p:4d} waves {waves(v)}"
        if base is not None and k in base:
            bv = base[k][0]
            line += f"   (was {bv:3d}, waves {waves(bv)})" + ("  ***" if waves(bv) != waves(v) else "")
        print(line)


if __name__ == "__main__":
    main()
