"""Turn raw rocprofv3 output (gpurun_out/...) into the summaries committed under profiles/.

    python tools/make_profiles.py --tag r01 --steps 15 \
        --stats gpurun_out/prof/stats/x_kernel_stats.csv \
        --pmc-fetch gpurun_out/prof/fetch/x_counter_collection.csv \
        --pmc-write gpurun_out/prof/write/x_counter_collection.csv \
        --traced 'gather_gemm_v2_kernel<64, 32, false, 1, 0>'

Commands that produce the inputs (each in its own run; --pmc is never combined with anything but --kernel-trace):
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir>/stats -o x -- python bench.py --steps S --warmup W --no-cpu-baseline
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir>/fetch -o x -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir>/write -o x -- python bench.py ...
HBM bytes = FETCH_SIZE[KiB] x 1024 x 2 + WRITE_SIZE[KiB] x 1024 (gfx950 correction calibrated by tools/pmc_calibrate.py as
MI355X_MICROARCH.md's HBM section prescribes).
"""
import argparse
import csv
import json
import os
import re
import shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAMILIES = [
    ("gather_gemm (conv fwd + bwd-input)", r"gather_gemm"),
    ("bwd_weight (+reduce)", r"bwd_weight"),
    ("bn_*", r"vc::bn_"),
    ("rulebook / geometry plan (hash/subm/sp_*/image_*/parity/row_order/keep)", r"vc::(hash_|subm_|sp_|scan_|row_order|flag_|image_|parity_|keep_count|random_keep|gather_coords|set_count|bev_pairs)"),
    ("group_sum (seg_sum / seg_fixup / plan keys; r1-r2: fixed-point + absmax + convert)", r"vc::(group_|seg_|absmax)"),
    ("sort (rocPRIM radix sort of the duplicate-pixel group plans)", r"rocprim"),
    ("loss (vc_weighted_sum: the benchmark's stand-in loss, fused product + reduction)", r"weighted_sum"),
    ("optimizer (vc_clip_adamw: gradient norm + clip + AdamW over the flat parameters)", r"vc::(grad_sqsum|clip_adamw)"),
    ("other vc:: (project, gather/scatter rows, dense, voxelizer)", r"vc::"),
    ("memset/copy", r"__amd_rocclr"),
    ("torch (loss, optimizer, randperm, cat, ...)", r"."),
]


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    return name[-90:]


def stats_summary(path: str, steps: int, out_md: str, title: str) -> None:
    rows = list(csv.DictReader(open(path)))
    once = [int(r["Calls"]) for r in rows if "dense_kernel<true>" in r["Name"] or "dense_fill_kernel" in r["Name"]]  # once per step
    if once:
        steps = once[0]  # counts the settle / warm-up / timed steps alike, whatever their number was
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
    calls = sum(int(r["Calls"]) for r in rows) / steps
    fam = defaultdict(float)
    for r in rows:
        for label, pat in FAMILIES:
            if re.search(pat, r["Name"]):
                fam[label] += float(r["TotalDurationNs"]) / steps / 1e6
                break
    with open(out_md, "w") as f:
        f.write(f"# {title}\n\n{steps} steps profiled; total kernel time {tot:.3f} ms/step, {calls:.0f} launches/step "
                "(main, plan and weight-gradient streams overlap, so this exceeds the wall step time).\n\n")
        f.write("| family | ms/step |\n|---|---|\n")
        for label, _ in FAMILIES:
            if fam[label]:
                f.write(f"| {label} | {fam[label]:.3f} |\n")
        f.write("\n| % | ms/step | launches/step | avg us | kernel |\n|---|---|---|---|---|\n")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
            ms = float(r["TotalDurationNs"]) / steps / 1e6
            f.write(f"| {100 * ms / tot:.2f} | {ms:.3f} | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"`{short(r['Name'])}` |\n")


def pmc_table(path: str, counter: str):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--steps", type=int, required=True, help="steps + warmup of the profiled bench command")
    ap.add_argument("--stats")
    ap.add_argument("--pmc-fetch")
    ap.add_argument("--pmc-write")
    ap.add_argument("--traced", default="bwd_weight_kernel<32, 32;gather_gemm_v2_kernel<64, 32, false;bn_bwd_dx_pow2_kernel;gather_gemm_v2_kernel<8, 8, false",
                    help="';'-separated name fragments of the kernels whose HBM bytes per launch go into profiles/<tag>_traffic_*.json")
    ap.add_argument("--command", default="python bench.py --steps 10 --warmup 5 --no-cpu-baseline")
    args = ap.parse_args()
    pdir = os.path.join(ROOT, "profiles")
    os.makedirs(pdir, exist_ok=True)
    if args.stats:
        shutil.copy(args.stats, os.path.join(pdir, f"{args.tag}_bench_kernel_stats.csv"))
        stats_summary(args.stats, args.steps, os.path.join(pdir, f"{args.tag}_bench_summary.md"),
                      f"rocprofv3 --kernel-trace --stats of `{args.command}` (MI355X)")
    if args.pmc_fetch and args.pmc_write:
        fe, wr = pmc_table(args.pmc_fetch, "FETCH_SIZE"), pmc_table(args.pmc_write, "WRITE_SIZE")
        names = sorted(set(fe) | set(wr), key=lambda k: -(fe[k][1] * 2 + wr[k][1]))
        with open(os.path.join(pdir, f"{args.tag}_pmc_hbm_traffic.md"), "w") as f:
            f.write("# HBM traffic per launch from rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE x1, separate passes), "
                    f"`{args.command}`, average over launches\n\n| kernel | launches | fetch MB (corrected) | write MB |\n|---|---|---|---|\n")
            for k in names:
                if "vc::" not in k:
                    continue
                n = max(fe[k][0], wr[k][0], 1)
                f.write(f"| `{k}` | {n} | {fe[k][1] * 2 * 1024 / max(fe[k][0], 1) / 1e6:.1f} | "
                        f"{wr[k][1] * 1024 / max(wr[k][0], 1) / 1e6:.1f} |\n")
        for traced in args.traced.split(";"):
            key = [k for k in names if traced.strip() in k]
            if not key:
                continue
            k = key[0]
            fk, wk = fe[k][1] / max(fe[k][0], 1), wr[k][1] / max(wr[k][0], 1)
            m = re.search(r"bwd_weight_kernel<(\d+), (\d+)", k)
            if "bn_bwd_dx" in k:
                fname = f"{args.tag}_traffic_bn_bwd_dx.json"
            elif m:
                fname = f"{args.tag}_traffic_bwd_weight_{m.group(1)}_{m.group(2)}.json"
            else:
                m = re.search(r"<(\d+), (\d+), (false|true)", k)
                fname = f"{args.tag}_traffic_gather_gemm_{m.group(1)}_{m.group(2)}_{'bwd' if m.group(3) == 'true' else 'fwd'}.json"
            json.dump({"kernel": k, "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- {args.command} "
                                               "(two separate passes)",
                       "launches": fe[k][0], "fetch_kb_raw_avg": fk, "write_kb_raw_avg": wk,
                       "hbm_bytes_per_launch_corrected": fk * 1024 * 2 + wk * 1024,
                       "correction": "FETCH_SIZE x2, WRITE_SIZE x1: calibrated on known byte counts in our own access patterns "
                                     "(tools/pmc_calibrate.py), as MI355X_MICROARCH.md's HBM section prescribes"},
                      open(os.path.join(pdir, fname), "w"), indent=1)
            print("traffic ->", fname, fk * 1024 * 2 + wk * 1024)


if __name__ == "__main__":
    main()
