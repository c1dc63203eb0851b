"""bench.py -- VirConv-L backbone train step (fwd + bwd + Adam) on synthetic KITTI-shaped frames.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" = one pass of the hot path over one batch: BASELINE.json configs[2]
(VirConv-L forward + backward + Adam, bs=4 frames per GPU, train mode, layer discard 0.1, NRConv 2-D branch on); frames
shard across ranks with no data-path collective, the only exchange is the DDP gradient all-reduce (RCCL).  Inputs
(voxel features/coords, calib, aug params) are resident in HBM before the timed region; weak scaling (4 frames per GPU).

`--gpus N` without a torch.distributed.run environment re-launches itself under it (one rank per GPU, 127.0.0.1 rendezvous).
`--model 8x` = BASELINE configs[3]'s backbone (VirConv8x: LiDAR stream + virtual-point stream, bs 2 per GPU, 16 000 + 16 000
voxels per frame, layer discard 0.15); `--frontend` puts the per-frame data front-end (input point discard + LiDAR-first
voxeliser + MeanVFE from raw device-resident points, virtual points as the fp16 of the .npy files) inside the timed step.

Rank 0 prints ONE JSON line: metric/value/unit/..., plus
  "roofline":     the dominant kernel (gather-GEMM forward/backward-input family, fp32 MFMA) -- achieved algorithmic
                  TFLOP/s from HIP events bracketing every launch of the traced instantiation inside the timed steps
  "cpu_baseline": the CPU oracle (torch-CPU index_select/mm/index_add port of the reference algorithm class) on a
                  bounded sample (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Must be set before the HIP runtime initialises.  The step uses 3 streams (main, geometry plan, dW side stream); with an
# RCCL communicator alive and the default 4 hardware queues the plan stream's count reads stall behind other queues
# (10.8 vs 7.9 ms/step measured, 16.4 ms with 8 queues); 3 queues is best with and without RCCL.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from virconv_amd import data, ops, parallel, synth  # noqa: E402
from virconv_amd.backbone import VirConvL8x  # noqa: E402

MODEL_CFG = dict(NAME="VirConvL8x", NUM_FILTERS=[16, 32, 64, 64], RETURN_NUM_FEATURES_AS_DICT=True, OUT_FEATURES=64,
                 LAYER_DISCARD_RATE=0.1, LAYER_DISCARD_MODE="spconv1_inplace")
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
MFMA_16BIT_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16/f16 dense MFMA peak (16x16x32 / 32x32x16; the 16x16x16 forms run at half of it)
HBM_PEAK_GBS = 8000.0
L2_GATHER_CEILING_TBS = 9.9   # tools/ubench/gather_ubench.hip, 16-byte row gathers with the best lane mapping (profiles/r02_gather_ubench.txt)


def make_batch(frame_seeds, device, training=True):
    """Synthetic frames -> reference data path (input discard, LiDAR-first fusion) -> GPU voxeliser+MeanVFE -> batch."""
    frames, calibs, augs = [], [], []
    for s in frame_seeds:
        fr = synth.make_frame(s)
        rng = np.random.default_rng(10_000 + s)
        frames.append(data.prepare_frame(fr["points_lidar"], fr["points_virtual"], training=training, rng=rng))
        calibs.append(fr["calib"])
        augs.append(fr["aug_param"])
    feats, coords, _ = data.voxelize_batch(frames, synth.POINT_CLOUD_RANGE, synth.VOXEL_SIZE, 5, 40000, True, device)
    return {
        "batch_size": len(frame_seeds),
        "voxel_features": feats,
        "voxel_coords": coords.float(),  # load_data_to_gpu casts everything to float (models/__init__.py:24)
        "calib": ops.calib_tensor(calibs, device),
        "aug_param": torch.from_numpy(np.stack(augs)).to(device),
    }


def make_raw_frames(frame_seeds, device):
    """Raw per-frame points resident on the device, as the dataset hands them over BEFORE the front-end: LiDAR (Pl, 8) fp32
    and the depth-completed virtual points (Pv, 8) in the fp16 of their .npy files."""
    raw, calibs, augs = [], [], []
    for s in frame_seeds:
        fr = synth.make_frame(s)
        raw.append((torch.from_numpy(fr["points_lidar"]).to(device),
                    torch.from_numpy(fr["points_virtual"].astype(np.float16)).to(device)))
        calibs.append(fr["calib"])
        augs.append(fr["aug_param"])
    return raw, {"batch_size": len(frame_seeds), "calib": ops.calib_tensor(calibs, device),
                 "aug_param": torch.from_numpy(np.stack(augs)).to(device)}


def front_end(raw, base, training=True):
    """The data front-end on the GPU: one fused call per frame (vc_frontend_voxelize_mean), one count read per batch.  It
    runs on the backbone's high-priority geometry stream: its count read then waits for a few short kernels, not for the
    previous step's backward that is still queued on the main stream (the host keeps its run-ahead)."""
    from virconv_amd.backbone import _plan_stream
    dev = raw[0][0].device
    main, side = torch.cuda.current_stream(), _plan_stream(dev)
    ready = base.get("inputs_ready_event")
    if ready is not None:
        side.wait_event(ready)
    else:
        side.wait_stream(main)
    with torch.cuda.stream(side):
        feats, coords = data.frontend_batch(raw, training, synth.POINT_CLOUD_RANGE, synth.VOXEL_SIZE, 5, 40000, True)
        coords = coords.float()
    main.wait_stream(side)
    feats.record_stream(main)
    coords.record_stream(main)
    bd = dict(base)
    bd["voxel_features"], bd["voxel_coords"] = feats, coords
    return bd


MODEL_CFG_8X = dict(NAME="VirConv8x", NUM_FILTERS=[16, 32, 64, 64], RETURN_NUM_FEATURES_AS_DICT=True, OUT_FEATURES=64,
                    LAYER_DISCARD_RATE=0.15, LAYER_DISCARD_MODE="spconv1_inplace", MM=True)


def make_batch_8x(frame_seeds, device):
    """VirConv-T/S backbone input (VirConv-T.yaml:9,119-122): a LiDAR-only voxel set and a fused (MM) voxel set per frame,
    16 000 voxels each at most."""
    lidar, mm, calibs, augs = [], [], [], []
    for s in frame_seeds:
        fr = synth.make_frame(s)
        rng = np.random.default_rng(10_000 + s)
        lidar.append(fr["points_lidar"])
        mm.append(data.prepare_frame(fr["points_lidar"], fr["points_virtual"], training=True, rng=rng))
        calibs.append(fr["calib"])
        augs.append(fr["aug_param"])
    f, c, _ = data.voxelize_batch(lidar, synth.POINT_CLOUD_RANGE, synth.VOXEL_SIZE, 5, 16000, True, device)
    fm, cm, _ = data.voxelize_batch(mm, synth.POINT_CLOUD_RANGE, synth.VOXEL_SIZE, 5, 16000, True, device)
    return {"batch_size": len(frame_seeds), "voxel_features": f, "voxel_coords": c.float(), "voxel_features_mm": fm,
            "voxel_coords_mm": cm.float(), "calib": ops.calib_tensor(calibs, device),
            "aug_param": torch.from_numpy(np.stack(augs)).to(device)}


def make_batch_8x_eval(frame_seeds, device, rot_num=3):
    """VirConv-T/S at test time (spconv_backbone.py:414-432, VirConv-T.yaml:119-122): `rot_num` rotated copies of every frame (test-time
    augmentation), <= 40 000 LiDAR and <= 40 000 fused voxels per frame and copy; the backbone concatenates the copies along x into one
    [81, 1600, 5632] tensor for the LiDAR stream.  transform_param[b, i] = [rot, flip, scale] of copy i."""
    rots = [0.0, 0.3925, -0.3925][:rot_num]
    out = {"batch_size": len(frame_seeds)}
    calibs = [synth.make_frame(s)["calib"] for s in frame_seeds]
    for i, rot in enumerate(rots):
        rid = "" if i == 0 else str(i)
        c_, s_ = np.cos(rot), np.sin(rot)
        rm = np.array([[c_, s_], [-s_, c_]], np.float32)
        lidar, mm = [], []
        for s in frame_seeds:
            fr = synth.make_frame(s)
            rng = np.random.default_rng(10_000 + s)
            pl, pv = fr["points_lidar"].copy(), fr["points_virtual"].copy()
            pl[:, :2] = pl[:, :2] @ rm
            pv[:, :2] = pv[:, :2] @ rm
            lidar.append(pl)
            mm.append(data.prepare_frame(pl, pv, training=False, rng=rng))
        f, c, _ = data.voxelize_batch(lidar, synth.POINT_CLOUD_RANGE, synth.VOXEL_SIZE, 5, 40000, True, device)
        fm, cm, _ = data.voxelize_batch(mm, synth.POINT_CLOUD_RANGE, synth.VOXEL_SIZE, 5, 40000, True, device)
        out.update({"voxel_features" + rid: f, "voxel_coords" + rid: c.float(), "voxel_features_mm" + rid: fm,
                    "voxel_coords_mm" + rid: cm.float()})
    tp = np.zeros((len(frame_seeds), len(rots), 3), np.float32)
    tp[:, :, 0] = np.asarray(rots, np.float32)[None, :]
    tp[:, :, 2] = 1.0
    out["transform_param"] = torch.from_numpy(tp).to(device)
    out["calib"] = ops.calib_tensor(calibs, device)
    return out


def make_loss_weights(device):
    g = torch.Generator(device="cpu").manual_seed(1234)
    w = {"dense": torch.randn((1, 64, 4, 200, 176), generator=g).to(device) * 0.01}
    for name, c in (("x_conv1", 16), ("x_conv2", 32), ("x_conv3", 64), ("x_conv4", 64)):
        w[name] = torch.randn((c,), generator=g).to(device) * 0.01
    return w


def synthetic_loss(out, lw):
    """loss = (out.dense() * G).sum() + sum_i (x_conv_i.features * g_i).sum()   (the heads are out of scope).
    Every term is one fused, deterministic multiply-and-reduce pass (ops.weighted_sum -> vc_weighted_sum; torch's product + sum
    cost 0.24 ms of the 5.4 ms step); on the CPU oracle it is plain tensor ops."""
    loss = ops.weighted_sum(out["encoded_spconv_tensor"].dense(), lw["dense"])   # one fused pass (vc_weighted_sum) on the GPU
    for group in ("multi_scale_3d_features", "multi_scale_3d_features_mm"):
        for name, t in out.get(group, {}).items():
            loss = loss + ops.weighted_sum(t.features, lw[name])
    return loss


def train_step(model, optimizer, batch, lw, grad_sync=None, raw=None, next_batch=None):
    """fwd + bwd + Adam.  loss = (out.dense()*G).sum() + sum_i (x_conv_i.features * g_i).sum()  (heads out of scope).
    `raw`: run the data front-end on the raw points first (the --frontend workload); `batch` then only carries calib / aug.
    `next_batch`: the batch of the NEXT step, as a prefetching loader holds it: the first half of its geometry plan (coordinates,
    keeps, row counts: VirConvL8x.plan_ahead_begin) is enqueued before this step's forward; the next call finishes and uses it."""
    optimizer.zero_grad(set_to_none=True)
    if raw is not None:
        bd = front_end(raw, batch)
    else:
        bd = dict(batch)
        bd["voxel_features"] = batch["voxel_features"].clone()  # the backbone zeroes RGB in place
    base = getattr(model, "module", model)
    if next_batch is not None:
        base.plan_ahead_begin(next_batch)
    out = model(bd)
    loss = synthetic_loss(out, lw)
    loss.backward()
    if grad_sync is not None:
        grad_sync()  # data-parallel exchange: one flat RCCL all-reduce of the gradients
    # train_utils.py:50.  The parameter list is taken once per model (walking the module tree for it costs 0.15 ms of host time per step)
    params = getattr(base, "_bench_param_list", None)
    if params is None:
        from virconv_amd import feature_pass
        params = base._bench_param_list = feature_pass.trainable_parameters(base)   # the flat parameters if the model was flattened
    if getattr(optimizer, "clips", False):
        optimizer.step()                                  # virconv_amd.optim.ClipAdamW: the clip is part of the step (vc_clip_adamw)
    else:
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        optimizer.step()
    return loss


def cpu_baseline(sample_frames=1, repeats=1):
    """Time the CPU oracle (kind 'port') on a bounded sample of the same workload: `sample_frames` frame(s), fwd+bwd+Adam."""
    from oracle.backend import OracleBackend
    cores = os.cpu_count() or 1
    threads = min(cores, 16)  # more threads than this slow the small gather/mm/scatter ops down
    torch.set_num_threads(threads)
    be = OracleBackend()
    with ops.use_backend(be):
        # inputs through the oracle voxeliser (CPU), same synthetic frames
        batch = make_batch(list(range(sample_frames)), "cpu", training=True)
        model = VirConvL8x(MODEL_CFG, input_channels=8, grid_size=synth.GRID_SIZE)
        model.train()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01)
        lw = make_loss_weights("cpu")
        for _ in range(2):
            train_step(model, opt, batch, lw)  # warm-ups (allocator, thread pools)
        repeats = max(5, repeats)              # median of >= 5; ~7 s of CPU work in total, so that the GPU phase is most of the run
        times = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            train_step(model, opt, batch, lw)
            times.append(time.perf_counter() - t0)
        dt = sorted(times)[len(times) // 2]
    return {"value": round(sample_frames / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{sample_frames} synthetic KITTI frame(s), VirConv-L fwd+bwd+Adam, oracle (torch-CPU gather-mm-scatter), "
                      f"median of {repeats} timed step(s) ({dt:.2f} s) after 2 warm-ups"}


def run_infer(args, model, batch, device, rank, world):
    """BASELINE configs[1]: VirConv-L forward only, eval mode (no discard), `--batch-size` frames per GPU (1 in the config)."""
    model.eval()

    # A serving loop holds the NEXT frame while it runs this one: the first half of that frame's geometry plan (coordinates, keeps, row
    # counts: VirConvL8x.plan_ahead_begin, no host synchronisation) is enqueued before this frame's forward, so its one count read has
    # long arrived when the next call asks for it.  VIRCONV_INFER_PLAN_AHEAD=0: plan in place (rounds 1-5).
    ahead = os.environ.get("VIRCONV_INFER_PLAN_AHEAD", "1") != "0" and hasattr(model, "plan_ahead_begin")

    def step():
        bd = dict(batch)
        for k in batch:
            if k.startswith("voxel_features"):
                bd[k] = batch[k].clone()      # the backbone zeroes RGB in place
        with torch.no_grad():
            if ahead:
                model.plan_ahead_begin(batch)     # the next frame (the same synthetic frame again)
            out = model(bd)
            return out["encoded_spconv_tensor"].dense()

    for _ in range(args.warmup):
        step()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, device)
    # roofline of the same gather-GEMM instantiation as the train line (here with the eval-BatchNorm epilogue): 10 more frames
    # AFTER the timed region -- at 1.2-1.8 ms per step the event records and pair counts of a trace are a visible share of a
    # forward-only step (measured 1.90 vs 1.76 ms at bs 4), which they are not of the 5.4 ms train step
    be = ops.get_backend()
    tdir, tck, tcn = args.trace.split(",")
    if tdir == "dw":      # the train default; a forward-only loop has no weight gradient: the round 1-4 traced kernel
        tdir, tck, tcn = "fwd", "64", "32"
    be.trace_begin(tdir, int(tck), int(tcn))
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    trace = be.trace_end()
    # the loop above keeps two frames in flight (the geometry plan of frame f + 1 over the feature pass of frame f): throughput.
    # Latency of ONE frame with an idle GPU in front of it, for the record (median of 10, outside the timed region):
    lat = []
    for _ in range(10):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t1) * 1e3)
    lat_ms = sorted(lat)[len(lat) // 2]
    if rank == 0:
        bs = args.batch_size
        is8 = args.model == "8x"
        return {"metric": ("KITTI frames/sec (forward only) VirConv8x backbone (VirConv-T/S), test-time rotations" if is8 else
                           "KITTI frames/sec (forward only) VirConv-L backbone"), "value": round(bs * world * args.steps / dt, 3),
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": _dtype(ops.get_backend(), args.operand), "data": "synthetic",
                          "config": {"workload": ("BASELINE configs[3] eval path: VirConv8x forward only, eval mode, rot_num = 3 test-time "
                                                  "rotations x-concatenated into one [81, 1600, 5632] tensor (spconv_backbone.py:414-432), "
                                                  "<= 40000 + 40000 voxels per frame and rotation" if args.model == "8x" else
                                                  "BASELINE configs[1]: VirConv-L forward only, eval mode, + dense(); two frames in "
                                                  "flight (plan of the next frame over the feature pass of this one)"),
                                     "frames_per_gpu": bs, "voxels_rank0": int(batch["voxel_features"].shape[0]),
                                     "single_step_latency_ms": round(lat_ms, 3)},
                          "roofline": _traced_roofline(trace, args, tdir, tck, tcn, pmc=False), "cpu_baseline": None}
    return None


def _f32_split(be) -> int:
    import ctypes
    v = ctypes.c_int64(0)
    if not hasattr(be, "lib") or be.lib.vc_debug_get(b"f32_split", ctypes.byref(v)) != 0:
        return 0
    return int(v.value)


def _dtype(be, operand: str) -> str:
    """The arithmetic type of the path.  f32 tensors and f32 accumulation always; the conv products are either exact fp32 products on
    v_mfma_f32_16x16x4_f32 or -- the library default since round 4 -- six exact bf16 x bf16 cross terms of operands cut exactly into
    three bf16 pieces (within 2^-24 of the exact product, half an fp32 ulp: oracle/split_ref.py; tests/test_split_gpu.py holds the kernels
    against float64)."""
    if operand != "f32":
        return f"{operand} MFMA operands, f32 accumulate, f32 tensors"
    if _f32_split(be):
        return "f32 (f32 tensors and accumulation; conv products as a 6-term exact bf16 split on the MFMA, within 2^-24 of the exact product)"
    return "f32"


def _emit(res):
    """The ONE JSON line of the contract, as the last thing on stdout: whatever native libraries left in C stdio's buffer (RCCL's
    version banner on a multi-rank run) is flushed first."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(res), flush=True)


def _kernel_name(tdir, tck, tcn, windowed=False):
    if tdir == "dw":
        return f"bwd_weight_kernel<CI={tck},CO={tcn}>"
    if windowed:
        return f"gather_gemm_v3_kernel<CK={tck},CN={tcn},BWD={'true' if tdir == 'bwd' else 'false'}> (LDS row windows)"
    return f"gather_gemm_v2_kernel<CK={tck},CN={tcn},BWD={'true' if tdir == 'bwd' else 'false'},RT=1>"


def _pmc_traffic_of(stem):
    """(bytes per launch, source) of the newest committed PMC profile `profiles/<tag>_traffic_<stem>.json`, or (None, None)."""
    for tag in ("r06", "r05", "r04", "r03", "r02"):
        rel = os.path.join("profiles", f"{tag}_traffic_{stem}.json")
        try:
            with open(os.path.join(ROOT, rel)) as f:
                return float(json.load(f)["hbm_bytes_per_launch_corrected"]), rel + " (rocprofv3 --pmc passes of this command, not this run)"
        except Exception:
            continue
    return None, None


def _hbm_line(trace, kernel, stem):
    """The HBM side of the roofline for one bandwidth-bound kernel: algorithmic bytes of its launches / their HIP-event durations."""
    if not trace:
        return None
    t_ms = sum(e["ms"] for e in trace)
    byts = sum(e["bytes"] for e in trace)
    traffic, src = _pmc_traffic_of(stem)
    ach = byts / (t_ms * 1e-3) / 1e12
    return {"kernel": kernel, "launches": len(trace), "algorithmic_mb_per_launch": round(byts / len(trace) / 1e6, 3),
            "avg_us": round(t_ms / len(trace) * 1e3, 2), "achieved_tb_s": round(ach, 3), "frac_of_8tb_s": round(ach * 1e3 / HBM_PEAK_GBS, 4),
            "traffic_mb": None if traffic is None else round(traffic / 1e6, 2), "traffic_source": src}


def _pmc_traffic(tdir, tck, tcn):
    """HBM bytes per launch of the traced kernel from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in separate
    runs of this same command and corrected as MI355X_MICROARCH.md prescribes; summary committed under profiles/).
    PMC counters cannot be read from inside the process: the number is the one of the newest committed profile of this command,
    and the JSON line says so (`traffic_source`); (None, None) if absent."""
    stem = f"bwd_weight_{tck}_{tcn}" if tdir == "dw" else f"gather_gemm_{tck}_{tcn}_{tdir}"
    return _pmc_traffic_of(stem)


def _traced_roofline(trace, args, tdir, tck, tcn, pmc):
    """`roofline` object of the traced gather-GEMM instantiation: algorithmic flops of its launches / their HIP-event durations."""
    n_launch = len(trace)
    if not n_launch:
        return None
    t_ms = sum(e["ms"] for e in trace)
    flops = sum(e["flops"] for e in trace)
    byts = sum(e["bytes"] for e in trace)
    ach = flops / (t_ms * 1e-3) / 1e12
    peak = MFMA_F32_PEAK_TFLOPS if args.operand == "f32" else MFMA_16BIT_PEAK_TFLOPS
    traffic, traffic_src = _pmc_traffic(tdir, tck, tcn) if (pmc and args.operand == "f32") else (None, None)
    # what the conv kernels really run against (profiles/r06_dw_wide.md): every active pair gathers one source row (the
    # weight gradient: two) from L2 into a CU; the row-gather micro-benchmark tops out at 8.7-9.9 TB/s on this chip (profiles/r02_gather_ubench.txt)
    gathered = sum(e.get("pairs", 0) * 4.0 * ((e.get("ck", 0) + e.get("cn", 0)) if e.get("dir") == "dw" else e.get("ck", 0)) for e in trace)
    gtb = gathered / (t_ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
            "l2_gather_tb_s": round(gtb, 3), "l2_gather_frac_of_9p9_tb_s": round(gtb / L2_GATHER_CEILING_TBS, 4),
            "kernel": _kernel_name(tdir, tck, tcn, all(e["windowed"] for e in trace)),
            "launches": n_launch, "avg_us": round(t_ms / n_launch * 1e3, 2),
            "algorithmic_gflop_per_launch": round(flops / n_launch / 1e9, 4),
            "algorithmic_mb_per_launch": round(byts / n_launch / 1e6, 3),
            "hbm_frac_of_algorithmic_bytes": round(byts / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def main(argv=None, plumbing=False):
    """`plumbing` (tests/test_bench_gloo.py only): run the same code path on the CPU with whatever operator backend the caller has
    installed (the oracle) over `gloo` -- sharding, barrier, max-over-ranks timing, shutdown and the one JSON line are exercised
    without a GPU; nothing about it is a measurement."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--batch-size", type=int, default=4, help="frames per GPU (BASELINE config 3: bs=4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="train = BASELINE configs[2] (default, the headline metric); infer = configs[1]: forward only, eval mode")
    ap.add_argument("--operand", default="f32", choices=["f32", "f16", "bf16"],
                    help="MFMA operand type of the conv kernels.  f32 (default) is the headline / parity configuration; f16 | "
                         "bf16 = BASELINE configs[4] 'fp16 MFMA contraction' experiment (fp32 tensors, fp32 accumulate), "
                         "reported with its own dtype and a 16-bit MFMA peak, never as the headline number")
    ap.add_argument("--trace", default="dw,32,32",
                    help="conv kernel instantiation bracketed by HIP events INSIDE the timed steps for the roofline: dir,CK,CN with dir = "
                         "fwd | bwd (gather-GEMM) | dw (weight gradient).  Default: the kernel with the largest share of the step's conv "
                         "kernel time (checked against the per-shape table of the family steps: `roofline.kernel_is_dominant`)")
    ap.add_argument("--model", default="L", choices=["L", "8x"],
                    help="L = VirConvL8x, BASELINE configs[2] (default, the headline); 8x = VirConv8x (LiDAR + virtual-point "
                         "streams), the backbone of BASELINE configs[3], bs 2 per GPU unless --batch-size is given")
    ap.add_argument("--family-steps", type=int, default=3,
                    help="extra untimed steps after the timed region with every conv launch event-bracketed (family / step roofline)")
    ap.add_argument("--exact-steps", type=int, default=1, choices=[0, 1],
                    help="1: after the timed region, time the same K steps with exact-fp32 MFMA products (vc_debug_set f32_split=0; "
                         "reported as `exact_f32_mfma`)")
    ap.add_argument("--frontend", action="store_true",
                    help="include the GPU data front-end (input point discard + LiDAR-first voxeliser + MeanVFE from raw "
                         "device-resident points) in every timed step (model L)")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank, local_rank, world = parallel.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch N ranks, or run without a launcher)"
    if plumbing:
        device, numa = torch.device("cpu"), "plumbing test (cpu)"
        sync = lambda: None   # noqa: E731
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
        device = torch.device("cuda", local_rank % torch.cuda.device_count())
        torch.cuda.set_device(device)
        numa = parallel.bind_to_gpu_numa(device.index)  # one process per GPU, on the cores next to that GPU
        sync = torch.cuda.synchronize
        # The loop runs on a HIGH-priority stream: HIP has two levels (high, normal) and the library's side streams -- above all the
        # weight gradients, which overlap the backward sweep -- are meant to fill what the main stream leaves free, not to compete with
        # it for dispatch (4.22 vs 4.30 ms per step, inference bs 1 0.80 vs 0.83; LOG.md A.20).  VIRCONV_MAIN_PRIORITY=default: the
        # default stream (A/B); INTEGRATION.md section 10 says the same to a training loop.
        main_prio = os.environ.get("VIRCONV_MAIN_PRIORITY", "-1")
        if main_prio != "default":
            torch.cuda.set_stream(torch.cuda.Stream(device=device, priority=int(main_prio)))
    be = ops.get_backend()
    ops.MFMA_OPERAND = args.operand
    for kv_ in filter(None, os.environ.get("VIRCONV_DEBUG_SET", "").split(",")):   # A/B switches: "key=value,key=value" -> vc_debug_set
        key, val = kv_.split("=")
        assert be.lib.vc_debug_set(key.encode(), int(val)) == 0, kv_
    can_trace = hasattr(be, "trace_begin")

    if args.model == "8x" and "--batch-size" not in " ".join(sys.argv):
        args.batch_size = 2                                        # VirConv-T.yaml: bs 2 per GPU
    assert not (args.frontend and (args.model != "L" or args.mode != "train")), "--frontend is a train-mode VirConv-L workload"
    bs = args.batch_size
    seeds = parallel.shard_frames(list(range(bs * world)), rank, world)
    raw = None
    torch.manual_seed(0)
    if args.model == "8x":
        from virconv_amd.backbone import VirConv8x
        batch = make_batch_8x_eval(seeds, device) if args.mode == "infer" else make_batch_8x(seeds, device)
        model = VirConv8x(MODEL_CFG_8X, input_channels=8, grid_size=synth.GRID_SIZE).to(device)
    else:
        batch = make_batch(seeds, device, training=True)
        if args.frontend:
            raw, base = make_raw_frames(seeds, device)
            n_vox = int(batch["voxel_features"].shape[0])
            batch = base
        model = VirConvL8x(MODEL_CFG, input_channels=8, grid_size=synth.GRID_SIZE).to(device)
    model.train()
    use_torch_ddp = os.environ.get("VIRCONV_TORCH_DDP") == "1"   # stock DistributedDataParallel instead (slower here)
    # One flat parameter tensor per native pass (feature_pass.flatten_parameters): the stock clip and the stock fused AdamW then walk one or
    # two tensors instead of 60 / 180, and the backward hands ONE gradient to autograd -- host time, not device time.  VIRCONV_FLAT_PARAMS=0:
    # the per-parameter form (what stock DistributedDataParallel needs).
    flat_on = os.environ.get("VIRCONV_FLAT_PARAMS", "1") != "0" and not use_torch_ddp and not plumbing
    from virconv_amd import feature_pass
    opt_params = feature_pass.flatten_parameters(model) if flat_on else list(model.parameters())
    ddp = parallel.wrap_ddp(model, device) if use_torch_ddp else model
    grad_sync = None if use_torch_ddp else parallel.FlatGradAllReduce(model, opt_params)
    # a16 (train_utils.py:50-51): clip_grad_norm_(10) + the Adam step with true weight decay.  On flat parameters these are the two launches of
    # virconv_amd.optim.ClipAdamW (vc_clip_adamw); VIRCONV_FUSED_OPT=0, per-module parameters and the CPU plumbing run keep the stock pair
    # (clip_grad_norm_ + torch's fused multi-tensor AdamW: 12 launches, 0.10 ms of device time on one flat tensor).
    from virconv_amd import optim as vc_optim
    fused_opt = flat_on and os.environ.get("VIRCONV_FUSED_OPT", "1") != "0" and vc_optim.supports(opt_params)
    if fused_opt:
        optimizer = vc_optim.ClipAdamW(opt_params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, max_norm=10.0)
    else:
        optimizer = torch.optim.AdamW(opt_params, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, fused=not plumbing)
    lw = make_loss_weights(device)
    torch.manual_seed(100 + rank)  # layer-discard permutations
    # the inputs are resident in HBM from here on: lets the backbone's geometry plan run ahead on its side stream
    sync()
    if not plumbing:
        batch["inputs_ready_event"] = torch.cuda.Event()
        batch["inputs_ready_event"].record()

        # Setup (untimed, not a step): park a few GB of blocks in torch's caching allocator.  Layer discard is random, so
        # tensor sizes differ from step to step and the first steps would otherwise pay hipMalloc for every new size.
        prime = [torch.empty((1 << 30,), dtype=torch.uint8, device=device) for _ in range(8)]
        del prime

    # Setup (untimed): take Python's cyclic garbage collector out of the loop.  `import torch` leaves ~10^6 long-lived container
    # objects behind; a full (generation-2) collection walks all of them (tens of ms), and WHEN the per-frame garbage triggers
    # one depends on the ratio of young to old objects -- three more ctypes function objects at import time moved the bs-1
    # inference loop from 1.35 to 2.8-5.2 ms per frame (bisected, VIRCONV_GC_FREEZE=0 reproduces it).  gc.freeze() moves
    # everything alive now into the permanent generation: later collections only look at what the loop itself allocates.
    # Any serving / training loop around this backbone should do the same.
    if os.environ.get("VIRCONV_GC_FREEZE", "1") != "0":
        import gc
        gc.collect()
        gc.freeze()

    if args.mode == "infer":
        return run_infer(args, model, batch, device, rank, world)

    # Setup (untimed, not a step): a second of steady-state steps before the W warm-up steps (allocator, clocks).  Measured:
    # this does NOT remove the first-process-on-a-fresh-box penalty (7.5 ms vs 6.7-7.0 ms for later processes on the same
    # box, with or without 4 s of settling), whose cause is outside this process.
    settle = 0.0 if plumbing else float(os.environ.get("VIRCONV_SETTLE_SEC", "1.0"))
    if world > 1:
        # every rank must enter the gradient collective the same number of times: a wall-clock loop per rank does not (found by the
        # two-ranks-on-one-GPU test of round 6: one rank ran one step more and waited for ever) -- a fixed count instead
        for _ in range(int(settle / 0.005)):
            train_step(ddp, optimizer, batch, lw, grad_sync, raw)
            sync()
    else:
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < settle:
            train_step(ddp, optimizer, batch, lw, grad_sync, raw)
            sync()

    for _ in range(args.warmup):
        train_step(ddp, optimizer, batch, lw, grad_sync, raw)

    for env, key in (("VIRCONV_PASS_DW_MAIN_TAIL", b"pass_dw_main_tail"), ("VIRCONV_PASS_BWD_EPILOGUE", b"pass_bwd_epilogue")):
        if os.environ.get(env):   # A/B switches of the feature pass, see virconv_amd/csrc/pass.hip
            assert be.lib.vc_debug_set(key, int(os.environ[env])) == 0
    tdir, tck, tcn = args.trace.split(",")
    if can_trace:
        be.trace_begin(tdir, int(tck), int(tcn))
    parallel.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        train_step(ddp, optimizer, batch, lw, grad_sync, raw)
    sync()
    parallel.barrier()
    dt = time.perf_counter() - t0
    trace = be.trace_end() if can_trace else []
    # every rank's own time and shard size, on rank 0 (load imbalance between ranks is visible in the line, not only its maximum)
    per_rank = parallel.gather_to_rank0({"rank": rank, "ms_per_step": round(dt / args.steps * 1e3, 3),
                                         "voxels": int(batch["voxel_features"].shape[0]) if "voxel_features" in batch else None,
                                         "frames": list(seeds)})
    dt = parallel.max_over_ranks(dt, device)

    # Second timed loop (reported beside the headline): the same K steps with the conv products on v_mfma_f32_16x16x4_f32 (exact fp32
    # products) instead of the six-term bf16 split that is the library default (csrc/conv_kernels.hip, split3: operands cut EXACTLY into
    # three bf16 pieces, six cross terms on v_mfma_f32_16x16x32_bf16, fp32 accumulation; within 2^-24 of the exact product, measured
    # against float64 in tests/test_split_gpu.py).  Tensors, accumulation and every other kernel are the same in both.
    exact = None
    if args.exact_steps and args.operand == "f32" and not plumbing and args.mode == "train":
        import ctypes
        cur, cur_w = ctypes.c_int64(0), ctypes.c_int64(0)
        assert be.lib.vc_debug_get(b"f32_split", ctypes.byref(cur)) == 0 and be.lib.vc_debug_get(b"bw_split", ctypes.byref(cur_w)) == 0
        if cur.value != 0 or cur_w.value != 0:
            assert be.lib.vc_debug_set(b"f32_split", 0) == 0 and be.lib.vc_debug_set(b"bw_split", 0) == 0
            for _ in range(args.warmup):
                train_step(ddp, optimizer, batch, lw, grad_sync, raw)
            parallel.barrier()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                train_step(ddp, optimizer, batch, lw, grad_sync, raw)
            sync()
            parallel.barrier()
            dt_e = parallel.max_over_ranks(time.perf_counter() - t0, device)
            assert be.lib.vc_debug_set(b"f32_split", int(cur.value)) == 0 and be.lib.vc_debug_set(b"bw_split", int(cur_w.value)) == 0
            exact = {"ms_per_step": round(dt_e / args.steps * 1e3, 3), "value": round(bs * world * args.steps / dt_e, 3), "unit": "frames/s",
                     "note": "same K steps with vc_debug_set f32_split = 0, bw_split = 0: every conv product on v_mfma_f32_16x16x4_f32 (the round 1-3 kernels)"}

    # Family- and step-level roofline (outside the timed region, rank 0): a few more steps with EVERY gather-GEMM and
    # weight-gradient launch bracketed by HIP events on its launch stream (vc_trace_begin direction -1)
    fam = None
    if args.family_steps > 0 and can_trace:          # every rank steps (the gradient all-reduce is collective); rank 0 records
        if rank == 0:
            be.trace_begin("all", 0, 0, max_records=256 * args.family_steps)
        for _ in range(args.family_steps):
            train_step(ddp, optimizer, batch, lw, grad_sync, raw)
        sync()
        if rank == 0:
            fam = be.trace_end()
    # The traced kernel also FINISHES the BatchNorm statistics of its output since round 3 (arrival tickets in its last blocks: a
    # 3-9 us tail that replaces two launches).  For a like-for-like number of the GEMM itself: the same kernel over a few more
    # steps with the sums left to the BatchNorm kernels (vc_debug_set conv_bn_finish = 0), outside the timed region.
    plain = None
    if args.family_steps > 0 and args.operand == "f32" and can_trace and tdir != "dw":
        assert be.lib.vc_debug_set(b"conv_bn_finish", 0) == 0
        if rank == 0:
            be.trace_begin(tdir, int(tck), int(tcn))
        for _ in range(args.family_steps):
            train_step(ddp, optimizer, batch, lw, grad_sync, raw)
        sync()
        if rank == 0:
            plain = be.trace_end()
        assert be.lib.vc_debug_set(b"conv_bn_finish", 1) == 0
    # The HBM side (BASELINE's metric names it: "HBM GB/s vs peak"): the largest bandwidth-bound kernel of the step by time -- the
    # BatchNorm backward dx pass, 3 x 4 x N x C bytes per launch -- and the stage-1 conv (8 -> 8 channels: its pair table is the
    # largest object it touches), each bracketed by HIP events inside a few more steps, outside the timed region.
    hbm_bn = hbm_conv = None
    if args.family_steps > 0 and can_trace and not plumbing:
        for name in ("bn", "conv"):
            if rank == 0:
                if name == "bn":
                    be.trace_begin("bn_dx", 0, 0, max_records=64 * args.family_steps)
                else:
                    be.trace_begin("fwd", 8, 8, max_records=64 * args.family_steps)
            for _ in range(args.family_steps):
                train_step(ddp, optimizer, batch, lw, grad_sync, raw)
            sync()
            if rank == 0:
                t_ = be.trace_end()
                if name == "bn":
                    hbm_bn = t_
                else:
                    hbm_conv = t_
    parallel.barrier()

    # Host side of a step, for the record next to ms_per_step (which side bounds the step?): K steps ENQUEUED from an idle,
    # synchronised GPU without any synchronisation, wall time / K.  The host may run up to two forwards ahead of the GPU
    # (backbone._bound_run_ahead), so over 5 steps it is never throttled; the one count read per step (vc_plan_wait) is part of it.
    host_ms = None
    if not plumbing:
        sync()
        k_host = 5
        t0 = time.perf_counter()
        for _ in range(k_host):
            train_step(ddp, optimizer, batch, lw, grad_sync, raw)
        host_ms = (time.perf_counter() - t0) / k_host * 1e3
        sync()
    parallel.barrier()

    if rank != 0:
        return None
    frames = bs * world * args.steps
    split_on = bool(_f32_split(be)) and args.operand == "f32"
    issue_peak = MFMA_16BIT_PEAK_TFLOPS / 6.0   # six bf16 x bf16 terms per fp32 product on v_mfma_f32_16x16x32_bf16
    traced = _traced_roofline(trace, args, tdir, tck, tcn, pmc=True)
    roof = None
    if traced is not None:
        peak = traced["peak"]
        uses_split = split_on and int(tck) >= 16 and int(tcn) >= 16     # C < 16 layers stay on v_mfma_f32_16x16x4_f32
        # FLAT keys first (the driver's record keeps the leading scalar keys of this object): the kernel measured inside the timed
        # steps, both fractions, the step-level fraction; tables and notes behind them
        roof = {"bound": "mfma", "achieved": traced["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": traced["frac"],
                "traffic": traced["traffic"], "kernel": traced["kernel"],
                "l2_gather_tb_s": traced["l2_gather_tb_s"], "l2_gather_frac_of_9p9_tb_s": traced["l2_gather_frac_of_9p9_tb_s"],
                "frac_issue_pipe": round(traced["achieved"] / issue_peak, 4) if uses_split else traced["frac"],
                "issue_pipe_peak": round(issue_peak, 1) if uses_split else peak,
                "step_frac": None, "family_frac": None, "kernel_is_dominant": None, "kernel_ms_per_step": round(
                    sum(e["ms"] for e in trace) / args.steps, 3), "launches_per_step": round(len(trace) / args.steps, 1),
                "avg_us": traced["avg_us"], "algorithmic_gflop_per_launch": traced["algorithmic_gflop_per_launch"],
                "algorithmic_mb_per_launch": traced["algorithmic_mb_per_launch"], "traffic_source": traced["traffic_source"],
                "measured": "HIP events around every launch of this kernel inside the K timed steps, on its launch stream",
                "frac_note": ("frac = algorithmic fp32 flops / time / 157.3 TF (v_mfma_f32_16x16x4_f32, the path's arithmetic type); "
                              "frac_issue_pipe = the same / (2500 / 6) TF: this kernel issues each fp32 product as six "
                              "v_mfma_f32_16x16x32_bf16 terms" if uses_split else
                              "frac = algorithmic flops / time / the dense MFMA peak of the operand type")}
        if fam:
            # family: all conv kernels of the step (forward, backward-input, weight gradient) -- algorithmic flops / kernel time;
            # step: the same flops over the WALL time of a step (everything else -- BatchNorm, rulebooks, optimizer -- counts as loss)
            k = args.family_steps
            f_flops = sum(e["flops"] for e in fam)
            f_ms = sum(e["ms"] for e in fam)
            per_dir = {}
            for e in fam:
                d = per_dir.setdefault(e["dir"], [0.0, 0.0])
                d[0] += e["flops"]; d[1] += e["ms"]
            # wall time during which at least one conv kernel was running (the weight gradients overlap the backward-input convs
            # on a second stream: their bracketed durations inflate each other, the union does not double-count)
            iv = sorted((e["t0_ms"], e["t0_ms"] + e["ms"]) for e in fam)
            union, cur_a, cur_b = 0.0, iv[0][0], iv[0][1]
            for a_, b_ in iv[1:]:
                if a_ > cur_b:
                    union += cur_b - cur_a
                    cur_a, cur_b = a_, b_
                else:
                    cur_b = max(cur_b, b_)
            union += cur_b - cur_a
            # per kernel shape: ms per step, launches, fraction -- the row with the most time is the dominant kernel
            by_shape = {}
            for e in fam:
                d = by_shape.setdefault((e["dir"], e["ck"], e["cn"]), [0.0, 0.0, 0])
                d[0] += e["flops"]; d[1] += e["ms"]; d[2] += 1
            (ddir, dck, dcn), dom = max(by_shape.items(), key=lambda kv_: kv_[1][1])
            roof["kernel_is_dominant"] = (ddir, str(dck), str(dcn)) == (tdir, str(tck), str(tcn))
            roof["step_frac"] = round(f_flops / k / (dt / args.steps) / 1e12 / peak, 4)
            roof["family_frac"] = round(f_flops / (union * 1e-3) / 1e12 / peak, 4)
            roof.update({
                "family_tflops": round(f_flops / (union * 1e-3) / 1e12, 2),
                "family_busy_ms_per_step": round(union / k, 3),
                "family_frac_of_summed_kernel_time": round(f_flops / (f_ms * 1e-3) / 1e12 / peak, 4),
                "family_gflop_per_step": round(f_flops / k / 1e9, 2),
                "dominant_by_time": {"kernel": _kernel_name(ddir, dck, dcn), "ms_per_step": round(dom[1] / k, 3),
                                     "launches_per_step": round(dom[2] / k, 1), "frac": round(dom[0] / (dom[1] * 1e-3) / 1e12 / peak, 4)},
                "family_kernel_ms_per_step": {d: round(v[1] / k, 3) for d, v in per_dir.items()},
                "per_shape_ms_per_step": {f"{d_}<{c1},{c2}>": round(v[1] / k, 3) for (d_, c1, c2), v in
                                          sorted(by_shape.items(), key=lambda kv_: -kv_[1][1])[:8]},
                "family_note": f"{k} extra steps after the timed region, HIP events around every conv launch on its own stream; "
                               "family_frac = algorithmic flops of all conv kernels (forward, backward-input, weight gradient) / "
                               "wall time with at least one of them running; step_frac = the same flops / ms_per_step"})
        hb = _hbm_line(hbm_bn, "bn_bwd_dx_pow2_kernel (BatchNorm + ReLU backward: dx; every launch of the step)", "bn_bwd_dx")
        hc = _hbm_line(hbm_conv, "gather_gemm_v2_kernel<CK=8,CN=8,BWD=false> (stage-1 SubM conv forward)", "gather_gemm_8_8_fwd")
        if hb is not None:    # flat keys: the bound the metric names besides frames/s
            roof.update({"hbm_kernel": hb["kernel"], "hbm_algorithmic_mb_per_launch": hb["algorithmic_mb_per_launch"], "hbm_avg_us": hb["avg_us"],
                         "hbm_achieved_tb_s": hb["achieved_tb_s"], "hbm_frac_of_8tb_s": hb["frac_of_8tb_s"], "hbm_traffic_mb": hb["traffic_mb"],
                         "hbm_launches": hb["launches"], "hbm_traffic_source": hb["traffic_source"]})
        if hc is not None:
            roof.update({"hbm_conv_kernel": hc["kernel"], "hbm_conv_algorithmic_mb_per_launch": hc["algorithmic_mb_per_launch"],
                         "hbm_conv_avg_us": hc["avg_us"], "hbm_conv_achieved_tb_s": hc["achieved_tb_s"],
                         "hbm_conv_frac_of_8tb_s": hc["frac_of_8tb_s"], "hbm_conv_traffic_mb": hc["traffic_mb"],
                         "hbm_conv_launches": hc["launches"], "hbm_conv_traffic_source": hc["traffic_source"]})
        if plain and tdir != "dw":
            pr = _traced_roofline(plain, args, tdir, tck, tcn, pmc=False)
            if pr is not None:
                roof["frac_gemm_only"] = pr["frac"]
                roof["avg_us_gemm_only"] = pr["avg_us"]
                roof["gemm_only_note"] = (f"{args.family_steps} extra steps after the timed region with vc_debug_set conv_bn_finish = 0 "
                                          "(the same kernel without its BatchNorm-statistics tail)")
    if args.model == "8x":
        metric = "KITTI frames/sec (fwd+bwd) VirConv8x backbone (VirConv-T/S)"
        workload = ("BASELINE configs[3] backbone: VirConv8x (LiDAR stream + virtual-point MM stream) train step (fwd+bwd+Adam), "
                    "train mode, layer discard 0.15, bs 2 per GPU, <=16000 LiDAR + <=16000 fused voxels per frame; the "
                    "cascade refinement head is out of scope (SURVEY 8f)")
    else:
        metric = "KITTI frames/sec (fwd+bwd) VirConv-L backbone"
        workload = ("BASELINE configs[2]: VirConv-L train step (fwd+bwd+Adam), train mode, layer discard 0.1, NRConv 2-D branch "
                    "on, synthetic KITTI frames (20k LiDAR + 60k virtual points, input discard 0.8, <=40000 voxels/frame)")
        if args.frontend:
            workload += ("; PLUS the GPU data front-end inside every timed step: input point discard (HIP) + LiDAR-first "
                         "voxeliser + MeanVFE from raw device-resident points (virtual points fp16)")
    res = {
        "metric": metric, "value": round(frames / dt, 3), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "host_enqueue_ms_per_step": None if host_ms is None else round(host_ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": _dtype(be, args.operand),
        "data": "synthetic",
        "config": {"workload": workload,
                   "frames_per_gpu": bs, "global_batch": bs * world,
                   "voxels_rank0": n_vox if args.frontend else int(batch["voxel_features"].shape[0]),
                   "parallelism": f"dp{world}", "cpu_affinity": numa, "per_rank": per_rank,
                   "untimed_setup": {"settle_seconds_of_steps_before_warmup": settle, "allocator_priming_gib": 8,
                                     "gc_freeze": os.environ.get("VIRCONV_GC_FREEZE", "1") != "0",
                                     "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                                     "streams": ("main (high priority) + geometry plan (high priority) + weight-gradient side stream (normal priority)"
                                                 if os.environ.get("VIRCONV_MAIN_PRIORITY", "-1") == "-1" else
                                                 "main + geometry plan (high priority) + weight-gradient side stream; main priority "
                                                 + os.environ["VIRCONV_MAIN_PRIORITY"]),
                                     "row_order": ops.ROW_ORDER,
                                     "parameters": (f"{len(opt_params)} flat tensor(s) aliased by the modules' parameters (feature_pass.flatten_parameters)"
                                                    if flat_on and len(opt_params) < 10 else "per module"),
                                     "optimizer": ("virconv_amd.optim.ClipAdamW (vc_clip_adamw: clip_grad_norm_(10) + AdamW in two launches)" if fused_opt
                                                   else "torch clip_grad_norm_(10) + torch.optim.AdamW(fused)")}},
        "roofline": roof,
        "exact_f32_mfma": exact,
    }
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline()
    else:
        res["cpu_baseline"] = None
    return res


if __name__ == "__main__":
    _res = main()            # rank 0: the result line; other ranks: None
    parallel.shutdown()      # tear the process group down BEFORE interpreter exit (its watchdog thread otherwise races HIP's teardown)
    if _res is not None:
        _emit(_res)
