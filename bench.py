#!/usr/bin/env python3
"""bench.py -- STFT frames/sec through log-mel -> K-means -> tokenise on MI355X.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): STFT frames/sec through K-means+tokenize, n_mels=64, vocab=8192.
Workload: configs[3] ("unbal_train 200k-clip subset, n_mels=64, vocab_size=8192, 8 GPUs") cut into
its eight per-GPU shards -- 22 500 train + 2 500 validation synthetic 10 s clips per GPU (weak
scaling: N GPUs process N shards; N = 8 is configs[3] itself).  One step = one full pass of the hot
path over the resident waveforms: fused log-mel (frame-major, unit rows) -> one FAISS-style
Kmeans.train per batch of 10 000 files (20 Lloyd iterations on a 2 097 152-row subsample, warm
started) -> centroid normalisation -> nearest-centroid tokens for every frame.  Inputs are in HBM
before the timed region starts.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2516.8  # same guide: BF16/F16 MFMA ~2.5 PF dense = 16 x the fp32 MFMA rate


def cpu_baseline(n_mels, vocab, L, hop, seed, clips, niter):
    """The oracle (CPU restatement, kind="port") timed on this box's host cores on a bounded
    sample of the same workload.  Only this function touches oracle/."""
    import oracle
    from audio_tokens_amd.synth import synth_clips
    oracle.build()
    wave = synth_clips(clips, L=L, seed=seed, first_clip=0, device="cpu").numpy()
    t0 = time.perf_counter()
    specs = [oracle.logmel(w, n_mels=n_mels, hop=hop) for w in wave]
    x = np.concatenate([s.T for s in specs], axis=0).astype(np.float32)
    x = oracle.l2norm_rows(x)
    t1 = time.perf_counter()
    r = oracle.kmeans_train(x, vocab, niter=niter)
    c = oracle.l2norm_rows(r.centroids)
    t2 = time.perf_counter()
    oracle.assign(x, c)
    t3 = time.perf_counter()
    frames = x.shape[0]
    return {
        "value": frames / (t3 - t0),
        "unit": "frames/s",
        "cores": oracle.num_threads(),
        "kind": "port",
        "sample": (f"{clips} clips ({frames} frames) of the same synthetic stream: log-mel, one Kmeans.train "
                   f"(k={vocab}, niter={niter} ~ the full job's 2.9 Lloyd point-iterations per frame), tokenise; "
                   f"seconds: logmel {t1 - t0:.2f}, kmeans {t2 - t1:.2f}, tokenise {t3 - t2:.2f}; "
                   f"host has {os.cpu_count()} logical cores"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--train-clips", type=int, default=22500, help="per GPU")
    ap.add_argument("--val-clips", type=int, default=2500, help="per GPU")
    ap.add_argument("--n-mels", type=int, default=64)
    ap.add_argument("--vocab", type=int, default=8192)
    ap.add_argument("--niter", type=int, default=20)
    ap.add_argument("--clip-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=256)
    ap.add_argument("--host-waves", action="store_true",
                    help="also time DevicePipeline.run_streaming on pinned host copies of the same waveforms and report it "
                         "as pcie_inclusive (never `value`)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (no CPU fallback)"
    # Rehearsal switches (not used by the driver): AT_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # AT_BENCH_BACKEND=gloo exchanges through the host, so the N > 1 code path can be exercised on a
    # one-GPU box.  The real multi-GPU run is one rank per GPU over RCCL ("nccl").
    if os.environ.get("AT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("AT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from audio_tokens_amd.backend import default_backend
    from audio_tokens_amd.pipeline import DevicePipeline
    from audio_tokens_amd.synth import synth_clips

    be = default_backend(device)
    sr, hop, n_fft = 22050, 128, 512
    L = int(round(args.clip_seconds * sr))
    T = be.num_frames(L, hop)
    seed = 4242
    n_tr, n_va = args.train_clips, args.val_clips
    # global clip ids: train clips first (rank-major inside every 10 000-file batch is implied by
    # the sharded Kmeans), then validation clips; any rank can generate its own shard
    wave_tr = synth_clips(n_tr, L=L, seed=seed, first_clip=rank * n_tr, device=device)
    wave_va = synth_clips(n_va, L=L, seed=seed, first_clip=world * n_tr + rank * n_va, device=device)

    pipe = DevicePipeline(n_mels=args.n_mels, vocab_size=args.vocab, niter=args.niter, sample_rate=sr,
                          n_fft=n_fft, hop_length=hop, clustering_batch_size=10000,
                          distributed=world > 1, backend=be)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        res = pipe.run(wave_tr, wave_va)
    be.assign_trace = []
    be.prune_stats(reset=True)
    be.filter_stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = pipe.run(wave_tr, wave_va)
    barrier()
    elapsed = time.perf_counter() - t0
    trace, be.assign_trace = be.assign_trace, None
    f_rows, f_listed, f_ms, f_sweeps, f_tiles, f_refined = be.filter_stats(timing=True)   # timed steps only
    needed, total = be.prune_stats()
    stage = pipe.run(wave_tr, wave_va, timing=True).stage_seconds  # one extra, untimed, per-stage split

    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    frames_per_step = (n_tr + n_va) * T * world
    value = frames_per_step * args.steps / elapsed

    # Roofline of the dominant kernel, from HIP events recorded on the launch stream around every
    # nearest-centroid launch of the timed steps.  Kinds: "pruned" = assign_mfma_pruned_kernel in exact
    # mode (Lloyd iterations 2..20, plus the last stage of the unguided search used by iteration 1 and
    # by tokenise); "coarse" = the same kernel as guess generator; "plain" = assign_mfma_kernel (here:
    # rows against the 256 group means); "hinted" = assign_mfma_hinted_kernel (only when pruning is
    # off).  ALGORITHMIC flops = 2*d*k per row for the exact kinds -- what IndexFlatL2.search must
    # evaluate; the pruned sweep provably (bit-exact results) skips most 32x32 accumulators, so its
    # algorithmic rate can exceed the MFMA peak.  executed_* prices only the accumulators computed.
    def agg(kind):
        sel = [t for t in trace if t[0] == kind]
        fl = sum(2.0 * n * d * k for (_, n, d, k, _, _) in sel)
        ms = sum(e0.elapsed_time(e1) for (_, _, _, _, e0, e1) in sel)
        return {"launches": len(sel), "flop": fl, "ms": ms}

    kinds = {kd: agg(kd) for kd in ("pruned", "coarse", "plain", "hinted")}
    filtered = f_sweeps > 0
    names = {"pruned": ("at_assign_pruned_f32 exact call: assign_f16filter_kernel<64,2,false,true,3> + exact_dist_todo_kernel + fp32 redo "
                        "of the listed rows (exact_rows_kernel<64>)") if filtered
             else "assign_mfma_pruned_reg_kernel<64,2> (at_assign_pruned_f32, exact mode)",
             "coarse": ("assign_f16filter_kernel<64,4,true,false>" if filtered else "assign_mfma_pruned_reg_kernel<64,2>") + " (guess generator)",
             "plain": "assign_mfma_kernel<64,2,4,DMA,2> (at_assign_f32)",
             "hinted": "assign_mfma_hinted_kernel<64,2,4> (at_assign_hinted_f32)"}
    dom = max(kinds, key=lambda kd: kinds[kd]["ms"])
    D = kinds[dom]
    exec_frac = needed / total if total else None
    traffic = None
    tfile = ROOT / "profiles" / "assign_traffic.json"
    if tfile.exists():
        try:
            tj = json.loads(tfile.read_text())
            want = "assign_f16filter_kernel" if (filtered and dom == "pruned") else "assign_mfma_pruned_reg_kernel"
            traffic = tj.get("hbm_bytes_per_launch") if want in tj.get("kernel", "") else None
        except Exception:
            traffic = None
    ms_all = sum(v["ms"] for v in kinds.values())
    per_kind = {kd: {"kernel": names[kd], "launches": v["launches"], "avg_launch_ms": v["ms"] / max(1, v["launches"]),
                     "algorithmic_tflops": v["flop"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0}
                for kd, v in kinds.items() if v["launches"]}
    if filtered and dom == "pruned":
        # Dominant kernel = the fp16-split filter sweep (stage 1 of every exact call), timed by HIP events
        # the library records on the launch stream around that kernel alone (at_filter_stats).  Its
        # algorithmic work = 2*d*k flop for every row it settles (rows it lists for the fp32 redo are
        # not credited).  It issues v_mfma_f32_32x32x16_f16: three fp16 MFMAs per fp32 one, on the
        # accumulators the exact pruning bound leaves.
        rows_settled = f_rows - f_listed
        flop = 2.0 * args.n_mels * args.vocab * rows_settled
        achieved = flop / (f_ms * 1e-3) / 1e12 if f_ms > 0 else 0.0
        ns = args.n_mels // 16
        mfma_issued = f_tiles * ns + f_refined * 2 * ns            # v_mfma_f32_32x32x16_f16 instructions
        exec_f16 = mfma_issued * 32768.0 / (f_ms * 1e-3) / 1e12 if f_ms > 0 else None
        roofline = {
            "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
            "kernel": "assign_f16filter_kernel<64,2,false,true,3> (stage 1 of at_assign_pruned_f32, exact mode)",
            "launches": f_sweeps, "avg_launch_ms": f_ms / f_sweeps, "flop_per_launch": flop / f_sweeps,
            "share_of_step_time": (f_ms * 1e-3) / elapsed if elapsed > 0 else None,
            "rows_listed_for_fp32_redo_fraction": f_listed / f_rows if f_rows else None,
            "accumulators_computed_fraction": exec_frac,
            "tiles_refined_with_lo_products_fraction": f_refined / f_tiles if f_tiles else None,
            "executed_mfma_dtype": "f16 (fp32 accumulate)", "executed_tflops": exec_f16,
            "executed_peak": PEAK_F16_MFMA_TFLOPS,
            "executed_frac": exec_f16 / PEAK_F16_MFMA_TFLOPS if exec_f16 is not None else None,
            "exact_call": {"kernel": names["pruned"], "launches": D["launches"],
                           "avg_ms": D["ms"] / max(1, D["launches"]),
                           "algorithmic_tflops": D["flop"] / (D["ms"] * 1e-3) / 1e12 if D["ms"] > 0 else 0.0,
                           "share_of_step_time": (D["ms"] * 1e-3) / elapsed if elapsed > 0 else None},
            "note": ("achieved/frac are ALGORITHMIC fp32 flop (2*d*k per row = the dense IndexFlatL2 search the contract "
                     "specifies) against the fp32 MFMA peak; > 1 because (a) a rounding-safe triangle-inequality bound skips "
                     "most 32x32 accumulators and (b) the surviving ones are evaluated with fp16 MFMAs (hi*hi first, the two lo "
                     "products only for tiles that can matter) whose error is bounded a priori, a row being accepted only when "
                     "its runner-up is provably out of reach of the fp32 contract; the other rows are redone in fp32.  ids/dist/centroids are bit-identical to the dense fp32 "
                     "sweep (tests/test_gpu_ops.py::test_assign_pruned_is_exact, test_filter_*).  executed_* prices the "
                     "fp16 MFMA flop actually issued against the dense fp16 peak.  The dense fp32 kernel "
                     "(assign_mfma_kernel) runs at 132 TFLOP/s = 84 % of the fp32 peak, see profiles/."),
            "all_nearest_centroid_launches": {"share_of_step_time": (ms_all * 1e-3) / elapsed if elapsed > 0 else None,
                                              **per_kind},
        }
    else:
        achieved = D["flop"] / (D["ms"] * 1e-3) / 1e12 if D["ms"] > 0 else 0.0
        roofline = {
            "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
            "kernel": names[dom], "launches": D["launches"],
            "avg_launch_ms": D["ms"] / max(1, D["launches"]), "flop_per_launch": D["flop"] / max(1, D["launches"]),
            "share_of_step_time": (D["ms"] * 1e-3) / elapsed if elapsed > 0 else None,
            # rocprofv3 reports one average per kernel symbol; the pruned sweep kernel also runs in guess-generator
            # mode ("coarse"), so this is the figure its average is to be compared with
            "kernel_avg_launch_ms_all_modes": ((kinds["pruned"]["ms"] + kinds["coarse"]["ms"]) /
                                               max(1, kinds["pruned"]["launches"] + kinds["coarse"]["launches"]))
            if dom == "pruned" else None,
            "accumulators_computed_fraction": exec_frac,
            "executed_tflops": achieved * exec_frac if (exec_frac is not None and dom == "pruned") else None,
            "executed_frac": (achieved * exec_frac / PEAK_F32_MFMA_TFLOPS) if (exec_frac is not None and dom == "pruned") else None,
            "note": ("achieved/frac are algorithmic (2*d*k flop per row, the dense IndexFlatL2 search): > 1 means the exact "
                     "pruned sweep skipped accumulators that a rounding-safe triangle-inequality bound rules out; results are "
                     "bit-identical to the dense sweep (tests/test_gpu_ops.py::test_assign_pruned_is_exact). executed_* counts "
                     "only computed accumulators; the dense kernel (assign_mfma_kernel) runs at 132 TFLOP/s = 84 % of peak, "
                     "see profiles/."),
            "all_nearest_centroid_launches": {"share_of_step_time": (ms_all * 1e-3) / elapsed if elapsed > 0 else None,
                                              **per_kind},
        }

    out = {
        "metric": "STFT frames/sec through K-means+tokenize, n_mels=64 vocab=8192",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": (f"configs[3] per-GPU shard x{world}: {n_tr}+{n_va} clips/GPU of {args.clip_seconds:g} s @22.05 kHz "
                         f"({(n_tr + n_va) * T} frames/GPU), n_mels={args.n_mels}, vocab_size={args.vocab}, niter={args.niter}, "
                         f"k-means batches of 10000 files on a {min(256 * args.vocab, 10000 * T)}-row subsample"),
            "frames_per_step": frames_per_step, "parallelism": f"dp{world}",
        },
        "stage_seconds": stage,
        "roofline": roofline,
    }
    if args.host_waves:
        host_tr, host_va = wave_tr.cpu().pin_memory(), wave_va.cpu().pin_memory()
        pipe.run_streaming(host_tr, host_va)                     # warm-up
        barrier()
        t0 = time.perf_counter()
        pipe.run_streaming(host_tr, host_va)
        barrier()
        dt = time.perf_counter() - t0
        out["pcie_inclusive"] = {"value": frames_per_step / world / dt * world, "unit": "frames/s", "ms_per_step": dt * 1e3,
                                 "note": "waveforms in pinned host memory, streamed in 5000-clip chunks; frames kept only per "
                                         "k-means batch and recomputed for tokenise; tokens returned to the host"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.n_mels, args.vocab, L, hop, seed, args.cpu_clips, niter=3)
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
