#!/usr/bin/env python3
"""bench.py -- STFT frames/sec through log-mel -> K-means -> tokenise on MI355X.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): STFT frames/sec through K-means+tokenize, n_mels=64, vocab=8192.
Default workload (--config 3): configs[3] ("unbal_train 200k-clip subset, n_mels=64, vocab_size=8192, 8 GPUs")
cut into its eight per-GPU shards -- 22 500 train + 2 500 validation synthetic 10 s clips per GPU (weak
scaling: N GPUs process N shards; N = 8 is configs[3] itself).  --config 1 / 2 run BASELINE.json's
single-GPU configs[1] (19 944 + 2 216 clips, n_mels=64, vocab 500) and configs[2] (same clips, n_mels=128,
vocab 8192) instead; their lines are committed under profiles/, the driver's line stays on --config 3.

One step = one full pass of the hot path over the resident waveforms, exactly what one run of the
reference's three stages computes: fused log-mel (frame-major, unit rows) -> one FAISS-style Kmeans.train
per batch of 10 000 files (subsample permutation, 20 Lloyd iterations on 256*k rows, warm started) ->
centroid normalisation -> nearest-centroid tokens for every frame.  Nothing is carried from one step to
the next (the subsample permutation is computed inside every step, on the device).  Inputs are in HBM before
the timed region starts.  Rank 0 prints ONE JSON line.

Kernel times for the roofline come from HIP events inside the timed steps: the pair the library records around the
fp16 filter sweep on every exact call, or -- when another kernel class dominates -- a pair around each launch of that
class.  Which class dominates is decided by one extra UNTIMED step behind the W warm-up steps in which every launch
is bracketed (`traced_step_ms`, `roofline.kernel_classes`); with W = 0 the timed steps themselves are traced in full.

Outside the timed region the same step is run once more with every acceleration off (dense fp32 sweeps):
`dense_floor` is its rate -- what the path does on data that defeats the pruning -- and `verified` says its
tokens and centroids equal the timed configuration's bit for bit.
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

# /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_MFMA_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2516.8  # BF16/F16 MFMA, dense = 16 x the fp32 MFMA rate
PEAK_HBM_GBS = 8000.0          # HBM3E

CONFIGS = {
    1: dict(name="configs[1]", train=19944, val=2216, n_mels=64, vocab=500),
    2: dict(name="configs[2]", train=19944, val=2216, n_mels=128, vocab=8192),
    3: dict(name="configs[3] per-GPU shard", train=22500, val=2500, n_mels=64, vocab=8192),
}


def cpu_logmel_torch(wave, n_mels, hop, n_fft=512, sr=22050):
    """SURVEY.md section 8d's CPU log-mel: torch.stft the way torchaudio's MelSpectrogram calls it (periodic Hann,
    center, reflect padding, power 2), the HTK filterbank, 10 log10(clamp) -- fp32 on torch's CPU threads."""
    from audio_tokens_amd.backend import HostHelpers
    fb = torch.from_numpy(HostHelpers().mel_filterbank(sr, n_fft, n_mels))
    win = torch.hann_window(n_fft, periodic=True)
    out = []
    for c0 in range(0, wave.shape[0], 32):
        st = torch.stft(wave[c0:c0 + 32], n_fft, hop_length=hop, win_length=n_fft, window=win, center=True,
                        pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        power = st.abs().pow(2.0)                                   # [clips, n_freq, T]
        mel = torch.matmul(power.transpose(1, 2), fb)               # [clips, T, n_mels]
        out.append(10.0 * torch.log10(torch.clamp(mel, min=1e-10)))
    return torch.cat(out).reshape(-1, n_mels).numpy()               # frame-major, as the k-means stage wants it


def cpu_baseline(n_mels, vocab, L, hop, seed, clips, niter, gpu_note=""):
    """The CPU path timed on this box's host cores on a bounded sample of the same workload (kind="port": faiss and
    torchaudio are not installable here).  Log-mel: torch.stft-based, fp32, torch's CPU threads;
    the oracle's double-precision log-mel is timed beside it.  K-means and tokenise: the oracle (C, OpenMP).
    Only this function touches oracle/."""
    import oracle
    from audio_tokens_amd.synth import synth_clips
    oracle.build()
    wave_t = synth_clips(clips, L=L, seed=seed, first_clip=0, device="cpu")
    t0 = time.perf_counter()
    x = cpu_logmel_torch(wave_t, n_mels, hop)
    x = oracle.l2norm_rows(np.ascontiguousarray(x, dtype=np.float32))
    t1 = time.perf_counter()
    r = oracle.kmeans_train(x, vocab, niter=niter)
    c = oracle.l2norm_rows(r.centroids)
    t2 = time.perf_counter()
    oracle.assign(x, c)
    t3 = time.perf_counter()
    wave = wave_t.numpy()
    specs = [oracle.logmel(w, n_mels=n_mels, hop=hop) for w in wave[:max(1, clips // 4)]]
    t4 = time.perf_counter()
    del specs
    frames = x.shape[0]
    return {
        "value": frames / (t3 - t0),
        "unit": "frames/s",
        "cores": oracle.num_threads(),
        "torch_threads": torch.get_num_threads(),
        "os_cpu_count": os.cpu_count(),
        "kind": "port",
        "sample": (f"{clips} clips ({frames} frames) of the same synthetic stream: torch.stft log-mel (fp32, {torch.get_num_threads()} "
                   f"torch threads), then the oracle with {oracle.num_threads()} OpenMP threads: one Kmeans.train (k={vocab}, niter={niter} "
                   f"-- {gpu_note}: {niter} iteration{'s' if niter != 1 else ''} over the sample's frames "
                   f"{'do the same work per frame' if niter > 1 else 'is the least a training can do'}) and tokenise; "
                   f"seconds: logmel {t1 - t0:.2f}, kmeans {t2 - t1:.2f}, tokenise {t3 - t2:.2f}; the oracle's own "
                   f"double-precision log-mel takes {(t4 - t3) * clips / max(1, clips // 4):.2f} s for the same clips; "
                   f"host has {os.cpu_count()} logical cores"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json configs[i]")
    ap.add_argument("--train-clips", type=int, default=None, help="per GPU (default: the config's)")
    ap.add_argument("--val-clips", type=int, default=None, help="per GPU")
    ap.add_argument("--n-mels", type=int, default=None)
    ap.add_argument("--vocab", type=int, default=None)
    ap.add_argument("--niter", type=int, default=20)
    ap.add_argument("--clip-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-floor", action="store_true", help="skip the untimed dense run (dense_floor / verified = null)")
    ap.add_argument("--no-hard-workload", action="store_true", help="skip the noise-dominated workload (hard_workload = null)")
    ap.add_argument("--cpu-clips", type=int, default=256)
    ap.add_argument("--host-waves", action="store_true",
                    help="also time DevicePipeline.run_streaming on pinned host copies of the same waveforms and report it "
                         "as pcie_inclusive (never `value`)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    n_tr = cfg["train"] if args.train_clips is None else args.train_clips
    n_va = cfg["val"] if args.val_clips is None else args.val_clips
    n_mels = cfg["n_mels"] if args.n_mels is None else args.n_mels
    vocab = cfg["vocab"] if args.vocab is None else args.vocab

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    # Rehearsal switches (not used by the driver): AT_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # AT_BENCH_BACKEND=gloo exchanges through the host, so the N > 1 code path can be exercised on a
    # one-GPU box.  The real multi-GPU run is one rank per GPU over RCCL ("nccl").
    if os.environ.get("AT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("AT_BENCH_BACKEND", "nccl")
    dist = None
    comm = None
    if world > 1:
        # the process group comes first: RCCL binds this rank to its device before anything else touches a GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        assert torch.cuda.device_count() > local_rank, f"rank {rank}: no device {local_rank}"
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        comm = {"backend": dist.get_backend(), "ranks": dist.get_world_size()}
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from audio_tokens_amd.backend import default_backend
    from audio_tokens_amd.pipeline import DevicePipeline
    from audio_tokens_amd.synth import synth_clips

    be = default_backend(device)
    be.debug_set("filter_timing", 1)   # the library brackets the stage-1 kernel of its exact calls with HIP events (roofline)
    sr, hop, n_fft = 22050, 128, 512
    L = int(round(args.clip_seconds * sr))
    T = be.num_frames(L, hop)
    seed = 4242
    # global clip ids: train clips first (rank-major inside every 10 000-file batch is implied by
    # the sharded Kmeans), then validation clips; any rank can generate its own shard
    wave_tr = synth_clips(n_tr, L=L, seed=seed, first_clip=rank * n_tr, device=device)
    wave_va = synth_clips(n_va, L=L, seed=seed, first_clip=world * n_tr + rank * n_va, device=device)

    def make_pipe(prune=True):
        return DevicePipeline(n_mels=n_mels, vocab_size=vocab, niter=args.niter, sample_rate=sr, n_fft=n_fft,
                              hop_length=hop, clustering_batch_size=10000, distributed=world > 1, backend=be, prune=prune)

    pipe = make_pipe()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    # Every traced launch is bracketed by two timing events on the launch stream, and those cost: the step is 170.5-172.5
    # ms with all ~150 launches of a step traced against 163.7-163.9 ms with none (same box).  So ONE MORE UNTIMED step
    # behind the warm-up steps is traced in full -- it names the dominant kernel class and gives the per-class table --
    # and the timed steps trace that one class only.  (The fp16 filter sweep needs no tracing at all: the library itself brackets that kernel with
    # events on every exact call, for at_filter_stats.)  Without a warm-up step the timed steps are traced in full.
    KINDS = ("logmel", "pruned", "coarse", "plain", "hinted")

    def agg(tr, kind):
        sel = [t for t in tr if t[0] == kind]
        ms = sum(e0.elapsed_time(e1) for (_, _, _, _, e0, e1) in sel)
        return {"launches": len(sel), "ms": ms, "sel": sel}

    warm_ms = []
    warm_kinds, warm_step_s, traced_ms = None, None, None
    for w in range(args.warmup):
        barrier()
        t0 = time.perf_counter()
        res = pipe.run(wave_tr, wave_va)
        barrier()
        warm_ms.append((time.perf_counter() - t0) * 1e3)
    if args.warmup > 0:   # one more untimed step, this one traced in full
        be.assign_trace, be.assign_trace_only = [], None
        be.filter_stats(reset=True)
        # (the class table wants every kernel's stand-alone duration: this one step keeps the log-mel launches on the main
        # stream; beside the training they share the chip and their own events would count the sharing as their time)
        overlap, pipe.overlap_logmel = pipe.overlap_logmel, False
        barrier()
        t0 = time.perf_counter()
        res = pipe.run(wave_tr, wave_va)
        barrier()
        traced_ms = (time.perf_counter() - t0) * 1e3
        pipe.overlap_logmel = overlap
        wtrace, be.assign_trace = be.assign_trace, None
        warm_kinds = {kd: agg(wtrace, kd) for kd in KINDS}
        _, _, w_fms, w_fsweeps, _, _ = be.filter_stats(timing=True)
        if w_fsweeps > 0:
            warm_kinds["pruned"]["kernel_ms"] = w_fms
        warm_step_s = traced_ms * 1e-3
    dom_hint = max(warm_kinds, key=lambda kd: warm_kinds[kd].get("kernel_ms", warm_kinds[kd]["ms"])) if warm_kinds else None
    be.assign_trace = []
    be.assign_trace_only = None if dom_hint is None else ({dom_hint} if not (dom_hint == "pruned" and "kernel_ms" in warm_kinds["pruned"]) else set())
    be.prune_stats(reset=True)
    be.filter_stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = pipe.run(wave_tr, wave_va)
    barrier()
    elapsed = time.perf_counter() - t0
    trace, be.assign_trace, be.assign_trace_only = (be.assign_trace or []), None, None
    f_rows, f_listed, f_ms, f_sweeps, _, _ = be.filter_stats(timing=True)   # timed steps only
    # One extra, untimed step: the per-stage split, and -- with the counting switch on, which the product leaves off --
    # the tiles / accumulators the sweeps computed.  The step is deterministic, so the timed steps computed the same.
    be.debug_set("filter_stats", 1)
    be.prune_stats(reset=True)
    be.filter_stats(reset=True)
    stage = pipe.run(wave_tr, wave_va, timing=True).stage_seconds
    _, _, _, _, f_tiles, f_refined = be.filter_stats(timing=True)
    f_tiles, f_refined = f_tiles * args.steps, f_refined * args.steps
    stage_all = None
    if dist is not None:   # every rank's split, for the scaling post-mortem
        assert dist.get_world_size() == world == int(os.environ["WORLD_SIZE"]), "process group size != WORLD_SIZE"
        stage_all = [None] * world
        dist.all_gather_object(stage_all, {k2: round(v2, 6) for k2, v2 in stage.items()})
    needed, total = be.prune_stats()
    be.debug_set("filter_stats", 0)

    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    frames_per_step = (n_tr + n_va) * T * world
    value = frames_per_step * args.steps / elapsed

    # ---- the same step with every acceleration off: the floor, and the check of what was timed -------------
    dense_floor, verified = None, None
    if not args.no_dense_floor:
        dpipe = make_pipe(prune=False)
        barrier()
        t0 = time.perf_counter()
        dres = dpipe.run(wave_tr, wave_va)
        barrier()
        dt = time.perf_counter() - t0
        same = (torch.equal(res.centroids.view(torch.int32), dres.centroids.view(torch.int32))
                and torch.equal(res.tokens_train, dres.tokens_train) and torch.equal(res.tokens_val, dres.tokens_val))
        if dist is not None:
            flags = torch.tensor([dt, 0.0 if same else 1.0], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(flags, op=dist.ReduceOp.MAX)
            dt, same = float(flags[0].item()), float(flags[1].item()) == 0.0
        verified = bool(same)
        dense_floor = {"value": frames_per_step / dt, "unit": "frames/s", "ms_per_step": dt * 1e3,
                       "note": "one untimed step with pruning, the fp16 filter and the guess generators off: plain dense fp32 MFMA "
                               "sweeps (assign_mfma_kernel / assign_mfma_hinted_kernel); `verified` compares its centroids and "
                               "tokens with the timed configuration's, bit for bit"}
        del dres

    # ---- roofline of the dominant kernel, from HIP events recorded on the launch stream ----------------------
    # Kinds: "logmel" = logmel_kernel launches; "pruned" = at_assign_pruned_f32 exact calls (the fp16-filter sweep
    # + redo); "coarse" = guess generators; "plain" = assign_mfma_kernel (dense); "hinted" = assign_mfma_hinted_kernel.
    kinds = {kd: agg(trace, kd) for kd in KINDS}                 # timed steps (the dominant class, or all without warm-up)
    filtered = f_sweeps > 0
    if filtered:   # the stage-1 kernel of the exact calls is timed by the library's own events around that kernel alone
        kinds["pruned"]["kernel_ms"] = f_ms
    dom = dom_hint if dom_hint is not None else max(kinds, key=lambda kd: kinds[kd].get("kernel_ms", kinds[kd]["ms"]))
    D = kinds[dom]
    step_share = lambda ms: (ms * 1e-3) / elapsed if elapsed > 0 else None   # noqa: E731
    traffic_file = ROOT / "profiles" / "kernel_traffic.json"
    traffic_tab = json.loads(traffic_file.read_text()) if traffic_file.exists() else {}

    def traffic_of(key):
        t = traffic_tab.get(key)
        return t.get("hbm_bytes_per_launch") if isinstance(t, dict) else None

    if dom == "logmel":
        frames = sum(n for (_, n, _, _, _, _) in D["sel"])
        byts = frames * (hop * 4 + n_mels * 4)
        achieved = byts / (D["ms"] * 1e-3) / 1e9
        kname = "logmel_kernel<true>"
        roofline = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                    "traffic": traffic_of(f"logmel_d{n_mels}"), "kernel": kname, "launches": D["launches"],
                    "avg_launch_ms": D["ms"] / max(1, D["launches"]), "bytes_per_launch": byts / max(1, D["launches"]),
                    "share_of_step_time": step_share(D["ms"]),
                    "note": "algorithmic bytes = (hop*4 + n_mels*4) per frame: every sample read once, every output written once"}
    elif dom == "pruned" and filtered:
        # Dominant kernel = the fp16-split filter sweep (stage 1 of every exact call).  `frac` prices the fp16 MFMA
        # instructions the kernel ISSUED (counted by the kernel itself per 32x32 tile) against the dense fp16 MFMA
        # peak -- a real fraction of a real roof.  The contract's dense work (2*d*k flop per row) divided by the same
        # time is reported separately as algorithmic_rate_vs_dense_fp32_peak: it exceeds 1 because a rounding-safe
        # bound skips most tiles and the rest are decided in fp16, with bit-identical results (`verified`).
        ns = n_mels // 16
        mfma_issued = f_tiles * ns + f_refined * 2 * ns            # v_mfma_f32_32x32x16_f16 instructions
        exec_tf = mfma_issued * 32768.0 / (f_ms * 1e-3) / 1e12 if f_ms > 0 else 0.0
        rows_settled = f_rows - f_listed
        alg_tf = 2.0 * n_mels * vocab * rows_settled / (f_ms * 1e-3) / 1e12 if f_ms > 0 else 0.0
        wps = "3" if n_mels == 64 else "2"
        key = f"filter_d{n_mels}"
        tr = traffic_of(key)
        avg_ms = f_ms / f_sweeps
        roofline = {"bound": "mfma", "achieved": exec_tf, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": exec_tf / PEAK_F16_MFMA_TFLOPS, "traffic": tr,
                    "kernel": f"assign_f16filter_kernel<{n_mels},...> (stage 1 of at_assign_pruned_f32 exact calls; Lloyd sweeps at {wps} "
                              "waves/SIMD and the long tokenise sweeps)",
                    "launches": f_sweeps, "avg_launch_ms": avg_ms, "flop_per_launch": mfma_issued * 32768.0 / f_sweeps,
                    "mfma_dtype": "f16 inputs, f32 accumulate (v_mfma_f32_32x32x16_f16)",
                    "share_of_step_time": step_share(f_ms),
                    "hbm_frac": (tr / (traffic_tab[key].get("launch_ms", avg_ms) * 1e-3) / 1e9 / PEAK_HBM_GBS) if tr else None,
                    "algorithmic_rate_vs_dense_fp32_peak": alg_tf / PEAK_F32_MFMA_TFLOPS,
                    "algorithmic_tflops": alg_tf,
                    "accumulators_computed_fraction": needed / total if total else None,
                    "tiles_refined_with_lo_products_fraction": f_refined / f_tiles if f_tiles else None,
                    "rows_listed_for_fp32_redo_fraction": f_listed / f_rows if f_rows else None,
                    # (the whole exact call -- sweep, exact distances, redo -- bracketed by events: from the traced
                    # untimed step when there is one, else from the timed steps)
                    "exact_call": ({"launches_per_step": warm_kinds["pruned"]["launches"],
                                    "avg_ms": warm_kinds["pruned"]["ms"] / max(1, warm_kinds["pruned"]["launches"]),
                                    "share_of_step_time": warm_kinds["pruned"]["ms"] * 1e-3 / warm_step_s}
                                   if warm_kinds is not None else
                                   {"launches": D["launches"], "avg_ms": D["ms"] / max(1, D["launches"]),
                                    "share_of_step_time": step_share(D["ms"])}),
                    "note": "frac = issued fp16 MFMA flop / dense fp16 MFMA peak (the kernel is latency-bound: see DESIGN.md section 5); "
                            "hbm_frac = PMC bytes per launch / launch time under the profiler / 8 TB/s, measured on the form of the kernel that is timed here (a Lloyd-shaped sweep over a random subsample: profiles/kernel_traffic.json)"}
    else:
        flop = sum(2.0 * n * d * k for (_, n, d, k, _, _) in D["sel"])
        achieved = flop / (D["ms"] * 1e-3) / 1e12 if D["ms"] > 0 else 0.0
        kname = {"plain": f"assign_mfma_kernel<{n_mels},...> (at_assign_f32, dense)",
                 "hinted": f"assign_mfma_hinted_kernel<{n_mels},...> (at_assign_hinted_f32, dense)",
                 "pruned": f"assign_mfma_pruned_reg_kernel<{n_mels},2> (exact pruned fp32 sweep)",
                 "coarse": "guess generator"}[dom]
        executed = achieved * (needed / total) if (dom == "pruned" and total) else achieved
        roofline = {"bound": "mfma", "achieved": executed, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": executed / PEAK_F32_MFMA_TFLOPS, "traffic": traffic_of(f"{dom}_d{n_mels}"), "kernel": kname,
                    "launches": D["launches"], "avg_launch_ms": D["ms"] / max(1, D["launches"]),
                    "flop_per_launch": flop / max(1, D["launches"]), "share_of_step_time": step_share(D["ms"]),
                    "algorithmic_rate_vs_dense_fp32_peak": achieved / PEAK_F32_MFMA_TFLOPS,
                    "note": "2*d*k flop per row per launch (the dense IndexFlatL2 search), fp32 MFMA"}
    if warm_kinds is not None:   # the per-class table comes from the fully traced untimed step
        roofline["kernel_classes"] = {
            "source": f"an untimed step behind the warm-up steps, every launch traced and nothing overlapped ({traced_ms:.1f} ms)",
            **{kd: {"launches": v["launches"], "ms_per_step": v["ms"], "share_of_step_time": v["ms"] * 1e-3 / warm_step_s}
               for kd, v in warm_kinds.items() if v["launches"]}}
    else:
        roofline["kernel_classes"] = {
            "source": "timed steps, every launch traced",
            **{kd: {"launches": v["launches"], "ms_per_step": v["ms"] / args.steps, "share_of_step_time": step_share(v["ms"])}
               for kd, v in kinds.items() if v["launches"]}}

    out = {
        "metric": f"STFT frames/sec through K-means+tokenize, n_mels={n_mels} vocab={vocab}",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": (f"{cfg['name']} x{world}: {n_tr}+{n_va} clips/GPU of {args.clip_seconds:g} s @22.05 kHz "
                         f"({(n_tr + n_va) * T} frames/GPU), n_mels={n_mels}, vocab_size={vocab}, niter={args.niter}, "
                         f"k-means batches of 10000 files on a {min(256 * vocab, 10000 * T)}-row subsample"),
            "frames_per_step": frames_per_step, "parallelism": f"dp{world}",
        },
        "stage_seconds": stage,
        "stage_seconds_note": ("stand-alone cost of each stage, from one extra step with a synchronisation between the stages"
                               + (f"; in the timed steps the log-mel of every k-means batch after the first and of the validation "
                                  f"clips runs on a background stream beside the training of the batches before it "
                                  f"({pipe.beside_clips} clips per launch), so the stages sum to more than ms_per_step"
                                  if pipe.overlaps_logmel(n_tr) else "")),
        "stage_seconds_per_rank": stage_all,
        "warmup_step_ms": warm_ms,
        "traced_step_ms": traced_ms,
        "verified": verified,
        "dense_floor": dense_floor,
        "hard_workload": None,
        "roofline": roofline,
        "comm": comm,
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
    # ---- a harder stream: noise-dominated clips (tones at a fifth of their amplitude under sigma 0.05-0.5) ------
    # The pruning and the filter feed on structure in the frames; `dense_floor` is the rate with none of it.  This
    # is a midpoint: the same step on clips that cluster poorly, one warm-up and one timed run, outside `value`.
    if not args.no_hard_workload:
        hw_tr = synth_clips(n_tr, L=L, seed=seed, first_clip=rank * n_tr, device=device, out=wave_tr, noisy=True)
        hw_va = synth_clips(n_va, L=L, seed=seed, first_clip=world * n_tr + rank * n_va, device=device, out=wave_va, noisy=True)
        pipe.run(hw_tr, hw_va)
        barrier()
        t0 = time.perf_counter()
        hres = pipe.run(hw_tr, hw_va)
        barrier()
        ht = time.perf_counter() - t0
        if dist is not None:
            tm = torch.tensor([ht], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            ht = float(tm.item())
        out["hard_workload"] = {"value": frames_per_step / ht, "unit": "frames/s", "ms_per_step": ht * 1e3,
                "note": "the same step on noise-dominated clips (synth_clips(noisy=True): tone amplitudes x 0.2, white noise "
                        "sigma ~ LogU(0.05, 0.5)); one warm-up and one timed run, outside the timed region of `value`"}
        del hres

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # Lloyd sweeps the GPU step makes per frame it processes: trainings x niter x subsample rows / frames (2.9 on the
        # configs[3] shard, 0.2 at configs[1]); the CPU sample trains with that many iterations over ITS frames, at least one
        n_trains = -(-n_tr // 10000)
        sub_rows = min(256 * vocab, min(n_tr, 10000) * T)
        per_frame = n_trains * args.niter * sub_rows / ((n_tr + n_va) * T)
        out["cpu_baseline"] = cpu_baseline(n_mels, vocab, L, hop, seed, args.cpu_clips, niter=max(1, round(per_frame)),
                                           gpu_note=f"the GPU step runs {n_trains * args.niter} Lloyd sweeps over {sub_rows} rows per "
                                                    f"{(n_tr + n_va) * T} frames, i.e. {per_frame:.1f} sweeps per frame")
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
