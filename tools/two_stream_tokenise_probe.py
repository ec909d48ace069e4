"""Would tokenising on two streams pay?  (development aid)  Chunk i's exact sweep on a second
context and stream beside chunk i+1's guess generator, against the one-stream chain; results compared bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend, HipBackend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips

be = default_backend()
be2 = HipBackend(be.device)
k = 8192
clips = 10000
wave = synth_clips(clips, device="cuda")
frames = be.logmel(wave, frame_major=True, l2norm=True); del wave
n, d = frames.shape
km = Kmeans(d, k, niter=20, backend=be); km.train(frames[:4307500])
C = be.l2norm_rows(km.centroids_device)
cperm = be.from_host(be.group_rows_kd(be.to_host(C)))
means = be.group_means(C, cperm)
gnbr = be.group_neighbours(means, 4)
dmin = be.group_min_dist(C, cperm)
dmin2 = be2.group_min_dist(C, cperm)
chunk = 4307500 // 2
chunks = [(a, min(n, a + chunk)) for a in range(0, n, chunk)]


def one_stream():
    out = []
    for a, b in chunks:
        x = frames[a:b]
        g, gd = be.assign_coarse(x, C, cperm, means, gnbr)
        out.append(be.assign_pruned(x, C, be.visit_order(g, gd, k), cperm, dmin, want_dist=False)[0])
    return out


side = torch.cuda.Stream()


def two_streams():
    main = torch.cuda.current_stream()
    out = []
    for a, b in chunks:
        x = frames[a:b]
        g, gd = be.assign_coarse(x, C, cperm, means, gnbr)          # main stream, context 1
        od = be.visit_order(g, gd, k)
        ev = torch.cuda.Event(); ev.record(main)
        with torch.cuda.stream(side):                                # exact sweep: second context, second stream
            side.wait_event(ev)
            out.append(be2.assign_pruned(x, C, od, cperm, dmin2, want_dist=False)[0])
    main.wait_stream(side)
    return out


def t(fn, it=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3, r


ms1, r1 = t(one_stream)
ms2, r2 = t(two_streams)
same = all(torch.equal(a, b) for a, b in zip(r1, r2))
print(f"{n} rows in chunks of {chunk}: one stream {ms1:.2f} ms, two streams {ms2:.2f} ms; same tokens {same}")
bad = 0
for rep in range(10):
    r = two_streams(); torch.cuda.synchronize()
    bad += sum(int((a != b).sum()) for a, b in zip(r1, r))
print(f"10 more two-stream passes ({10 * n} rows): {bad} tokens differ from the one-stream result")
