"""The three stage classes through the reference's on-disk protocol, and the device pipeline, on the
GPU -- compared with the same stages run by the oracle on the same files."""
import json
import warnings
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture()
def workdir(tmp_path):
    from audio_tokens_amd.audio_tokens_config import AudioTokensConfig
    from audio_tokens_amd.synth import synth_clips
    ytids = [f"yt{i:03d}abcde" for i in range(14)]
    wave = synth_clips(len(ytids), L=22050 * 2, seed=4242, device="cpu").numpy()
    src = tmp_path / "audio"
    for y, w in zip(ytids, wave):
        p = src / "bal_train" / y[:2]
        p.mkdir(parents=True, exist_ok=True)
        np.save(p / f"{y}.npy", w)
    split = {"train": ytids[:11] + ["missing_clip"], "validation": ytids[11:]}
    (tmp_path / "out").mkdir()
    (tmp_path / "out" / "split.json").write_text(json.dumps(split))
    cfg = AudioTokensConfig(
        split_file=str(tmp_path / "out" / "split.json"), audio_source_path=str(src),
        dest_spec_path=tmp_path / "spectrograms", source_spec_path=tmp_path / "spectrograms",
        centroids_path=tmp_path / "out" / "centroids.npy", dest_tokenized_path=str(tmp_path / "tok"),
        vocab_size=32, niter=6, clustering_batch_size=6, tokenizer_batch_size=5, spectrogram_batch_size=4)
    return cfg, split, dict(zip(ytids, wave))


def test_run_pipeline_files(workdir, oracle, monkeypatch, tmp_path):
    from audio_tokens_amd import run_pipeline
    cfg, split, waves = workdir
    monkeypatch.chdir(tmp_path)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        run_pipeline.main(cfg)

    # stage 1: spectrograms/<split>/<ytid>.npy float32 [n_mels, T]; the missing clip is skipped
    for s in ("train", "validation"):
        files = sorted((Path(cfg.dest_spec_path) / s).glob("*.npy"))
        assert [f.stem for f in files] == sorted(y for y in split[s] if y in waves)
        for f in files:
            spec = np.load(f)
            assert spec.dtype == np.float32 and spec.shape == (64, 345)
            ref = oracle.logmel(waves[f.stem])
            P, Pr = 10.0 ** (spec.astype(np.float64) / 10), 10.0 ** (ref.astype(np.float64) / 10)
            assert (np.abs(P - Pr) <= 2e-5 * Pr + 1e-9 * Pr.max(0, keepdims=True) + 1e-14).all()

    # stage 2 on the files the GPU wrote: the oracle must reproduce centroids.npy bit for bit
    train_files = sorted((Path(cfg.dest_spec_path) / "train").glob("*.npy"))
    cent = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(0, len(train_files), cfg.clustering_batch_size):
            batch = np.concatenate([np.load(f).T for f in train_files[i:i + cfg.clustering_batch_size]], 0)
            batch = oracle.l2norm_rows(batch.astype(np.float32))
            cent = oracle.kmeans_train(batch, cfg.vocab_size, niter=cfg.niter, init_centroids=cent).centroids
    cent = oracle.l2norm_rows(cent)
    got = np.load(cfg.centroids_path)
    assert got.dtype == np.float32 and got.shape == (32, 64)
    assert np.array_equal(bits(got), bits(cent))
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-6)

    # stage 3: tokenized_audio/<split>/<ytid>.npy int64 [T], identical to the oracle's search
    for s in ("train", "validation"):
        for f in sorted((Path(cfg.dest_spec_path) / s).glob("*.npy")):
            tok = np.load(Path(cfg.dest_tokenized_path) / s / f.name)
            assert tok.dtype == np.int64 and tok.shape == (345,)
            ids, _ = oracle.assign(oracle.l2norm_rows(np.load(f).T.astype(np.float32)), cent)
            assert np.array_equal(tok, ids)


def test_stage_class_surface(workdir):
    from audio_tokens_amd.processors import SpectrogramGenerator
    cfg, split, waves = workdir
    sg = SpectrogramGenerator(cfg)
    y = split["train"][0]
    path = sg.find_audio_file(y)
    assert path is not None and sg.find_audio_file("nope") is None
    w = sg.preprocess_waveform(path)
    assert tuple(w.shape) == (1, 44100)
    spec = sg.generate_mel_spectrogram(w)
    assert tuple(spec.shape) == (64, 345)
    assert not sg.check_for_nan_inf(spec)
    assert sg.check_for_nan_inf(torch.tensor([float("nan")]))
    stereo = torch.stack([w[0], -w[0]])
    assert torch.equal(SpectrogramGenerator.convert_to_mono(stereo), torch.zeros(1, 44100))
    n = SpectrogramGenerator.normalize_spectrogram(spec)
    assert float(n.min()) == 0.0 and float(n.max()) == 1.0
    specs = sg.populate_specs(split["train"])
    assert [s["filename"] for s in specs] == [f"{y}.npy" for y in split["train"] if y in waves]


def test_generator_resamples_foreign_rates(workdir, oracle):
    """A 16-bit stereo 44.1 kHz .wav: decode -> mono -> device resampler -> log-mel, as the reference's
    preprocess_waveform + generate_mel_spectrogram (spectrogram_generator.py:105-126)."""
    import wave as wavmod
    from audio_tokens_amd.processors import SpectrogramGenerator
    cfg, split, waves = workdir
    rng = np.random.default_rng(5)
    pcm = (rng.standard_normal((44100, 2)) * 3000).astype("<i2")
    y = "zzwavclip01"
    p = Path(cfg.audio_source_path) / "bal_train" / y[:2]
    p.mkdir(parents=True, exist_ok=True)
    with wavmod.open(str(p / f"{y}.wav"), "wb") as f:
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(44100); f.writeframes(pcm.tobytes())
    sg = SpectrogramGenerator(cfg)
    w = sg.preprocess_waveform(sg.find_audio_file(y))
    assert tuple(w.shape) == (1, 22050) and w.device.type == "cuda"
    mono = (pcm.astype(np.float32) / 32768.0).mean(1, dtype=np.float32)
    want = oracle.resample(mono, 44100, 22050)
    np.testing.assert_allclose(w[0].cpu().numpy(), want, rtol=0, atol=2e-6)
    specs = sg.populate_specs([y, split["train"][0]])          # mixed host / device clips, two lengths
    assert [s["filename"] for s in specs] == [f"{y}.wav", f"{split['train'][0]}.npy"]
    ref = oracle.logmel(w[0].cpu().numpy())
    P, Pr = 10.0 ** (specs[0]["spec"].double().numpy() / 10), 10.0 ** (ref.astype(np.float64) / 10)
    assert (np.abs(P - Pr) <= 2e-5 * Pr + 1e-9 * Pr.max(0, keepdims=True) + 1e-14).all()


def test_device_pipeline_matches_oracle(be, oracle):
    from audio_tokens_amd.pipeline import DevicePipeline
    from audio_tokens_amd.synth import synth_clips
    wave = synth_clips(10, L=22050 * 3, seed=7, device="cuda")
    pipe = DevicePipeline(n_mels=64, vocab_size=64, niter=5, clustering_batch_size=4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = pipe.run(wave[:8], wave[8:])
        T = res.frames_per_clip
        frames = be.logmel(wave, frame_major=True, l2norm=True).cpu().numpy()
        cent = None
        for c0 in range(0, 8, 4):
            cent = oracle.kmeans_train(frames[c0 * T:(c0 + 4) * T], 64, niter=5, init_centroids=cent).centroids
    cent = oracle.l2norm_rows(cent)
    assert np.array_equal(bits(res.centroids.cpu().numpy()), bits(cent))
    ids, _ = oracle.assign(frames, cent)
    assert np.array_equal(res.tokens_train.cpu().numpy(), ids[:8 * T])
    assert np.array_equal(res.tokens_val.cpu().numpy(), ids[8 * T:])


@pytest.mark.parametrize("clips,seconds,k,batch,chunk,pinned", [(10, 3, 64, 4, 3, False), (96, 2, 1024, 40, 17, True)])
def test_streaming_pipeline_equals_resident(be, clips, seconds, k, batch, chunk, pinned):
    """Host-resident waveforms streamed through pinned staging buffers (configs[4]'s mode): same
    centroids and tokens, bit for bit, as the run with everything resident in HBM."""
    from audio_tokens_amd.pipeline import DevicePipeline
    from audio_tokens_amd.synth import synth_clips
    wave = synth_clips(clips, L=22050 * seconds, seed=11, device="cuda")
    n_val = max(2, clips // 8)
    pipe = DevicePipeline(n_mels=64, vocab_size=k, niter=4, clustering_batch_size=batch, spectrogram_batch_size=chunk + 2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = pipe.run(wave[:-n_val], wave[-n_val:])
        host = wave.cpu()
        if pinned:
            host = host.pin_memory()
        got = pipe.run_streaming(host[:-n_val], host[-n_val:], chunk_clips=chunk)
    assert torch.equal(got.centroids.view(torch.int32), ref.centroids.view(torch.int32))
    assert got.tokens_train.device.type == "cpu"
    assert torch.equal(got.tokens_train, ref.tokens_train.cpu())
    assert torch.equal(got.tokens_val, ref.tokens_val.cpu())
