"""SpecTokenizer -- the reference's stage 3 (processors/spec_tokenizer.py:22-240 of
danavery/audio-tokens) on the MI355X nearest-centroid kernel.

Same constructor, methods and artefacts (tokenized_audio/{train,validation}/<stem>.npy: int64 [T]).
`faiss.IndexFlatL2` is audio_tokens_amd.ops.IndexFlatL2; the per-batch sequence (load, transpose,
concatenate, normalise rows, search(.,1), slice per file, save) is the reference's.  A batch
crosses to the device once (convolution, row normalisation and the search run there); the token
statistics at the end of the train split are computed on the device from a histogram accumulated while
tokenising, and the plots are drawn only if matplotlib is importable.
"""
import logging
import shutil
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
from tqdm import tqdm

from ..audio_tokens_config import AudioTokensConfig
from ..ops import IndexFlatL2, normalize_rows
from ..utils.prefetch import prefetch
from ..utils.set_seed import set_seed

logging.basicConfig(
    level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s"
)


class SpecTokenizer:
    def __init__(self, config: AudioTokensConfig):
        self.config = config
        set_seed(self.config.random_seed)
        self.logger = logging.getLogger()
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

        self.source_path = Path(self.config.source_spec_path)
        self.dest_tokenized_path = Path(self.config.dest_tokenized_path)
        self.centroid_path = Path(self.config.centroids_path)
        self.index = self.load_centroid_index()

        if self.config.use_convolution:
            self.conv = self.create_convolution_layer()
        self._hist = None                 # device histogram of the tokens written since setup_output_directory()
        self.return_token_lists = True    # process_batch / tokenize_directory return tokens.tolist() like the reference

    def run(self):
        for split in ["train", "validation"]:
            source_spec_dir = self.source_path / split
            tokenized_dir = self.dest_tokenized_path / split

            self.setup_output_directory(tokenized_dir)
            self.logger.info(f"Tokenizing {split} set: {source_spec_dir} --> {tokenized_dir}")
            # (run() needs no O(frames) Python list: the statistics come from the device histogram)
            self.return_token_lists = False
            try:
                self.tokenize_directory(source_spec_dir, tokenized_dir)
            finally:
                self.return_token_lists = True
            if split == "train":
                self.analyze_tokens()
                self.plot_token_distribution()

    def tokenize_directory(self, source_dir: Path, tokenized_dir: Path):
        all_tokens = []
        spec_files = sorted(source_dir.glob("*.npy"))  # reference: Path.glob order (unspecified)

        starts = range(0, len(spec_files), self.config.tokenizer_batch_size)

        def load_batches():  # the next batch of files is read while this one is searched
            for i in starts:
                batch_files = spec_files[i: i + self.config.tokenizer_batch_size]
                yield batch_files, [np.load(f).T for f in batch_files]

        for batch_files, batch_specs in tqdm(prefetch(load_batches()), total=len(starts)):
            batch_tokens = self.process_batch(batch_files, tokenized_dir, batch_specs)
            all_tokens.extend(batch_tokens)

        return all_tokens

    def process_batch(self, batch_files, tokenized_dir: Path, batch_specs=None):
        if batch_specs is None:  # (the reference's signature: load here)
            batch_specs = [np.load(spec_file).T for spec_file in batch_files]
        batch_data = np.concatenate(batch_specs, axis=0)

        # one host->device copy per batch; convolution, row normalisation and the search stay on the device
        be = self.index.backend
        processed_batch = None
        if batch_data.size > 0:
            processed_batch = be._f32(batch_data)
            if self.config.use_convolution:
                processed_batch = self._convolve_device(processed_batch)

        if processed_batch is not None and processed_batch.numel() > 0:
            processed_batch = normalize_rows(processed_batch, be)
            tokens_dev, _ = self.index.assign(processed_batch, want_dist=False)
            hist = be.token_histogram(tokens_dev, max(1, self.index.ntotal))
            self._hist = hist if self._hist is None else self._hist + hist
            tokens = be.to_host(tokens_dev)

            start = 0
            for spec_file, spec in zip(batch_files, batch_specs):
                end = start + len(spec)
                file_tokens = tokens[start:end]
                output_file = tokenized_dir / f"{spec_file.stem}.npy"
                np.save(output_file, file_tokens)
                start = end

            return tokens.tolist() if self.return_token_lists else []

        return []

    def _convolve_device(self, frames):
        """frames [n, n_mels] (device) -> [n, num_kernels * n_mels] (device): the module's Conv1d along the mel axis, feature
        index = mel * num_kernels + kernel, as the reference's transpose(1, 2).reshape lays it out -- one HIP kernel
        (at_conv1d_mel_f32) with the module's weights; the module itself only provides them (torch's RNG, as in the
        reference)."""
        from ..backend import default_backend
        be = default_backend()
        with torch.no_grad():
            return be.conv1d_mel(frames, self.conv.weight.detach(), self.conv.bias.detach() if self.conv.bias is not None else None,
                                 padding=int(self.conv.padding[0]))

    def apply_convolution(self, batch):
        if len(batch) == 0:
            self.logger.warning("Received empty batch for convolution")
            return None
        batch_tensor = torch.as_tensor(np.asarray(batch), device=self.device).float()
        return self._convolve_device(batch_tensor).cpu().numpy()

    @staticmethod
    def normalize_vectors(vectors):
        return normalize_rows(vectors)

    def setup_output_directory(self, tokenized_dir):
        shutil.rmtree(tokenized_dir, ignore_errors=True)
        tokenized_dir.mkdir(parents=True)
        self._hist = None

    def create_convolution_layer(self):
        return nn.Conv1d(
            in_channels=1,
            out_channels=self.config.num_kernels,
            kernel_size=self.config.kernel_size,
            padding=self.config.kernel_size // 2,
        ).to(self.device)

    def load_centroid_index(self):
        centroids = np.load(self.centroid_path)
        index = IndexFlatL2(centroids.shape[1])
        index.add(centroids)
        return index

    # ---- token statistics (reference: spec_tokenizer.py:129-240) ---------------------------------
    # The reference keeps every token of the split as a Python int, counts them with a Counter, sorts the counts
    # on the host and fits Zipf's law with scipy.  Here the histogram is accumulated on the device while the
    # batches are tokenised (process_batch), at_token_stats_f64 sorts it and computes the cumulative-share rank and
    # the log-log regression there, and only the k sorted counts come back -- for the prints and the optional plots.
    # Differences in what is REPORTED (never in any artefact): equal counts rank in ascending token id (the
    # reference: first-seen order), and the p-value / standard error of the fit, which the reference never prints,
    # are not computed.
    def token_statistics(self, all_tokens=None):
        """-> dict(total, unique, tokens, frequencies, top_80, slope, intercept, r_value, n_fit) of `all_tokens`
        (any int sequence / array / device tensor), or of the tokens seen since the last setup_output_directory()
        when None.  tokens / frequencies: numpy, most frequent first, tokens that occur only."""
        be = self.index.backend
        k = max(1, self.index.ntotal)
        if all_tokens is None:
            counts = self._hist if self._hist is not None else be.zeros((k,), torch.int64)
        elif isinstance(all_tokens, torch.Tensor):
            counts = be.token_histogram(all_tokens, k)
        else:
            counts = be.token_histogram(np.asarray(all_tokens, dtype=np.int64), k)
        sc, stok, st = be.token_stats(counts)
        st = be.to_host(st)
        unique = int(st[1])
        return dict(total=int(st[0]), unique=unique, tokens=be.to_host(stok)[:unique].astype(np.int64),
                    frequencies=be.to_host(sc)[:unique], top_80=int(st[2]), slope=float(st[3]), intercept=float(st[4]),
                    r_value=float(st[5]), n_fit=int(st[6]))

    def analyze_tokens(self, all_tokens=None):
        ts = self.token_statistics(all_tokens)
        self.logger.info(f"Total tokens: {ts['total']}")
        self.logger.info(f"Unique tokens: {ts['unique']}")
        if ts["unique"]:
            self.logger.info(f"Most common token: {[(int(ts['tokens'][0]), int(ts['frequencies'][0]))]}")
            self.logger.info(f"Least common token: {(int(ts['tokens'][-1]), int(ts['frequencies'][-1]))}")
        plt = self._pyplot()
        if plt is None or not ts["unique"]:
            return ts
        plt.figure(figsize=(12, 6))
        plt.bar(ts["tokens"], ts["frequencies"])
        plt.title("Distribution of Assigned Tokens")
        plt.xlabel("Token ID")
        plt.ylabel("Frequency")
        Path("output").mkdir(exist_ok=True)
        plt.savefig("output/token_distribution.png")
        plt.close()
        return ts

    def plot_token_distribution(self, all_tokens=None):
        ts = self.token_statistics(all_tokens)
        if not ts["unique"]:
            return ts
        tokens, freq = ts["tokens"], ts["frequencies"]
        plt = self._pyplot()
        if plt is not None:
            ranks = np.arange(1, len(freq) + 1)
            fig, (top, bottom) = plt.subplots(2, 1, figsize=(15, 10))
            top.loglog(ranks, freq)
            top.set_title("Distribution of Assigned Tokens (Sorted by Frequency)")
            top.set_xlabel("Token Rank")
            top.set_ylabel("Frequency")
            bottom.bar(ranks, freq)
            bottom.set_xlabel("Token Rank")
            bottom.set_ylabel("Frequency")
            fig.tight_layout()
            fig.savefig("correct_token_distribution.png")
            plt.close(fig)
        print(f"Total unique tokens: {ts['unique']}")
        print(f"Total token occurrences: {ts['total']}")
        print(f"Most common token (rank 1): Token {tokens[0]} (used {freq[0]} times)")
        print(f"Least common token (rank {len(tokens)}): Token {tokens[-1]} (used {freq[-1]} times)")
        print(f"Top {ts['top_80'] + 1} tokens account for 80% of all token occurrences")
        print(f"Frequency ratio between most and least common: {freq[0] / freq[-1]:.2f}")
        self.analyze_zipf_and_tail(freq, _stats=ts)
        return ts

    def analyze_zipf_and_tail(self, frequencies, _stats=None):
        """frequencies: counts sorted descending (the reference's argument).  The fit and the tail are computed on
        the device from them (a histogram that is already sorted stays as it is)."""
        ts = _stats
        if ts is None:
            be = self.index.backend
            f = np.ascontiguousarray(np.asarray(frequencies, dtype=np.int64))
            sc, _, st = be.token_stats(be.from_host(f)) if f.size else (None, None, None)
            if st is None:
                return None
            st = be.to_host(st)
            ts = dict(unique=int(st[1]), frequencies=be.to_host(sc)[:int(st[1])], top_80=int(st[2]), slope=float(st[3]),
                      intercept=float(st[4]), r_value=float(st[5]), n_fit=int(st[6]))
        if ts["n_fit"] < 2:
            return ts
        plt = self._pyplot()
        if plt is not None:
            lr = np.log(np.arange(1, ts["unique"] + 1))
            plt.figure(figsize=(12, 8))
            plt.scatter(lr, np.log(ts["frequencies"]), alpha=0.5, label="Observed")
            plt.plot(lr, ts["intercept"] + ts["slope"] * lr, color="red", label=f"Fitted (slope = {ts['slope']:.2f})")
            plt.xlabel("Log Rank")
            plt.ylabel("Log Frequency")
            plt.title("Zipf's Law Analysis")
            plt.legend()
            plt.savefig("zipf_law_analysis.png")
            plt.close()
        tail_start = ts["top_80"]
        ts["tail_proportion"] = 1 - (tail_start / ts["unique"])
        print(f"Zipf's law slope: {ts['slope']:.2f} (closer to -1 indicates closer fit to Zipf's law)")
        print(f"R-squared value: {ts['r_value'] ** 2:.2f}")
        print(f"Proportion of tokens in the tail (last 20% of occurrences): {ts['tail_proportion']:.2%}")
        print(f"Number of tokens accounting for 80% of occurrences: {tail_start}")
        return ts

    @staticmethod
    def _pyplot():
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            return plt
        except Exception:
            return None


if __name__ == "__main__":
    SpecTokenizer(AudioTokensConfig()).run()
