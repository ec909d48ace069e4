"""SpecTokenizer -- the reference's stage 3 (processors/spec_tokenizer.py:22-240 of
danavery/audio-tokens) on the MI355X nearest-centroid kernel.

Same constructor, methods and artefacts (tokenized_audio/{train,validation}/<stem>.npy: int64 [T]).
`faiss.IndexFlatL2` is audio_tokens_amd.ops.IndexFlatL2; the per-batch sequence (load, transpose,
concatenate, normalise rows, search(.,1), slice per file, save) is the reference's.  The token
statistics / plots at the end of the train split are reporting: the counts are computed on the
device and the plots are drawn only if matplotlib is importable.
"""
import logging
import shutil
from collections import Counter
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

    def run(self):
        for split in ["train", "validation"]:
            source_spec_dir = self.source_path / split
            tokenized_dir = self.dest_tokenized_path / split

            self.setup_output_directory(tokenized_dir)
            self.logger.info(f"Tokenizing {split} set: {source_spec_dir} --> {tokenized_dir}")
            all_tokens = self.tokenize_directory(source_spec_dir, tokenized_dir)
            if split == "train":
                self.analyze_tokens(all_tokens)
                self.plot_token_distribution(all_tokens)

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

        if self.config.use_convolution:
            processed_batch = self.apply_convolution(batch_data)
        else:
            processed_batch = batch_data.astype(np.float32)

        if processed_batch is not None and processed_batch.size > 0:
            processed_batch = self.normalize_vectors(processed_batch)
            _, tokens = self.index.search(processed_batch, 1)
            tokens = np.squeeze(tokens, 1)

            start = 0
            for spec_file, spec in zip(batch_files, batch_specs):
                end = start + len(spec)
                file_tokens = tokens[start:end]
                output_file = tokenized_dir / f"{spec_file.stem}.npy"
                np.save(output_file, file_tokens)
                start = end

            return tokens.tolist()

        return []

    def apply_convolution(self, batch):
        if len(batch) == 0:
            self.logger.warning("Received empty batch for convolution")
            return None
        batch_tensor = torch.tensor(batch, device=self.device).float().unsqueeze(1)
        conv_output = self.conv(batch_tensor)
        return (
            conv_output.transpose(1, 2)
            .reshape(-1, self.config.num_kernels * self.config.n_mels)
            .cpu()
            .detach()
            .numpy()
        )

    @staticmethod
    def normalize_vectors(vectors):
        return normalize_rows(vectors)

    def setup_output_directory(self, tokenized_dir):
        shutil.rmtree(tokenized_dir, ignore_errors=True)
        tokenized_dir.mkdir(parents=True)

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

    # ---- reporting (host side, outside the accelerated path) --------------------------------
    def _token_counts(self, all_tokens):
        """Counter(all_tokens); long token lists are counted on the device (at_token_histogram_i64)
        instead of in a Python loop.  Ties then come in ascending token id, not first-seen order."""
        if len(all_tokens) < 1_000_000:
            return Counter(all_tokens)
        be = self.index.backend
        counts = be.to_host(be.token_histogram(np.asarray(all_tokens, dtype=np.int64), self.index.ntotal))
        return Counter({int(t): int(c) for t, c in enumerate(counts) if c})

    def analyze_tokens(self, all_tokens):
        token_counts = self._token_counts(all_tokens)
        self.logger.info(f"Total tokens: {len(all_tokens)}")
        self.logger.info(f"Unique tokens: {len(token_counts)}")
        if token_counts:
            self.logger.info(f"Most common token: {token_counts.most_common(1)}")
            self.logger.info(f"Least common token: {token_counts.most_common()[-1]}")
        plt = self._pyplot()
        if plt is None or not token_counts:
            return
        plt.figure(figsize=(12, 6))
        plt.bar(token_counts.keys(), token_counts.values())
        plt.title("Distribution of Assigned Tokens")
        plt.xlabel("Token ID")
        plt.ylabel("Frequency")
        Path("output").mkdir(exist_ok=True)
        plt.savefig("output/token_distribution.png")
        plt.close()

    def plot_token_distribution(self, all_tokens):
        token_counts = self._token_counts(all_tokens)
        if not token_counts:
            return
        sorted_counts = sorted(token_counts.items(), key=lambda x: x[1], reverse=True)
        tokens, frequencies = zip(*sorted_counts)
        ranks = range(1, len(tokens) + 1)

        plt = self._pyplot()
        if plt is not None:
            plt.figure(figsize=(15, 10))
            plt.subplot(2, 1, 1)
            plt.plot(ranks, frequencies)
            plt.title("Distribution of Assigned Tokens (Sorted by Frequency)")
            plt.xlabel("Token Rank")
            plt.ylabel("Frequency")
            plt.yscale("log")
            plt.xscale("log")
            plt.subplot(2, 1, 2)
            plt.bar(ranks, frequencies)
            plt.xlabel("Token Rank")
            plt.ylabel("Frequency")
            plt.tight_layout()
            plt.savefig("correct_token_distribution.png")
            plt.close()

        total_tokens = sum(frequencies)
        cumulative_freq = np.cumsum(frequencies) / total_tokens
        top_80_percent = np.searchsorted(cumulative_freq, 0.8) + 1
        print(f"Total unique tokens: {len(tokens)}")
        print(f"Total token occurrences: {total_tokens}")
        print(f"Most common token (rank 1): Token {tokens[0]} (used {frequencies[0]} times)")
        print(f"Least common token (rank {len(tokens)}): Token {tokens[-1]} (used {frequencies[-1]} times)")
        print(f"Top {top_80_percent} tokens account for 80% of all token occurrences")
        print(f"Frequency ratio between most and least common: {frequencies[0] / frequencies[-1]:.2f}")
        self.analyze_zipf_and_tail(frequencies)

    def analyze_zipf_and_tail(self, frequencies):
        from scipy import stats
        ranks = np.arange(1, len(frequencies) + 1)
        log_ranks = np.log(ranks)
        log_frequencies = np.log(frequencies)
        start_fit = int(0.1 * len(frequencies))
        end_fit = int(0.9 * len(frequencies))
        if end_fit - start_fit < 2:
            return
        slope, intercept, r_value, p_value, std_err = stats.linregress(
            log_ranks[start_fit:end_fit], log_frequencies[start_fit:end_fit]
        )
        plt = self._pyplot()
        if plt is not None:
            plt.figure(figsize=(12, 8))
            plt.scatter(log_ranks, log_frequencies, alpha=0.5, label="Observed")
            plt.plot(log_ranks, intercept + slope * log_ranks, color="red", label=f"Fitted (slope = {slope:.2f})")
            plt.xlabel("Log Rank")
            plt.ylabel("Log Frequency")
            plt.title("Zipf's Law Analysis")
            plt.legend()
            plt.savefig("zipf_law_analysis.png")
            plt.close()

        total_occurrences = sum(frequencies)
        cumulative_freq = np.cumsum(frequencies) / total_occurrences
        tail_start = np.searchsorted(cumulative_freq, 0.8)
        tail_proportion = 1 - (tail_start / len(frequencies))
        print(f"Zipf's law slope: {slope:.2f} (closer to -1 indicates closer fit to Zipf's law)")
        print(f"R-squared value: {r_value**2:.2f}")
        print(f"Proportion of tokens in the tail (last 20% of occurrences): {tail_proportion:.2%}")
        print(f"Number of tokens accounting for 80% of occurrences: {tail_start}")

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
