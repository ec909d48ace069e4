"""ClusterCreator -- the reference's stage 2 (processors/cluster_creator.py:21-117 of
danavery/audio-tokens) on the MI355X k-means.

Same constructor, methods and artefact (centroids.npy: float32 [vocab_size, d], unit-norm rows).
run() is the reference's sequence verbatim: for every batch of `clustering_batch_size` spectrogram
files build the frame-major matrix, L2-normalise its rows, `kmeans.train(batch)` on the first batch
and `kmeans.train(batch, init_centroids=kmeans.centroids)` afterwards, normalise the final
centroids, save, visualise.  `faiss.Kmeans` is audio_tokens_amd.ops.Kmeans.
"""
import logging
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
from tqdm import tqdm

from ..audio_tokens_config import AudioTokensConfig
from ..ops import Kmeans, normalize_rows
from ..utils.prefetch import prefetch
from ..utils.set_seed import set_seed

logging.basicConfig(
    level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s"
)


class ClusterCreator:
    def __init__(self, config):
        self.logger = logging.getLogger(__name__)
        self.config = config
        set_seed(self.config.random_seed)
        self.gpu = True
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if self.config.use_convolution:
            self.conv = nn.Conv1d(
                in_channels=1,
                out_channels=self.config.num_kernels,
                kernel_size=self.config.kernel_size,
                padding=self.config.kernel_size // 2,
            ).to(self.device)

    def run(self):
        n_freq_bins = self.config.n_mels
        if self.config.use_convolution:
            n_freq_bins *= self.config.num_kernels

        self.logger.info("starting clustering")
        kmeans = Kmeans(
            n_freq_bins,
            self.config.vocab_size,
            niter=self.config.niter,
            verbose=True,
            gpu=self.gpu,
        )
        # the next batch of files is read while this one trains (same batches, same order)
        for i, batch in enumerate(prefetch(self._batch_generator(self.config.clustering_batch_size))):
            batch = self.normalize_vectors(batch)
            if i == 0:
                kmeans.train(batch)
            else:
                kmeans.train(batch, init_centroids=kmeans.centroids)

        centroids = kmeans.centroids
        centroids = self.normalize_vectors(centroids)
        kmeans.lend_grouping(centroids)   # (a tokeniser in this process starts from this grouping)
        self.logger.info(f"Centroids shape: {centroids.shape}")
        Path(self.config.centroids_path).parent.mkdir(parents=True, exist_ok=True)
        np.save(self.config.centroids_path, centroids)
        self.visualize_centroids(centroids)

    def normalize_vectors(self, vectors):
        """vectors / (||vectors||_2 + 1e-10) row-wise; same bits as the reference's numpy lines."""
        return normalize_rows(vectors)

    def apply_convolution(self, time_slice_batch):
        time_slice_batch = np.array(time_slice_batch)
        time_slice_batch = torch.tensor(time_slice_batch, device=self.device).float().unsqueeze(1)
        conv_output = self.conv(time_slice_batch)
        return (
            conv_output.transpose(1, 2)
            .reshape(-1, self.config.num_kernels * self.config.n_mels)
            .cpu()
            .detach()
            .numpy()
        )

    def _batch_generator(self, batch_size):
        spec_dir = Path(self.config.source_spec_path) / "train"
        # the reference takes Path.glob's order (file-system dependent); sorted is one such order
        files = sorted(spec_dir.glob("*.npy"))

        for i in tqdm(range(0, len(files), batch_size)):
            batch_files = files[i:i + batch_size]
            batch_data = []
            for file in batch_files:
                spec = np.load(file)
                batch_data.append(spec.T)
            all_time_slices = np.concatenate(batch_data, axis=0)
            if self.config.use_convolution:
                yield self.apply_convolution(all_time_slices)
            else:
                yield all_time_slices.astype(np.float32)

    def visualize_centroids(self, centroids):
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            from sklearn.decomposition import PCA
        except Exception as e:  # plotting is reporting, not part of the hot path
            self.logger.info(f"Centroids visualization skipped ({e})")
            return
        centroids_2d = PCA(n_components=2).fit_transform(centroids)
        plt.figure(figsize=(10, 8))
        plt.scatter(centroids_2d[:, 0], centroids_2d[:, 1])
        plt.title("2D PCA of Centroids")
        Path("output").mkdir(exist_ok=True)
        plt.savefig("output/centroids_visualization.png")
        plt.close()
        self.logger.info("Centroids visualization saved")

    def evaluate_clustering(self, data, labels):
        from sklearn.metrics import silhouette_score
        score = silhouette_score(data, labels, sample_size=10000)
        self.logger.info(f"Silhouette Score: {score}")


if __name__ == "__main__":
    set_seed()
    ClusterCreator(AudioTokensConfig()).run()
