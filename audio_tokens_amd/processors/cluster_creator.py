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
        # The next batch of files is read while this one trains (same batches, same order).  A batch crosses to
        # the device ONCE: convolution (if any), row normalisation and the training run there, and the warm start
        # takes the previous centroids where they already are.
        be = kmeans.backend
        for i, batch in enumerate(prefetch(self._frame_batches(self.config.clustering_batch_size))):
            batch = be._f32(batch)
            if self.config.use_convolution:
                batch = self._convolve_device(batch)
            batch = normalize_rows(batch, be)
            if i == 0:
                kmeans.train(batch)
            else:
                kmeans.train(batch, init_centroids=kmeans.centroids_device)

        centroids = be.to_host(normalize_rows(kmeans.centroids_device, be))
        kmeans.lend_grouping(centroids)   # (a tokeniser in this process starts from this grouping)
        self.logger.info(f"Centroids shape: {centroids.shape}")
        Path(self.config.centroids_path).parent.mkdir(parents=True, exist_ok=True)
        np.save(self.config.centroids_path, centroids)
        self.visualize_centroids(centroids)

    def normalize_vectors(self, vectors):
        """vectors / (||vectors||_2 + 1e-10) row-wise; same bits as the reference's numpy lines."""
        return normalize_rows(vectors)

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

    def apply_convolution(self, time_slice_batch):
        frames = torch.as_tensor(np.asarray(time_slice_batch), device=self.device).float()
        return self._convolve_device(frames).cpu().numpy()

    def _frame_batches(self, batch_size):
        """Host frame matrices [sum T, n_mels] float32, one per `batch_size` spectrogram files."""
        spec_dir = Path(self.config.source_spec_path) / "train"
        # the reference takes Path.glob's order (file-system dependent); sorted is one such order
        files = sorted(spec_dir.glob("*.npy"))
        for i in tqdm(range(0, len(files), batch_size)):
            yield np.concatenate([np.load(f).T for f in files[i:i + batch_size]], axis=0).astype(np.float32, copy=False)

    def _batch_generator(self, batch_size):
        """The reference's generator (cluster_creator.py:83-102): numpy batches, convolved if configured."""
        for frames in self._frame_batches(batch_size):
            yield self.apply_convolution(frames) if self.config.use_convolution else frames

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
