"""Trainer base class -- mirror of `summarizer/models/__init__.py` (reference): same constructor, same methods, same
return values, same exceptions, so `main.train` (summarizer/main.py:25-35,65-66) drives it unchanged.

What differs underneath: `test()` / `predict_dataset()` score ALL requested videos in ONE packed launch on the GPU
(`model.score_packed`) instead of one forward per video, and the dataset may be any object with the h5py mapping
protocol (utils/datasets.py).  Evaluation (rank correlation, knapsack key-shots, F-score) stays on the host like
the reference (utils/eval.py)."""
import os
import numpy as np
import torch

from ..utils.eval import generate_summary, evaluate_summary, generate_scores, evaluate_scores, rank_users
from ..utils.datasets import open_dataset


class Trainer:
    """Abstract class handling the training process"""
    def __init__(self, hps, splits_file):
        self.hps = hps
        self.log = hps.logger
        self.splits_file = splits_file
        self.dataset = open_dataset(hps.dataset_of_file[splits_file], "r")
        self.dataset_name = hps.dataset_name_of_file[splits_file]
        self.best_weights = None
        self._dev_cache = {}      # key -> (features, normalised gtscore) resident in HBM (datasets are ~100 MB)
        self._dev_cache_bytes = 0
        self._rank_cache = {}     # key -> annotator ranks (constant per video)

    def reset(self):
        """Reset between two folds of the cross-validation"""
        self.model = self._init_model()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        if self.hps.use_cuda:
            self.model.cuda()
        return self

    def _get_train_test_keys(self, fold):
        """Train/Test keys from current split file and fold"""
        self.fold = fold
        self.split = self.hps.splits_of_file[self.splits_file][fold]
        return self.split["train_keys"][:], self.split["test_keys"][:]

    def _init_model(self):
        """Initialize here your model"""
        raise Exception("_init_model has not been implemented")

    def train(self, fold):
        """Train model on train_keys"""
        raise Exception("train has not been implemented")

    # ------------------------------------------------------------------ feature ingest
    def _device(self):
        return next(self.model.parameters()).device

    def _video_on_device(self, key, dev, want_target=False):
        """(features (T,D), min-max normalised gtscore (T,) or None) as device tensors.  The reference re-reads the HDF5
        file and re-uploads every video at every step (vasnet.py:194-205); a whole dataset is ~100 MB, so each video is
        uploaded ONCE (pinned staging buffer, async copy) and stays in HBM (cache capped at 16 GiB)."""
        hit = self._dev_cache.get((key, str(dev)))
        if hit is None:
            d = self.dataset[key]
            f = torch.from_numpy(np.ascontiguousarray(d["features"][...], dtype=np.float32))
            if torch.cuda.is_available() and str(dev).startswith("cuda"):
                f = f.pin_memory()
            feats = f.to(dev, non_blocking=True)
            target = None
            if "gtscore" in d:
                t = torch.from_numpy(np.asarray(d["gtscore"][...], dtype=np.float32)).view(-1)
                t = t - t.min()                              # vasnet.py:201-202 / dsn.py:104-105
                t = t / (t.max() - t.min())
                target = t.to(dev, non_blocking=True)
            hit = (feats, target)
            nbytes = feats.numel() * 4
            if self._dev_cache_bytes + nbytes <= (16 << 30):
                self._dev_cache[(key, str(dev))] = hit
                self._dev_cache_bytes += nbytes
        return hit

    # ------------------------------------------------------------------ batched scoring (the hot path)

    def _score_keys(self, keys, max_frames_per_launch=1 << 17):
        """{key: (seq_len,) float32 numpy} for `keys`, scoring many videos per launch.  Models with positional
        embeddings (VASNet max_pos) go through the per-video reference interface."""
        dev = self._device()
        out = {}
        if getattr(self.model, "max_length", None):
            for key in keys:
                seq = torch.from_numpy(self.dataset[key]["features"][...]).unsqueeze(1).to(dev)
                out[key] = self.model(seq).squeeze().detach().cpu().numpy()
            return out
        batch, frames = [], 0
        def flush():
            nonlocal batch, frames
            if not batch:
                return
            feats = [self._video_on_device(k, dev)[0] for k in batch]
            lens = [f.shape[0] for f in feats]
            x = torch.cat(feats) if len(feats) > 1 else feats[0]
            s = self.model.score_packed(x, lens).detach().cpu().numpy()
            off = np.concatenate([[0], np.cumsum(lens)])
            for i, k in enumerate(batch):
                out[k] = s[off[i]:off[i + 1]].copy() if lens[i] > 1 else s[off[i]:off[i + 1]].reshape(())   # .squeeze() quirk: T==1 -> 0-d
            batch, frames = [], 0
        for key in keys:
            T = self.dataset[key]["features"].shape[0]
            if frames + T > max_frames_per_launch:
                flush()
            batch.append(key); frames += T
        flush()
        return out

    def test(self, fold):
        """Test model on test_keys"""
        self.model.eval()
        _, test_keys = self._get_train_test_keys(fold)
        with torch.no_grad():
            summary = self._score_keys(test_keys)
        avg_corr = self._eval_scores(summary, test_keys)
        avg_f_score, max_f_score = self._eval_summary(summary, test_keys)
        return avg_corr, (avg_f_score, max_f_score)

    def _eval_scores(self, machine_summary_activations, test_keys):
        """Average (over test keys) of the mean Spearman correlation with each annotator."""
        avg_corrs = []
        for key in test_keys:
            d = self.dataset[key]
            probs = machine_summary_activations[key]
            if "user_scores" not in d:
                raise Exception(f"No /user_scores in video {key} for score evaluation, "
                                "make sure you have up-to-date .h5 dataset files.")
            user_scores = d["user_scores"][...]
            n_frames = d["n_frames"][()]
            positions = d["picks"][...]
            machine_scores = generate_scores(probs, n_frames, positions)
            ranks = self._rank_cache.get(key)
            if ranks is None:
                ranks = self._rank_cache[key] = rank_users(user_scores)
            avg_corrs.append(evaluate_scores(machine_scores, user_scores, metric="spearmanr", user_ranks=ranks))
        return np.mean(avg_corrs)

    def _eval_summary(self, machine_summary_activations, test_keys):
        """Average over test keys of the (avg, max) F-score of the generated key-shot summary."""
        avg_f_scores, max_f_scores = [], []
        for key in test_keys:
            d = self.dataset[key]
            probs = machine_summary_activations[key]
            if "change_points" not in d:
                raise Exception(f"No /change_points in video {key} for summary evaluation, "
                                "make sure you have up-to-date .h5 dataset files.")
            cps = d["change_points"][...]
            num_frames = d["n_frames"][()]
            nfps = d["n_frame_per_seg"][...].tolist()
            positions = d["picks"][...]
            user_summary = d["user_summary"][...]
            machine_summary = generate_summary(probs, cps, num_frames, nfps, positions, self.hps.summary_proportion,
                                               self.hps.selection_algorithm)
            avg_f_score, max_f_score = evaluate_summary(machine_summary, user_summary)
            avg_f_scores.append(avg_f_score)
            max_f_scores.append(max_f_score)
        return np.mean(avg_f_scores), np.mean(max_f_scores)

    def draw_gtscores(self, fold, keys, norm=True):
        """Draw datasets ground truth scores distribution in Tensorboard histograms"""
        for key in keys:
            d = self.dataset[key]
            i = int(key.split("_")[1])
            gtscore = d["gtscore"][...]
            if norm:
                gtscore -= gtscore.min()
                gtscore /= gtscore.max() - gtscore.min()
            self.hps.writer.add_histogram(f"{self.dataset_name}/Fold_{fold+1}/Train/gtscores", gtscore, i)

    def draw_scores(self, fold, dist_scores):
        """Draw predicted scores distribution in Tensorboard histograms"""
        for key, scores in dist_scores.items():
            i = int(key.split("_")[1])
            if torch.is_tensor(scores):
                scores = scores.detach().cpu().numpy()
            self.hps.writer.add_histogram(f"{self.dataset_name}/Fold_{fold+1}/Train/final_scores", scores, i)

    def predict_dataset(self, pred_path):
        """Predict on all videos in the dataset and save them (HDF5 when h5py is present, else the same groups in an .npz)"""
        self.model.load_state_dict(self.best_weights)
        self.model.eval()
        keys = list(self.dataset.keys())
        with torch.no_grad():
            all_scores = self._score_keys(keys)
        with open_dataset(pred_path, "w") as f:
            dataset_file = os.path.basename(str(self.hps.dataset_of_file[self.splits_file]))
            g = f.create_group(dataset_file)
            for key in keys:
                d = self.dataset[key]
                cps = d["change_points"][...]
                n_frames = d["n_frames"][()]
                nfps = d["n_frame_per_seg"][...].tolist()
                positions = d["picks"][...]
                user_summary = d["user_summary"][...]
                scores = all_scores[key]
                machine_summary = generate_summary(scores, cps, n_frames, nfps, positions, self.hps.summary_proportion,
                                                   self.hps.selection_algorithm)
                machine_scores = generate_scores(scores, n_frames, positions)
                k = g.create_group(key)
                k.create_dataset("scores", data=scores)
                k.create_dataset("user_summary", data=user_summary)
                k.create_dataset("machine_summary", data=machine_summary)
                k.create_dataset("machine_scores", data=machine_scores)

    def save_best_weights(self, weights_path):
        """Dump current best weights"""
        if self.best_weights is None:
            raise Exception("best_weights property is empty, can't save model's weights")
        torch.save(self.best_weights, weights_path)

    def load_weights(self, weights_path):
        """Load weights"""
        self.model.load_state_dict(torch.load(weights_path))
