"""`Trainer` -- the base class `main.train` drives (reference protocol: summarizer/models/__init__.py:9-187; callers:
summarizer/main.py:25-35,65-66).  Public surface kept: `Trainer(hps, splits_file)`, `.reset()`, `.train(fold)`,
`.test(fold) -> (avg_corr, (avg_f, max_f))`, `.predict_dataset(path)`, `.save_best_weights(path)`, `.load_weights(path)`,
attributes `.model`, `.best_weights`, `.dataset`, `.dataset_name`, and the helper names subclasses/users call
(`_get_train_test_keys`, `_init_model`, `_eval_scores`, `_eval_summary`, `draw_gtscores`, `draw_scores`).

Built differently underneath (MI355X-first):
  * scoring is BATCHED: all requested videos go through ONE packed launch (`model.score_packed`), not one forward each;
  * every video's features / normalised target are uploaded once (pinned staging, async copy) and stay in HBM;
  * per-video evaluation metadata (annotator ranks, change points, ...) is read from the dataset once and cached;
  * the dataset is anything with the h5py mapping protocol (utils/datasets.py), HDF5 when h5py is installed.
"""
import os
from types import SimpleNamespace

import numpy as np
import torch

from .. import kernels
from ..utils import eval as ev
from ..utils import eval_native
from ..utils.datasets import open_dataset

_HBM_CACHE_LIMIT = 16 << 30      # bytes of features kept resident per trainer (a whole dataset is ~0.1 GB)


class Trainer:
    """Abstract class handling the training process"""

    # ------------------------------------------------------------------ construction / protocol
    def __init__(self, hps, splits_file):
        self.hps, self.log, self.splits_file = hps, hps.logger, splits_file
        self.dataset = open_dataset(hps.dataset_of_file[splits_file], "r")
        self.dataset_name = hps.dataset_name_of_file[splits_file]
        self.best_weights = None
        self._hbm, self._hbm_bytes, self._meta = {}, 0, {}

    def reset(self):
        """Fresh model for the next cross-validation fold; returns self (main.py chains `.reset().train(fold)`)."""
        self.model = self._init_model()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        if self.hps.use_cuda:
            self.model.cuda()
        return self

    def _get_train_test_keys(self, fold):
        self.fold = fold
        self.split = self.hps.splits_of_file[self.splits_file][fold]
        return list(self.split["train_keys"]), list(self.split["test_keys"])

    def _init_model(self):
        raise Exception("_init_model has not been implemented")

    def train(self, fold):
        raise Exception("train has not been implemented")

    def _device(self):
        return next(self.model.parameters()).device

    # ------------------------------------------------------------------ per-fold bookkeeping shared by every trainer
    @staticmethod
    def _fold_best():
        """What train(fold) hands back, as a mutable record: [best correlation, best mean F-score, best max F-score]."""
        return [-1.0, 0.0, 0.0]

    def _evaluate_epoch(self, fold, epoch, best):
        """End of a training epoch.  On the epochs the configuration asks for, the fold's test videos are scored and the three
        metrics go to the TensorBoard writer under the reference's tag scheme (`<dataset>/Fold_<n>/Test/...`, vasnet.py:222-236 --
        the tags are the protocol benchmark.py's readers expect).  `best` (see _fold_best) keeps the running maxima; whenever the
        correlation improves, `best_weights` is pointed at the live state_dict, as the reference does."""
        if epoch % self.hps.test_every_epochs != 0:
            return
        corr, (f_avg, f_max) = self.test(fold)
        self.model.train()
        stem = f"{self.dataset_name}/Fold_{fold+1}/Test/"
        for tag, value in (("Correlation", corr), ("F-score_avg", f_avg), ("F-score_max", f_max)):
            self.hps.writer.add_scalar(stem + tag, value, epoch)
        best[1], best[2] = max(best[1], f_avg), max(best[2], f_max)
        if corr > best[0]:
            best[0] = corr
            self.best_weights = self.model.state_dict()

    # ------------------------------------------------------------------ feature ingest
    def _video_on_device(self, key, dev, want_target=False):
        """(features (T,D), min-max normalised gtscore (T,) or None) as device tensors, uploaded once.  The reference
        re-reads the HDF5 file and re-uploads every video at every step (vasnet.py:194-205)."""
        slot = (key, str(dev))
        hit = self._hbm.get(slot)
        if hit is not None:
            return hit
        rec = self.dataset[key]
        host = torch.from_numpy(np.ascontiguousarray(rec["features"][...], dtype=np.float32))
        if str(dev).startswith("cuda"):
            host = host.pin_memory()
        feats = host.to(dev, non_blocking=True)
        target = None
        if "gtscore" in rec:
            g = torch.from_numpy(np.asarray(rec["gtscore"][...], dtype=np.float32)).reshape(-1)
            g = g - g.min()                                   # same normalisation as vasnet.py:201-202 / dsn.py:104-105
            target = (g / (g.max() - g.min())).to(dev, non_blocking=True)
        hit = (feats, target)
        if self._hbm_bytes + feats.numel() * 4 <= _HBM_CACHE_LIMIT:
            self._hbm[slot] = hit
            self._hbm_bytes += feats.numel() * 4
        return hit

    def _video_meta(self, key, need):
        """Evaluation-side fields of one video, read from the dataset once.  `need` = "scores" | "summary" selects which
        presence check applies (same exception text as the reference)."""
        m = self._meta.get(key)
        if m is None:
            m = self._meta[key] = SimpleNamespace(rec=self.dataset[key], loaded=set())
        if need not in m.loaded:
            rec = m.rec
            if "n_frames" not in m.__dict__:
                m.n_frames = rec["n_frames"][()]
                m.picks = rec["picks"][...]
            if need == "scores":
                if "user_scores" not in rec:
                    raise Exception(f"No /user_scores in video {key} for score evaluation, "
                                    "make sure you have up-to-date .h5 dataset files.")
                m.user_scores = rec["user_scores"][...]
                m.user_ranks = ev.rank_users(m.user_scores)       # constant per video: ranked once, not per evaluation
            else:
                if "change_points" not in rec:
                    raise Exception(f"No /change_points in video {key} for summary evaluation, "
                                    "make sure you have up-to-date .h5 dataset files.")
                m.cps = rec["change_points"][...]
                m.nfps = rec["n_frame_per_seg"][...].tolist()
                m.user_summary = rec["user_summary"][...]
            m.loaded.add(need)
        return m

    # ------------------------------------------------------------------ batched scoring (the hot path)
    def _score_keys(self, keys, max_frames_per_launch=1 << 17):
        """{key: (seq_len,) float32 numpy}.  Videos are packed back to back and scored in as few launches as the frame
        budget allows; models with positional embeddings go through the per-video interface (they index by position)."""
        dev, out = self._device(), {}
        if getattr(self.model, "max_length", None):
            for key in keys:
                feats = self._video_on_device(key, dev)[0]
                out[key] = self.model(feats.unsqueeze(1).clone()).squeeze().detach().cpu().numpy()
            kernels.health_check()
            return out
        groups, cur, frames = [], [], 0
        for key in keys:
            T = self._video_on_device(key, dev)[0].shape[0]
            if cur and frames + T > max_frames_per_launch:
                groups.append(cur); cur, frames = [], 0
            cur.append(key); frames += T
        if cur:
            groups.append(cur)
        for grp in groups:
            feats = [self._video_on_device(k, dev)[0] for k in grp]
            lens = [f.shape[0] for f in feats]
            packed = feats[0] if len(feats) == 1 else torch.cat(feats)
            flat = self.model.score_packed(packed, lens).detach().cpu().numpy()
            kernels.health_check()      # the D2H above synchronised: fail loudly if a persistent recurrence kernel timed out
            for k, piece in zip(grp, np.split(flat, np.cumsum(lens)[:-1])):
                out[k] = piece.copy() if piece.shape[0] > 1 else piece.reshape(())    # `.squeeze()` of the reference: T == 1 -> 0-d
        return out

    def test(self, fold):
        """Score the fold's test videos, then rank correlation and key-shot F-scores: (avg_corr, (avg_f, max_f)).
        The whole evaluation tail of all videos runs in ONE native, multi-threaded call (utils/eval_native.py); the
        per-video numpy methods below (`_eval_scores`, `_eval_summary`) compute the same numbers and remain available."""
        self.model.eval()
        test_keys = self._get_train_test_keys(fold)[1]
        with torch.no_grad():
            dev_res = self._test_on_device(test_keys)
            if dev_res is not None:          # scores never left HBM: upsample, segment means and Spearman ran on the device
                corr, f_avg, f_max = dev_res
                return np.mean(corr), (np.mean(f_avg), np.mean(f_max))
            activations = self._score_keys(test_keys)
        corr, f_avg, f_max, _ = self._evaluate_native(activations, test_keys)
        return np.mean(corr), (np.mean(f_avg), np.mean(f_max))

    def _test_on_device(self, keys, max_frames_per_launch=1 << 17):
        """Device-side evaluation tail: ONE packed scoring launch, then sumk_eval_device on the scores where they
        are; a single small D2H carries segment means + correlations to the host knapsack / F-score threads.  None when the batch does
        not qualify (positional embeddings, more frames than one launch takes, metadata the kernel does not cover): the caller then
        takes the host tail, which computes the same numbers."""
        if getattr(self.model, "max_length", None) or not keys or not hasattr(self.model, "score_packed"):
            return None
        dev = self._device()
        if dev.type != "cuda":
            return None
        metas = [self._native_meta(k) for k in keys]
        if not all(eval_native.device_ready(m) for m in metas):
            return None
        # the fold's test set packed once and kept in HBM (a second copy of its features: 49 MB for 50 TVSum videos of 288 GB): every later
        # call of the same key list starts at the scoring launch
        pc = self.__dict__.setdefault("_packed_test_sets", {})
        hit = pc.get((tuple(keys), str(dev)))
        if hit is None:
            feats = [self._video_on_device(k, dev)[0] for k in keys]
            lens = [f.shape[0] for f in feats]
            if sum(lens) > max_frames_per_launch:
                return None
            packed = feats[0] if len(feats) == 1 else torch.cat(feats)
            if len(pc) >= 2:
                pc.pop(next(iter(pc)))
            pc[(tuple(keys), str(dev))] = (packed, lens)
        else:
            packed, lens = hit
        scores = self.model.score_packed(packed, lens).detach().contiguous()
        corr, f_avg, f_max, _ = eval_native.evaluate_batch_device(metas, scores, lens, self.hps.summary_proportion, self.hps.selection_algorithm)
        kernels.health_check()      # the D2H inside synchronised: fail loudly if a persistent recurrence kernel timed out
        return corr, f_avg, f_max

    def _native_meta(self, key):
        m = self._video_meta(key, "scores")
        m = self._video_meta(key, "summary")
        if "native" not in m.__dict__:
            m.native = eval_native.prepare_video(m.n_frames, m.picks, m.cps, m.nfps, m.user_summary, m.user_ranks)
        return m.native

    def _evaluate_native(self, activations, keys, want_summaries=False):
        return eval_native.evaluate_batch([self._native_meta(k) for k in keys], [activations[k] for k in keys],
                                          self.hps.summary_proportion, self.hps.selection_algorithm, want_summaries)

    def _eval_scores(self, machine_summary_activations, test_keys):
        """Mean over videos of the mean Spearman correlation with each annotator's scores."""
        per_video = []
        for key in test_keys:
            m = self._video_meta(key, "scores")
            frame_scores = ev.generate_scores(machine_summary_activations[key], m.n_frames, m.picks)
            per_video.append(ev.evaluate_scores(frame_scores, m.user_scores, metric="spearmanr", user_ranks=m.user_ranks))
        return np.mean(per_video)

    def _machine_summary(self, key, activations):
        m = self._video_meta(key, "summary")
        return m, ev.generate_summary(activations, m.cps, m.n_frames, m.nfps, m.picks, self.hps.summary_proportion,
                                      self.hps.selection_algorithm)

    def _eval_summary(self, machine_summary_activations, test_keys):
        """Mean over videos of the (average, maximum) F-score of the generated key-shot summary against the annotators."""
        f = np.array([ev.evaluate_summary(summ, m.user_summary)
                      for m, summ in (self._machine_summary(k, machine_summary_activations[k]) for k in test_keys)])
        return np.mean(f[:, 0]), np.mean(f[:, 1])

    # ------------------------------------------------------------------ logging helpers
    def _histogram(self, tag, key, values):
        self.hps.writer.add_histogram(f"{self.dataset_name}/{tag}", values, int(key.split("_")[1]))

    def draw_gtscores(self, fold, keys, norm=True):
        """Ground-truth score distributions of the training videos as TensorBoard histograms."""
        for key in keys:
            gt = self.dataset[key]["gtscore"][...]
            if norm:
                gt = (gt - gt.min()) / (gt.max() - gt.min())
            self._histogram(f"Fold_{fold+1}/Train/gtscores", key, gt)

    def draw_scores(self, fold, dist_scores):
        """Predicted score distributions (last epoch) as TensorBoard histograms."""
        for key, scores in dist_scores.items():
            self._histogram(f"Fold_{fold+1}/Train/final_scores", key,
                            scores.detach().cpu().numpy() if torch.is_tensor(scores) else scores)

    # ------------------------------------------------------------------ predictions / checkpoints
    def predict_dataset(self, pred_path):
        """Scores + machine summary of EVERY video of the dataset with the best weights, written under
        `<dataset_file>/<key>/{scores,user_summary,machine_summary,machine_scores}` (HDF5, or an .npz with those paths)."""
        self.model.load_state_dict(self.best_weights)
        self.model.eval()
        keys = list(self.dataset.keys())
        with torch.no_grad():
            activations = self._score_keys(keys)
        root_name = os.path.basename(str(self.hps.dataset_of_file[self.splits_file]))
        summaries = dict(zip(keys, self._evaluate_native(activations, keys, want_summaries=True)[3]))   # all videos, one native call
        with open_dataset(pred_path, "w") as sink:
            root = sink.create_group(root_name)
            for key in keys:
                m, summary = self._video_meta(key, "summary"), summaries[key]
                grp = root.create_group(key)
                for name, value in (("scores", activations[key]), ("user_summary", m.user_summary),
                                    ("machine_summary", summary),
                                    ("machine_scores", ev.generate_scores(activations[key], m.n_frames, m.picks))):
                    grp.create_dataset(name, data=value)

    def save_best_weights(self, weights_path):
        if self.best_weights is None:
            raise Exception("best_weights property is empty, can't save model's weights")
        torch.save(self.best_weights, weights_path)

    def load_weights(self, weights_path):
        self.model.load_state_dict(torch.load(weights_path))
