"""Drop-in ``Trainer`` for recipes/dns_interspeech_2020/fullsubnet/trainer.py on top of
audio_zen/trainer/base_trainer.py: same constructor signature and the same protected surface -

  ``_train_epoch(epoch)``            fullsubnet/trainer.py:33-76
  ``_validation_epoch(epoch) -> float``   fullsubnet/trainer.py:78-181 (enhancement + loss on the HIP path)
  ``_save_checkpoint / _resume_checkpoint / _preload_model / _is_best_epoch``   base_trainer.py:138-257
  ``train()``                        base_trainer.py:372-420

with the data-parallel strategy of base_trainer.py:32 (DistributedDataParallel over the "nccl" = RCCL backend;
the custom autograd functions of fullsubnet_amd/train.py are ordinary graph nodes, so DDP's bucketed gradient
all-reduce works unchanged).

Checkpoints are the reference's ``latest_model.tar`` / ``best_model.tar`` / ``model_<epoch>.pth`` files with the
reference's keys (epoch, best_score, optimizer, scaler, model), so either side can resume from the other's files
and ``BaseInferencer._load_model`` (base_inferencer.py:146-160) accepts them.

Arithmetic: ``meta.use_amp = true`` (every shipped train TOML) is the reference's autocast step - 16-bit matrix-core
operands with fp32 accumulation on the sub-band kernels (``meta.amp_dtype``: "f16" default, "bf16") under the
reference's own ``torch.amp.GradScaler`` (same ``scaler`` entry in the checkpoint); ``use_amp = false`` is fp32.

Validation metrics: STOI / WB_PESQ live in third-party packages (pystoi, pesq) that are outside this path and not
in this image.  ``_validation_epoch`` scores with the metrics registered in ``self.metrics`` (name -> fn(reference,
estimate)); SI_SDR (audio_zen/metrics.py:6-31) is built in, STOI / WB_PESQ are registered when their packages
import.  With both present the returned score is the reference's (STOI + transformed WB_PESQ) / 2 on the
"With_reverb" set (base_trainer.py:362-369, fullsubnet/trainer.py:181); otherwise it is the mean SI_SDR of the
enhanced validation utterances (higher is better as well) and ``self.validation_score_kind`` says so.
TensorBoard and the spectrogram plots of the reference are host-side tooling and not reproduced; the scalars
they would receive are kept in ``self.history``."""
from pathlib import Path

import numpy as np
import torch

from .acoustics.feature import istft, stft
from .acoustics.mask import build_complex_ideal_ratio_mask, decompress_cIRM
from . import _lib
from .train import mse_loss, train_step


def si_sdr(reference, estimation, sr=16000):
    """audio_zen/metrics.py:6-31."""
    estimation, reference = np.broadcast_arrays(estimation, reference)
    reference_energy = np.sum(reference ** 2, axis=-1, keepdims=True)
    optimal_scaling = np.sum(reference * estimation, axis=-1, keepdims=True) / reference_energy
    projection = optimal_scaling * reference
    noise = estimation - projection
    return 10 * np.log10(np.sum(projection ** 2, axis=-1) / np.sum(noise ** 2, axis=-1))


def _default_metrics():
    m = {"SI_SDR": si_sdr}
    try:  # audio_zen/metrics.py:34-44; packages outside this path, absent from this image
        from pystoi.stoi import stoi
        m["STOI"] = lambda ref, est, sr=16000: stoi(ref, est, sr, extended=False)
    except Exception:
        pass
    try:
        from pesq import pesq
        m["WB_PESQ"] = lambda ref, est, sr=16000: pesq(sr, ref, est, "wb")
    except Exception:
        pass
    return m


def transform_pesq_range(pesq_score):
    """audio_zen/acoustics/utils.py: map [-0.5, 4.5] to [0, 1]."""
    return (pesq_score + 0.5) / 5


class Trainer:
    def __init__(self, dist, rank, config, resume, only_validation, model, loss_function, optimizer,
                 train_dataloader, validation_dataloader=None):
        self.dist = dist
        self.rank = rank
        self.config = config
        self.device = torch.device("cuda", rank)
        model = model.to(self.device)
        self._distributed = (dist is not None and dist.is_available() and dist.is_initialized()
                             and dist.get_world_size() > 1)
        if self._distributed:
            model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[rank])  # base_trainer.py:32
        self.model = model
        self.optimizer = optimizer
        self.loss_function = loss_function
        self.train_dataloader = train_dataloader
        self.valid_dataloader = validation_dataloader

        meta = config.get("meta", {})
        # meta.use_amp = true (every shipped TOML) = fp16 autocast + GradScaler (fullsubnet/trainer.py:56,63-69,
        # base_trainer.py:63): the LSTM products of the sub-band kernels take 16-bit operands with fp32 accumulation
        # (Model.train_arithmetic; meta.amp_dtype = "bf16" selects bfloat16 operands instead of the recipe's fp16), the
        # transforms and the cIRM target stay fp32 as they are outside the autocast context there, and the reference's
        # own GradScaler scales the loss, skips overflowed steps and adapts the scale.
        self.use_amp = bool(meta.get("use_amp", False))
        self.scaler = torch.amp.GradScaler("cuda", enabled=self.use_amp)
        amp_dtype = str(meta.get("amp_dtype", "f16")).lower()
        aliases = {"f16": "f16", "fp16": "f16", "float16": "f16", "half": "f16", "torch.float16": "f16",
                   "bf16": "bf16", "bfloat16": "bf16", "torch.bfloat16": "bf16"}
        if self.use_amp and amp_dtype not in aliases:
            raise ValueError(f"meta.amp_dtype = {meta.get('amp_dtype')!r}: the autocast arithmetic is \"f16\" (fp16, the "
                             f"recipe's) or \"bf16\" (aliases: {sorted(aliases)})")
        self._inner().train_arithmetic = aliases[amp_dtype] if self.use_amp else "f32"
        # meta.amp_saves (optional): "16" (default under use_amp) keeps the gates the backward pass re-reads in the 16-bit type, as
        # the vendor LSTM does under autocast; "32" keeps them fp32 (fullsubnet_amd/train.py:train_arith_of)
        if "amp_saves" in meta:
            saves = str(meta["amp_saves"])
            if saves not in ("16", "32"):
                raise ValueError(f'meta.amp_saves = {saves!r}: "16" or "32"')
            self._inner().train_saves = saves

        ac = config["acoustics"]
        self.acoustic_config = ac
        self.n_fft, self.hop_length, self.win_length = ac["n_fft"], ac["hop_length"], ac["win_length"]
        # base_trainer.py:55-60; the phase (never read by this trainer) is not computed
        self.torch_stft = lambda y: stft(y, self.n_fft, self.hop_length, self.win_length, return_phase=False)
        self.torch_istft = lambda f, length=None, input_type="mag_phase": istft(
            f, self.n_fft, self.hop_length, self.win_length, length=length, input_type=input_type)

        tc = config.get("trainer", {})
        self.train_config = tc.get("train", {})
        self.epochs = self.train_config.get("epochs", 1)
        self.save_checkpoint_interval = self.train_config.get("save_checkpoint_interval", 1)
        self.clip_grad_norm_value = self.train_config.get("clip_grad_norm_value", 10.0)
        assert self.save_checkpoint_interval >= 1, \
            "Check the 'save_checkpoint_interval' parameter in the config. It should be large than one."
        self.validation_config = tc.get("validation", {})
        self.validation_interval = self.validation_config.get("validation_interval", 1)
        self.save_max_metric_score = self.validation_config.get("save_max_metric_score", True)
        assert self.validation_interval >= 1, \
            "Check the 'validation_interval' parameter in the config. It should be large than one."
        self.visualization_config = tc.get("visualization", {})

        self.start_epoch = 1
        self.best_score = -np.inf if self.save_max_metric_score else np.inf
        self.save_dir = None
        self.checkpoints_dir = None
        if meta.get("save_dir") is not None:
            self.save_dir = Path(meta["save_dir"]).expanduser().absolute() / meta.get("experiment_name", "experiment")
            self.checkpoints_dir = self.save_dir / "checkpoints"
        self.metrics = _default_metrics()
        self.validation_score_kind = None
        self.history = {"Loss/Train": {}, "Loss/Validation_Total": {}, "Loss/With_reverb": {}, "Loss/No_reverb": {},
                        "Score": {}}
        self.last_loss = None
        self.persistent_timeouts = 0  # training steps in which a persistent kernel ran out of time (update skipped)
        self.nonfinite_losses = 0     # training steps whose loss was not finite (update skipped on the device)

        if resume:
            self._resume_checkpoint()
        self.only_validation = only_validation
        if meta.get("preloaded_model_path"):
            self._preload_model(Path(meta["preloaded_model_path"]))
        if self.rank == 0 and self.checkpoints_dir is not None:
            self.checkpoints_dir.mkdir(parents=True, exist_ok=True)

    # ---- checkpoints: base_trainer.py:138-257 ------------------------------------------------
    def _inner(self):
        return self.model.module if isinstance(self.model, torch.nn.parallel.DistributedDataParallel) else self.model

    def _need_dir(self):
        if self.checkpoints_dir is None:
            raise RuntimeError("config['meta']['save_dir'] is not set: there is no checkpoint directory")
        return self.checkpoints_dir

    def _preload_model(self, model_path):
        model_path = Path(model_path).expanduser().absolute()
        assert model_path.exists(), f"The file {model_path.as_posix()} is not exist. please check path."
        model_checkpoint = torch.load(model_path.as_posix(), map_location="cpu", weights_only=False)
        self._inner().load_state_dict(model_checkpoint["model"], strict=False)
        self.model.to(self.device)
        if self.rank == 0:
            print(f"Model preloaded successfully from {model_path.as_posix()}.")

    def _resume_checkpoint(self):
        latest_model_path = self._need_dir().expanduser().absolute() / "latest_model.tar"
        assert latest_model_path.exists(), f"{latest_model_path} does not exist, can not load latest checkpoint."
        checkpoint = torch.load(latest_model_path.as_posix(), map_location="cpu", weights_only=False)
        if self._distributed:
            self.dist.barrier()  # nobody starts loading before the saving is finished
        self.start_epoch = checkpoint["epoch"] + 1
        self.best_score = checkpoint["best_score"]
        self.optimizer.load_state_dict(checkpoint["optimizer"])
        if checkpoint.get("scaler") and self.scaler.is_enabled():  # base_trainer.py:186
            self.scaler.load_state_dict(checkpoint["scaler"])
        state = {k.replace("module.", ""): v for k, v in checkpoint["model"].items()}
        self._inner().load_state_dict(state)
        if self.rank == 0:
            print(f"Model checkpoint is loaded. Training will begin at epoch {self.start_epoch}.")

    def _save_checkpoint(self, epoch, is_best_epoch=False):
        d = self._need_dir()
        d.mkdir(parents=True, exist_ok=True)
        bad = [k for k, v in self._inner().state_dict().items() if not bool(torch.isfinite(v).all())]
        if bad:  # never write a poisoned state over latest_model.tar
            raise RuntimeError(f"refusing to checkpoint: non-finite values in {bad[:4]}{' ...' if len(bad) > 4 else ''}")
        state_dict = {
            "epoch": epoch,
            "best_score": self.best_score,
            "optimizer": self.optimizer.state_dict(),
            "scaler": self.scaler.state_dict(),
            "model": {k: v.detach().cpu() for k, v in self._inner().state_dict().items()},
        }
        torch.save(state_dict, (d / "latest_model.tar").as_posix())
        torch.save(state_dict["model"], (d / f"model_{str(epoch).zfill(4)}.pth").as_posix())
        if is_best_epoch:
            torch.save(state_dict, (d / "best_model.tar").as_posix())

    def _is_best_epoch(self, score, save_max_metric_score=True):
        if save_max_metric_score and score >= self.best_score:
            self.best_score = score
            return True
        if not save_max_metric_score and score <= self.best_score:
            self.best_score = score
            return True
        return False

    def _set_models_to_train_mode(self):
        self.model.train()

    def _set_models_to_eval_mode(self):
        self.model.eval()

    # ---- epochs --------------------------------------------------------------------------------
    def _train_epoch(self, epoch):
        """fullsubnet/trainer.py:33-76."""
        loss_total, n = 0.0, 0
        # a persistent kernel that runs out of time in the MIDDLE of a step (its record is read at enqueue time by
        # every later persistent launch) must not raise out of forward / backward - under DistributedDataParallel the
        # peers would be left waiting in the gradient all-reduce.  The launches go ahead instead (NaN in, NaN out), the
        # fused optimizer skips the update on the device, and the record is read below, once per step
        # (every optimizer path of train_step skips the update on a non-finite gradient norm: ClipAdam and GradScaler on the
        # device, a stock optimizer after a host check of clip_grad_norm_'s result)
        _lib.stream_timeout_policy("defer", self.device)
        try:
            loss_total, n = self._train_batches(loss_total, n)
        finally:  # an exception in the loop must not leave the stream deferring: later inference would yield NaN silently
            _lib.stream_timeout_policy("refuse", self.device)
        self.last_loss = loss_total / max(n, 1)
        self.history["Loss/Train"][epoch] = self.last_loss
        return self.last_loss

    def _train_batches(self, loss_total, n):
        for noisy, clean in self.train_dataloader:
            loss = train_step(self.model, self.optimizer, noisy.to(self.device), clean.to(self.device), self.n_fft,
                              self.hop_length, self.win_length, self.clip_grad_norm_value, self.loss_function,
                              scaler=self.scaler)
            value = loss.item()  # host sync every step, like trainer.py:71
            # the step is complete on the device: did one of its persistent kernels run out of time (include/fsn_hip.h,
            # "residency contract")?  Its outputs were NaN then, the update was skipped on the device (non-finite
            # gradient norm), and the stream would refuse further persistent launches until the record is cleared
            status, events = _lib.stream_status(self.device, synchronize=False, raise_on_timeout=False)
            if status:
                self.persistent_timeouts += 1
                _lib.stream_status_clear(self.device)
                if self.rank == 0:
                    print(f"fullsubnet_amd.Trainer: a persistent kernel ran out of time in this step (status {status}, "
                          f"{events} outputs poisoned); the update was skipped")
            if not np.isfinite(value):
                self.nonfinite_losses += 1  # an overflow or a timeout: the update was skipped, keep it out of the mean
                continue
            loss_total += value
            n += 1
        return loss_total, n

    @torch.no_grad()
    def _validation_epoch(self, epoch):
        """fullsubnet/trainer.py:78-181: per utterance stft -> model -> loss against the cIRM target ->
        decompress -> complex mask -> istft, then the metric score of the "With_reverb" set.  The model runs
        without band dropping (validation batches are single utterances: model.py:114 only drops for B > 1)."""
        if self.valid_dataloader is None:
            raise RuntimeError("no validation dataloader")
        model = self._inner()
        loss_fn = self.loss_function or (lambda target, pred: mse_loss(pred, target))
        kinds = ("With_reverb", "No_reverb")
        loss_total = 0.0
        loss_list = {k: 0.0 for k in kinds}
        lists = {k: {"noisy": [], "clean": [], "enhanced": []} for k in kinds}
        n_items = 0
        for noisy, clean, name, speech_type in self.valid_dataloader:
            assert len(name) == 1, "The batch size for the validation stage must be one."
            speech_type = speech_type[0]
            noisy = noisy.to(self.device)
            clean = clean.to(self.device)
            noisy_mag, _, noisy_real, noisy_imag = self.torch_stft(noisy)
            _, _, clean_real, clean_imag = self.torch_stft(clean)
            cirm = build_complex_ideal_ratio_mask(noisy_real, noisy_imag, clean_real, clean_imag)  # [B, F, T, 2]
            try:
                crm = model(noisy_mag.unsqueeze(1)).permute(0, 2, 3, 1).contiguous()
                loss = float(loss_fn(cirm, crm))  # host sync: the utterance's launches are complete
                status, _ = _lib.stream_status(self.device, synchronize=False, raise_on_timeout=False)
            except _lib.FsnTimeout:
                status = 1
            if status:  # a persistent kernel ran out of time: NaN outputs, keep the utterance out of loss and score
                self.persistent_timeouts += 1
                _lib.stream_status_clear(self.device)
                continue
            m = decompress_cIRM(crm)
            enhanced_real = m[..., 0] * noisy_real - m[..., 1] * noisy_imag
            enhanced_imag = m[..., 1] * noisy_real + m[..., 0] * noisy_imag
            enhanced = self.torch_istft((enhanced_real, enhanced_imag), length=noisy.size(-1), input_type="real_imag")
            noisy_np = noisy.detach().squeeze(0).cpu().numpy()
            clean_np = clean.detach().squeeze(0).cpu().numpy()
            enhanced_np = enhanced.detach().squeeze(0).cpu().numpy()
            assert len(noisy_np) == len(clean_np) == len(enhanced_np)
            loss_total += loss
            loss_list[speech_type] += loss
            lists[speech_type]["noisy"].append(noisy_np)
            lists[speech_type]["clean"].append(clean_np)
            lists[speech_type]["enhanced"].append(enhanced_np)
            n_items += 1
        n_items = max(n_items, 1)
        self.history["Loss/Validation_Total"][epoch] = loss_total / n_items
        scores = {}
        for k in kinds:
            self.history[f"Loss/{k}"][epoch] = loss_list[k] / n_items  # the reference divides by the loader length
            scores[k] = self.metrics_score(lists[k]["noisy"], lists[k]["clean"], lists[k]["enhanced"])
        self.history["Score"][epoch] = scores
        self.last_validation = lists
        return float(scores["With_reverb"])

    def metrics_score(self, noisy_list, clean_list, enhanced_list):
        """base_trainer.py:319-369 without TensorBoard: (STOI + transform(WB_PESQ)) / 2 when both metrics are
        registered, the mean SI_SDR of the enhanced signals otherwise."""
        if not enhanced_list:
            return 0.0
        sr = self.acoustic_config.get("sr", 16000)
        if "STOI" in self.metrics and "WB_PESQ" in self.metrics:
            self.validation_score_kind = "(STOI + WB_PESQ) / 2"
            stoi = np.mean([self.metrics["STOI"](r, e, sr) for r, e in zip(clean_list, enhanced_list)])
            pesq = np.mean([self.metrics["WB_PESQ"](r, e, sr) for r, e in zip(clean_list, enhanced_list)])
            return float((stoi + transform_pesq_range(pesq)) / 2)
        self.validation_score_kind = "SI_SDR"
        return float(np.mean([self.metrics["SI_SDR"](r, e) for r, e in zip(clean_list, enhanced_list)]))

    def train(self):
        """base_trainer.py:372-420."""
        for epoch in range(self.start_epoch, self.epochs + 1):
            if self.only_validation and self.rank == 0:
                self._set_models_to_eval_mode()
                metric_score = self._validation_epoch(epoch)
                if self._is_best_epoch(metric_score, save_max_metric_score=self.save_max_metric_score):
                    self._save_checkpoint(epoch, is_best_epoch=True)
                continue
            self._set_models_to_train_mode()
            self._train_epoch(epoch)
            if (self.rank == 0 and self.checkpoints_dir is not None and self.save_checkpoint_interval != 0
                    and epoch % self.save_checkpoint_interval == 0):
                self._save_checkpoint(epoch)
            if self.rank == 0 and self.valid_dataloader is not None and epoch % self.validation_interval == 0:
                self._set_models_to_eval_mode()
                metric_score = self._validation_epoch(epoch)
                if self._is_best_epoch(metric_score, save_max_metric_score=self.save_max_metric_score):
                    if self.checkpoints_dir is not None:
                        self._save_checkpoint(epoch, is_best_epoch=True)
