"""Drop-in ``Trainer`` for recipes/dns_interspeech_2020/fullsubnet/trainer.py: same constructor
signature, same ``_train_epoch(epoch)`` (fullsubnet/trainer.py:33-76) with the data-parallel strategy
of audio_zen/trainer/base_trainer.py:32 (DistributedDataParallel over the "nccl" = RCCL backend; the
custom autograd function of fullsubnet_amd/train.py is an ordinary graph node, so DDP's bucketed
gradient all-reduce works unchanged).  fp32 compute (``use_amp = true`` is accepted and runs in fp32,
a precision superset of the reference's fp16 autocast); TensorBoard / PESQ / STOI
validation of the reference is host-side tooling and not part of this path."""
import torch

from .acoustics.feature import istft, stft  # noqa: F401
from .train import train_step


class Trainer:
    def __init__(self, dist, rank, config, resume, only_validation, model, loss_function, optimizer,
                 train_dataloader, validation_dataloader=None):
        self.dist = dist
        self.rank = rank
        self.config = config
        self.device = torch.device("cuda", rank)
        model = model.to(self.device)
        if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[rank])  # base_trainer.py:32
        self.model = model
        self.optimizer = optimizer
        self.loss_function = loss_function
        self.train_dataloader = train_dataloader
        self.valid_dataloader = validation_dataloader
        ac = config["acoustics"]
        self.n_fft, self.hop_length, self.win_length = ac["n_fft"], ac["hop_length"], ac["win_length"]
        tc = config.get("trainer", {}).get("train", {})
        self.epochs = tc.get("epochs", 1)
        self.clip_grad_norm_value = tc.get("clip_grad_norm_value", 10.0)
        # meta.use_amp = true (every shipped train TOML) asks for fp16 autocast + GradScaler
        # (fullsubnet/trainer.py:56,63-69).  The HIP training kernels compute in fp32 - at least the
        # precision the flag asks for - so the flag is accepted and loss scaling becomes the identity.
        self.use_amp = bool(config.get("meta", {}).get("use_amp", False))
        if self.use_amp and rank == 0:
            print("fullsubnet_amd.Trainer: meta.use_amp = true -> computing in fp32 (no 16-bit kernels yet); "
                  "GradScaler is not needed")
        self.last_loss = None

    def _train_epoch(self, epoch):
        """fullsubnet/trainer.py:33-76."""
        self.model.train()
        loss_total, n = 0.0, 0
        for noisy, clean in self.train_dataloader:
            loss = train_step(self.model, self.optimizer, noisy.to(self.device), clean.to(self.device), self.n_fft,
                              self.hop_length, self.win_length, self.clip_grad_norm_value, self.loss_function)
            loss_total += loss.item()  # host sync every step, like trainer.py:71
            n += 1
        self.last_loss = loss_total / max(n, 1)
        return self.last_loss

    def train(self):
        """base_trainer.py:372-420 without checkpoint / validation tooling."""
        for epoch in range(1, self.epochs + 1):
            self._train_epoch(epoch)
