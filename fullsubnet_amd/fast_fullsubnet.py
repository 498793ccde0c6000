"""Drop-in ``Model`` for recipes/dns_interspeech_2020/fast_fullsubnet/model.py:11-202 (BASELINE
config 4): mel filtering -> F_l2m encoder (2 LSTM blocks) -> sub-band bottleneck S on B * num_mels
rows at 1/shrink_size of the frame rate -> F_m2l decoder (2 LSTM blocks), every LSTM / Linear block
on the HIP kernels through ``SequenceModel``; mel matmul, unfold, time down/up-sampling and norms are
tensor-algebra glue (< 1 % of the FLOPs).

The reference takes the mel filterbank from torchaudio (``audio.transforms.MelScale(n_mels, 16000,
f_min=0, f_max=8000, n_stft)``, model.py:57-63), which is not part of the reference tree: ``MelScale``
below restates torchaudio's documented HTK / norm=None triangular filterbank (parity unpinned at
that boundary); the buffer keeps torchaudio's name ``mel_scale.fb`` so checkpoints load.
"""
import math

import torch
import torch.nn as nn

from .base_model import BaseModel, look_ahead_pad
from .sequence_model import SequenceModel


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """Triangular filters on the HTK mel scale, no area normalisation: [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)  # [n_freqs, n_mels + 2]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0)


class MelScale(nn.Module):
    def __init__(self, n_mels, sample_rate, f_min, f_max, n_stft):
        super().__init__()
        self.register_buffer("fb", melscale_fbanks(n_stft, f_min, f_max, n_mels, sample_rate))

    def forward(self, specgram):
        """[..., F, T] -> [..., n_mels, T]."""
        return torch.matmul(specgram.transpose(-1, -2), self.fb).transpose(-1, -2)


class Model(BaseModel):
    def __init__(self, look_ahead, shrink_size, sequence_model, num_mels, encoder_input_size,
                 bottleneck_hidden_size, bottleneck_num_layers, noisy_input_num_neighbors,
                 encoder_output_num_neighbors, norm_type="offline_laplace_norm", weight_init=False):
        super().__init__()
        assert sequence_model in ("GRU", "LSTM"), f"{self.__class__.__name__} only support GRU and LSTM."
        # F_l2m (model.py:35-54)
        self.encoder = nn.Sequential(
            SequenceModel(input_size=64, hidden_size=384, output_size=0, num_layers=1, bidirectional=False,
                          sequence_model=sequence_model, output_activate_function=None),
            SequenceModel(input_size=384, hidden_size=257, output_size=64, num_layers=1, bidirectional=False,
                          sequence_model=sequence_model, output_activate_function="ReLU"),
        )
        self.mel_scale = MelScale(n_mels=num_mels, sample_rate=16000, f_min=0, f_max=8000, n_stft=encoder_input_size)
        # S (model.py:66-74)
        self.bottleneck = SequenceModel(
            input_size=(noisy_input_num_neighbors * 2 + 1) + (encoder_output_num_neighbors * 2 + 1), output_size=1,
            hidden_size=bottleneck_hidden_size, num_layers=bottleneck_num_layers, bidirectional=False,
            sequence_model=sequence_model, output_activate_function="ReLU")
        # F_m2l (model.py:77-96)
        self.decoder_lstm = nn.Sequential(
            SequenceModel(input_size=64 + 64, hidden_size=512, output_size=0, num_layers=1, bidirectional=False,
                          sequence_model=sequence_model, output_activate_function=None),
            SequenceModel(input_size=512, hidden_size=512, output_size=257 * 2, num_layers=1, bidirectional=False,
                          sequence_model=sequence_model, output_activate_function=None),
        )
        self.shrink_size = shrink_size
        self.look_ahead = look_ahead
        self.num_mels = num_mels
        self.noisy_input_num_neighbors = noisy_input_num_neighbors
        self.enc_output_num_neighbors = encoder_output_num_neighbors
        self.norm = self.norm_wrapper(norm_type)
        if weight_init:
            self.apply(self.weight_init)

    def real_time_downsampling(self, input):
        """model.py:108-129: frame 0 kept, then means over consecutive blocks of shrink_size frames
        (the last block may be shorter): [B, C, F, T] -> [B, C, F, 1 + ceil((T - 1) / shrink_size)]."""
        s = self.shrink_size
        rest = input[..., 1:]
        n_full = rest.shape[-1] // s
        parts = [input[..., 0:1]]
        if rest.shape[-1] % s == 0:
            # the reference's "last block" is then a full block: blocks 0 .. n_full-2 stacked, last apart
            n_full -= 1
        if n_full > 0:
            parts.append(rest[..., :n_full * s].reshape(*rest.shape[:-1], n_full, s).mean(dim=-1))
        parts.append(rest[..., n_full * s:].mean(dim=-1, keepdim=True))
        return torch.cat(parts, dim=-1)

    def real_time_upsampling(self, input, target_len=False):
        """model.py:131-140: repeat every frame shrink_size times, trim to target_len."""
        out = input.repeat_interleave(self.shrink_size, dim=-1)
        return out[..., :target_len] if target_len else out

    def forward(self, mix_mag):
        """mix_mag [B, 1, F, T] -> [B, 2, F, T] (model.py:143-202)."""
        assert mix_mag.dim() == 4
        mix_mag = look_ahead_pad(mix_mag, self.look_ahead)
        batch_size, num_channels, num_freqs, num_frames = mix_mag.size()
        assert num_channels == 1, f"{self.__class__.__name__} takes a magnitude feature as the input."
        mix_mel_mag = self.mel_scale(mix_mag)  # [B, 1, F_mel, T]
        enc_input = self.norm(mix_mel_mag).reshape(batch_size, -1, num_frames)
        enc_output = self.encoder(enc_input).reshape(batch_size, num_channels, -1, num_frames)
        noisy_unfold = self.freq_unfold(mix_mel_mag, num_neighbors=self.noisy_input_num_neighbors)
        noisy_unfold = noisy_unfold.reshape(batch_size, self.num_mels, self.noisy_input_num_neighbors * 2 + 1, num_frames)
        enc_unfold = self.freq_unfold(enc_output, num_neighbors=self.enc_output_num_neighbors)
        enc_unfold = enc_unfold.reshape(batch_size, self.num_mels, self.enc_output_num_neighbors * 2 + 1, num_frames)
        bn_input = torch.cat([noisy_unfold, enc_unfold], dim=2)
        num_sb_unit_freqs = bn_input.shape[2]
        bn_input_shrink = self.norm(self.real_time_downsampling(bn_input))
        bn_input_shrink = bn_input_shrink.reshape(batch_size * self.num_mels, num_sb_unit_freqs, -1)
        bn_output_shrink = self.bottleneck(bn_input_shrink)  # [B * F_mel, 1, T // shrink]
        bn_output_shrink = bn_output_shrink.reshape(batch_size, self.num_mels, 1, -1).permute(0, 2, 1, 3)
        bn_output = self.real_time_upsampling(bn_output_shrink, target_len=num_frames)  # [B, 1, F_mel, T]
        dec_input = torch.cat([enc_output, bn_output], dim=2).reshape(batch_size, -1, num_frames)
        dec_output = self.decoder_lstm(dec_input).reshape(batch_size, 2, num_freqs, num_frames)
        return dec_output[:, :, :, self.look_ahead:]
