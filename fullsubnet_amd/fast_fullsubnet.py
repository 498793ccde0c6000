"""Drop-in ``Model`` for recipes/dns_interspeech_2020/fast_fullsubnet/model.py:11-202 (BASELINE
config 4): mel filtering -> F_l2m encoder (2 LSTM blocks) -> sub-band bottleneck S on B * num_mels
rows at 1/shrink_size of the frame rate -> F_m2l decoder (2 LSTM blocks), every LSTM / Linear block
on the HIP kernels through ``SequenceModel``.  In inference the tensors between the blocks stay time-major and the glue
(look-ahead pad, norms, unit windows, time down/up-sampling, concatenations, the final reshape) is HIP too
(``_forward_rows``, csrc/fast_glue_kernels.hip); the tensor-algebra ``forward`` below it is what autograd records in
training and what host-side unit tests of the glue run.

The reference takes the mel filterbank from torchaudio (``audio.transforms.MelScale(n_mels, 16000,
f_min=0, f_max=8000, n_stft)``, model.py:57-63), which is not part of the reference tree: ``MelScale``
below restates torchaudio's documented HTK / norm=None triangular filterbank (parity unpinned at
that boundary); the buffer keeps torchaudio's name ``mel_scale.fb`` so checkpoints load.
"""
import math

import torch
import torch.nn as nn

from .base_model import BaseModel, look_ahead_pad
from . import _lib
from .sequence_model import SequenceModel, linear_infer, pair_forward, pair_forward_rows, pair_fusable


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

    def linear_weights(self):
        """The filterbank as an nn.Linear weight [n_mels, F] and a zero bias (made once per buffer version)."""
        key = (self.fb.data_ptr(), self.fb._version, str(self.fb.device))
        if getattr(self, "_w_key", None) != key:
            self._w = self.fb.detach().t().contiguous()  # [n_mels, F]
            self._b = torch.zeros(self._w.shape[0], dtype=torch.float32, device=self._w.device)
            self._w_key = key
        return self._w, self._b

    def forward(self, specgram):
        """[..., F, T] -> [..., n_mels, T]: specgram^T fb, through the in-tree MFMA GEMM (fsn_linear_forward with the
        filterbank as an nn.Linear weight [n_mels, F] and a zero bias) - no vendor BLAS on the path."""
        if not specgram.is_cuda or (torch.is_grad_enabled() and specgram.requires_grad):
            # host-side callers (CPU unit tests of the glue) and inputs that want a gradient (a learned front end): the
            # same product as tensor algebra, autograd-tracked - the pattern of base_model._hip_norm
            return torch.matmul(specgram.transpose(-1, -2), self.fb).transpose(-1, -2)
        F, T = specgram.shape[-2], specgram.shape[-1]
        lead = specgram.shape[:-2]
        self.linear_weights()
        Fp = (F + 15) // 16 * 16
        x2 = torch.zeros((specgram.numel() // (F * T), T, Fp), dtype=torch.float32, device=specgram.device)
        x2[..., :F] = specgram.detach().reshape(-1, F, T).transpose(1, 2)
        mel = linear_infer(x2.reshape(-1, Fp), self._w, self._b, False)  # [rows, n_mels]
        return mel.reshape(*lead, T, -1).transpose(-1, -2)


def _lstm_block(kind, n_in, n_hidden, n_out, layers=1, act=None):
    return SequenceModel(n_in, n_out, n_hidden, layers, False, kind, act)


class Model(BaseModel):
    """Constructor keywords and parameter names (``encoder.{0,1}``, ``mel_scale.fb``, ``bottleneck``,
    ``decoder_lstm.{0,1}``) are the reference's (fast_fullsubnet/model.py:12-106)."""

    def __init__(self, look_ahead, shrink_size, sequence_model, num_mels, encoder_input_size,
                 bottleneck_hidden_size, bottleneck_num_layers, noisy_input_num_neighbors,
                 encoder_output_num_neighbors, norm_type="offline_laplace_norm", weight_init=False):
        super().__init__()
        if sequence_model not in ("GRU", "LSTM"):
            raise AssertionError(f"{self.__class__.__name__} only support GRU and LSTM.")
        self.look_ahead, self.shrink_size, self.num_mels = look_ahead, shrink_size, num_mels
        self.noisy_input_num_neighbors = noisy_input_num_neighbors
        self.enc_output_num_neighbors = encoder_output_num_neighbors
        self.norm = self.norm_wrapper(norm_type)
        cell = sequence_model
        # the widths below are literals in the reference too (model.py:36-96): 64 mel bands, 257 bins
        self.encoder = nn.Sequential(_lstm_block(cell, 64, 384, 0),                    # F_l2m, part 1
                                     _lstm_block(cell, 384, 257, 64, act="ReLU"))      # F_l2m, part 2
        self.mel_scale = MelScale(n_mels=num_mels, sample_rate=16000, f_min=0, f_max=8000, n_stft=encoder_input_size)
        unit_width = (2 * noisy_input_num_neighbors + 1) + (2 * encoder_output_num_neighbors + 1)
        self.bottleneck = _lstm_block(cell, unit_width, bottleneck_hidden_size, 1, bottleneck_num_layers, "ReLU")  # S
        self.decoder_lstm = nn.Sequential(_lstm_block(cell, 64 + 64, 512, 0),         # F_m2l, part 1
                                          _lstm_block(cell, 512, 512, 257 * 2))       # F_m2l, part 2
        if weight_init:
            self.apply(self.weight_init)

    def real_time_downsampling(self, input):
        """model.py:108-129: frame 0 is kept, the remaining frames are averaged in consecutive blocks of
        ``shrink_size`` (a shorter last block is averaged over what it has):
        [B, C, F, T] -> [B, C, F, 1 + ceil((T - 1) / shrink_size)]."""
        s = self.shrink_size
        head, rest = input[..., :1], input[..., 1:]
        n_rest = rest.shape[-1]
        n_whole = n_rest // s
        if n_rest % s == 0:
            n_whole -= 1  # the reference averages its last block separately even when it is a whole one
        pieces = [head]
        if n_whole > 0:
            pieces.append(rest[..., :n_whole * s].unflatten(-1, (n_whole, s)).mean(dim=-1))
        pieces.append(rest[..., n_whole * s:].mean(dim=-1, keepdim=True))
        return torch.cat(pieces, dim=-1)

    def real_time_upsampling(self, input, target_len=False):
        """model.py:131-140: hold every low-rate frame for ``shrink_size`` frames, cut to ``target_len``."""
        held = torch.repeat_interleave(input, self.shrink_size, dim=-1)
        return held[..., :target_len] if target_len else held

    def _unit_windows(self, x, neighbors):
        """[B, 1, M, T] -> [B, M, 2 n + 1, T]: every mel band with its ``neighbors`` on each side."""
        b, _, m, t = x.shape
        return self.freq_unfold(x, num_neighbors=neighbors).reshape(b, m, 2 * neighbors + 1, t)

    def _rows_path_ok(self, mix_mag):
        enc0, enc1 = self.encoder
        dec0, dec1 = self.decoder_lstm
        B = mix_mag.shape[0]
        return (getattr(self, "rows_path", True) and self.norm == self.offline_laplace_norm and self.num_mels % 16 == 0
                and self.num_mels == enc0.input_size == enc1.output_size and dec0.input_size == 2 * self.num_mels
                and dec1.output_size == 2 * mix_mag.shape[2] and self.bottleneck.cell == "LSTM"
                and self.bottleneck.output_size == 1 and pair_fusable(enc0, enc1, B) and pair_fusable(dec0, dec1, B)
                and mix_mag.shape[3] + self.look_ahead >= 2 and mix_mag.dtype == torch.float32)

    def _bottleneck_rows(self, units, N):
        """The bottleneck block on time-major rows [Ts, Np, Wp] -> ([Ts, Np(, 1)] output, 1 if it is still the output
        layer's PRE-activation).  A two-layer LSTM stack whose last layer the persistent kernel takes with its fused output
        layer (fsn_lstm_layer_forward_fc: batch 128+ at 64 bands) never writes that layer's [Ts, Np, 384] hidden sequence;
        the ReLU then rides on the decoder-input kernel."""
        from .sequence_model import lstm_layer_infer
        L = _lib.lib()
        bn = self.bottleneck
        Ts, Np, _ = units.shape
        H = bn.hidden_size
        if (bn.cell == "LSTM" and bn.num_layers == 2 and bn.output_size == 1 and bn.output_activate_function == "ReLU"
                and getattr(self, "fused_output_layer", True) and L.fsn_lstm_layer_fc_supported(Ts, Np, H, H, H, 1)):
            layers, fc = bn._inference_weights()
            h0 = lstm_layer_infer(units, *layers[0])                                      # [Ts, Np, H]
            out = torch.empty((Ts, Np), dtype=torch.float32, device=units.device)
            ws = _lib.workspace(L.fsn_lstm_layer_fc_workspace_bytes(Ts, Np, H, H), units.device)
            w_ih, w_hh, b_ih, b_hh = layers[1]
            _lib.check(L.fsn_lstm_layer_forward_fc(
                _lib.dev_ptr(h0), H, _lib.dev_ptr(w_ih), _lib.dev_ptr(w_hh), _lib.dev_ptr(b_ih), _lib.dev_ptr(b_hh), Ts, Np, H, H,
                _lib.dev_ptr(fc[0]), _lib.dev_ptr(fc[1]), 1, _lib.dev_ptr(out), None, Np, ws.data_ptr(), ws.numel(),
                _lib.stream_ptr(units.device)))
            return out, 1
        return bn.forward_time_major(units, N, rows_out=True), 0

    def _forward_rows(self, mix_mag):
        """The inference forward with every tensor between the blocks time-major and the glue on fast_glue_kernels.hip
        (model.py:143-202 line by line in the comments)."""
        L = _lib.lib()
        dev = mix_mag.device
        st = _lib.stream_ptr(dev)
        mag = mix_mag.detach().contiguous()
        B, _, F, T0 = mag.shape
        M, s, la = self.num_mels, self.shrink_size, self.look_ahead
        T = T0 + la
        Bp, Fp = (B + 15) // 16 * 16, (F + 15) // 16 * 16
        f32 = dict(dtype=torch.float32, device=dev)
        ws = _lib.workspace(L.fsn_fast_glue_workspace_bytes(T, B, M, s), dev)
        # :151-157 pad + mel_scale
        rows = torch.empty((T, Bp, Fp), **f32)
        _lib.check(L.fsn_fast_spec_rows(_lib.dev_ptr(mag, "mix_mag"), B, F, T0, la, _lib.dev_ptr(rows), Bp, Fp, st))
        w, b = self.mel_scale.linear_weights()
        mel = linear_infer(rows.reshape(T * Bp, Fp), w, b, False)                       # [T Bp, M]
        # :160 norm -> encoder
        enc_in = torch.empty((T, Bp, M), **f32)
        _lib.check(L.fsn_fast_norm_rows(_lib.dev_ptr(mel), T, B, Bp, M, _lib.dev_ptr(enc_in), ws.data_ptr(), ws.numel(), st))
        enc = pair_forward_rows(*self.encoder, enc_in)                                   # [T, Bp, M]
        # :163-178 unit windows, down-sampling, norm -> bottleneck
        n_mel, n_enc = self.noisy_input_num_neighbors, self.enc_output_num_neighbors
        W = (2 * n_mel + 1) + (2 * n_enc + 1)
        Wp, N = (W + 15) // 16 * 16, B * M
        # the bottleneck's rows: padded to a count the persistent kernels take whole where that is the faster plan (e.g. 112
        # utterances x 64 bands = 448 row tiles: 224 workgroups x 2 tiles instead of 256 x 1 + 192 tiles step by step)
        Np, Ts = L.fsn_lstm_layer_plan_rows((N + 15) // 16 * 16, self.bottleneck.hidden_size), L.fsn_fast_low_rate_frames(T, s)
        units = torch.empty((Ts, Np, Wp), **f32)
        _lib.check(L.fsn_fast_bottleneck_input(_lib.dev_ptr(mel), _lib.dev_ptr(enc), enc.stride(1), T, B, Bp, M, n_mel, n_enc, s,
                                               _lib.dev_ptr(units), Np, Wp, ws.data_ptr(), ws.numel(), st))
        slow, relu = self._bottleneck_rows(units, N)                                     # [Ts, Np(, 1)]
        # :180-190 up-sampling, cat -> decoder
        dec_in = torch.empty((T, Bp, 2 * M), **f32)
        _lib.check(L.fsn_fast_decoder_input(_lib.dev_ptr(enc), enc.stride(1), _lib.dev_ptr(slow), slow.stride(0), slow.stride(1),
                                            relu, T, B, Bp, M, s, _lib.dev_ptr(dec_in), st))
        out = pair_forward_rows(*self.decoder_lstm, dec_in)                              # [T, Bp, 2 F]
        # :200-202 reshape + look-ahead slice
        mask = torch.empty((B, 2, F, T0), **f32)
        _lib.check(L.fsn_fast_mask_out(_lib.dev_ptr(out), out.stride(1), T, B, Bp, F, la, _lib.dev_ptr(mask), st))
        return mask

    def forward(self, mix_mag):
        """mix_mag [B, 1, F, T] -> [B, 2, F, T] (model.py:143-202)."""
        if mix_mag.dim() != 4 or mix_mag.shape[1] != 1:
            raise AssertionError(f"{self.__class__.__name__} takes a magnitude feature as the input ([B, 1, F, T]).")
        if mix_mag.is_cuda and not torch.is_grad_enabled():
            # the model has no cross-utterance term: a batch well beyond ONE round of the bottleneck's persistent kernels
            # (4 row tiles per workgroup: 256 utterances x 64 bands on 256 CUs) runs as whole rounds plus a remainder -
            # 512 utterances 124 -> 2 x 51 ms (the block pairs' per-step launches and the multi-round launch cost more)
            chunk = torch.cuda.get_device_properties(mix_mag.device).multi_processor_count * 4 * 16 // self.num_mels
            if chunk >= 16 and mix_mag.shape[0] >= chunk + chunk // 2:
                return torch.cat([self.forward(mix_mag[i:i + chunk]) for i in range(0, mix_mag.shape[0], chunk)], dim=0)
            if self._rows_path_ok(mix_mag):
                return self._forward_rows(mix_mag)
        if torch.is_grad_enabled():
            # the trainer's arithmetic (Model.train_arithmetic: "f16" under use_amp = true, train_shrinkSize2.toml:5) reaches
            # the bottleneck, 90 % of the step's products; the encoder / decoder blocks (72 rows) stay fp32
            from .train import train_arith_of
            self.bottleneck.train_arithmetic = train_arith_of(self)
        mag = look_ahead_pad(mix_mag, self.look_ahead)
        n_batch, _, n_bins, n_frames = mag.shape
        mel = self.mel_scale(mag)                                                         # [B, 1, M, T]
        enc = pair_forward(*self.encoder, self.norm(mel).reshape(n_batch, -1, n_frames))  # [B, M, T]
        enc = enc.reshape(n_batch, 1, -1, n_frames)
        units = torch.cat([self._unit_windows(mel, self.noisy_input_num_neighbors),
                           self._unit_windows(enc, self.enc_output_num_neighbors)], dim=2)  # [B, M, W, T]
        width = units.shape[2]
        slow = self.norm(self.real_time_downsampling(units))                              # [B, M, W, T / s]
        slow_out = self.bottleneck(slow.reshape(n_batch * self.num_mels, width, -1))      # [B M, 1, T / s]
        slow_out = slow_out.reshape(n_batch, self.num_mels, 1, -1).permute(0, 2, 1, 3)
        band_gain = self.real_time_upsampling(slow_out, target_len=n_frames)              # [B, 1, M, T]
        dec_in = torch.cat([enc, band_gain], dim=2).reshape(n_batch, -1, n_frames)
        mask = pair_forward(*self.decoder_lstm, dec_in).reshape(n_batch, 2, n_bins, n_frames)
        return mask[..., self.look_ahead:]
