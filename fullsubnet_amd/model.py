"""Drop-in ``Model`` for ``recipes/dns_interspeech_2020/fullsubnet/model.py`` backed by libfsn_hip.so.

Same constructor keywords, same ``state_dict()`` keys / shapes (released checkpoints load with
``strict=True``, base_inferencer.py:158), same ``forward(noisy_mag [B,1,F,T]) -> [B,2,F,T]``.
The PyTorch modules below only *hold* the parameters; no ATen LSTM / Linear / unfold kernel runs.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from .acoustics.feature import drop_band
from .sequence_model import SequenceModel


class Model(nn.Module):
    def __init__(self, num_freqs, look_ahead, sequence_model, fb_num_neighbors, sb_num_neighbors,
                 fb_output_activate_function, sb_output_activate_function, fb_model_hidden_size,
                 sb_model_hidden_size, norm_type="offline_laplace_norm", num_groups_in_drop_band=2,
                 weight_init=True):
        """fullsubnet/model.py:10-70 (same arguments)."""
        super().__init__()
        assert sequence_model in ("GRU", "LSTM"), f"{self.__class__.__name__} only support GRU and LSTM."
        # configurations of the shipped TOMLs run fused in libfsn_hip (fsn_fullsubnet_forward / fsn_enhance);
        # every other combination the reference constructor accepts (GRU, fb_num_neighbors > 0, other output
        # activations, the three extra norms) runs composed from SequenceModel blocks (_forward_composed)
        from .base_model import BaseModel
        BaseModel().norm_wrapper(norm_type)  # unknown norm: NotImplementedError here, like model.py:62
        self._fused = (sequence_model == "LSTM" and fb_num_neighbors == 0 and fb_output_activate_function == "ReLU"
                       and not sb_output_activate_function and norm_type in _lib.NORM_TYPES
                       and sb_model_hidden_size == 384 and fb_model_hidden_size % 64 == 0)
        self.fb_model = SequenceModel(num_freqs, num_freqs, fb_model_hidden_size, 2, False, sequence_model,
                                      fb_output_activate_function)
        self.sb_model = SequenceModel((sb_num_neighbors * 2 + 1) + (fb_num_neighbors * 2 + 1), 2,
                                      sb_model_hidden_size, 2, False, sequence_model, sb_output_activate_function)
        self.num_freqs = num_freqs
        self.sb_num_neighbors = sb_num_neighbors
        self.fb_num_neighbors = fb_num_neighbors
        self.look_ahead = look_ahead
        self.norm_type = norm_type
        self.num_groups_in_drop_band = num_groups_in_drop_band
        self._cfg = _lib.Cfg(num_freqs, look_ahead, sb_num_neighbors, fb_model_hidden_size, sb_model_hidden_size,
                             _lib.NORM_TYPES.get(norm_type, 0), _lib.ARITH[_lib.default_arith()])
        self._packed = None
        self._packed_key = None
        if weight_init:
            self.apply(self.weight_init)

    @property
    def arithmetic(self):
        """"f32" (default: fp32 MFMA, what every parity claim refers to) or "f16x3" (opt-in experiment)."""
        return {v: k for k, v in _lib.ARITH.items()}[self._cfg.arith]

    @arithmetic.setter
    def arithmetic(self, name):
        if name not in _lib.ARITH:
            raise _lib.FsnError(f"unknown arithmetic {name!r} (choose from {sorted(_lib.ARITH)})")
        self._cfg.arith = _lib.ARITH[name]

    # base_model.py:374-439 (pure initialisation, no kernels involved)
    @staticmethod
    def weight_init(m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_normal_(m.weight.data)
            nn.init.normal_(m.bias.data)
        elif isinstance(m, (nn.LSTM, nn.GRU)):  # base_model.py:416-421: both recurrent cell types
            for param in m.parameters():
                if len(param.shape) >= 2:
                    nn.init.orthogonal_(param.data)
                else:
                    nn.init.normal_(param.data)

    # ------------------------------------------------------------------------------------------
    def _params_in_abi_order(self):
        sd = dict(self.named_parameters())
        return [sd[k] for k in _lib.STATE_KEYS]

    def packed_weights(self):
        """Weights re-tiled into MFMA fragment order; rebuilt whenever a parameter changed."""
        ps = self._params_in_abi_order()
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in ps)
        if self._packed is None or key != self._packed_key:
            L = _lib.lib()
            dev = ps[0].device
            tensors = [p.detach().contiguous() for p in ps]
            prm = _lib.Params(*[_lib.dev_ptr(t, k) for t, k in zip(tensors, _lib.STATE_KEYS)])
            nbytes = L.fsn_fullsubnet_packed_bytes(ctypes.byref(self._cfg))
            packed = _lib.workspace(nbytes, dev)
            _lib.check(L.fsn_fullsubnet_pack(ctypes.byref(self._cfg), ctypes.byref(prm), packed.data_ptr(),
                                             packed.numel(), _lib.stream_ptr(dev)))
            self._packed, self._packed_key = packed, key
        return self._packed

    def forward(self, noisy_mag):
        """fullsubnet/model.py:72-136.  noisy_mag [B, 1, F, T] -> compressed cIRM [B, 2, F, T]
        ([B, 2, F // g, T] when B > 1 and num_groups_in_drop_band = g > 1, quirk Q1)."""
        assert noisy_mag.dim() == 4
        batch_size, num_channels, num_freqs, num_frames = noisy_mag.size()
        assert num_channels == 1, f"{self.__class__.__name__} takes the mag feature as inputs."
        assert num_freqs == self.num_freqs
        if not self._fused:
            if self._composed_rows_ok(noisy_mag):
                return self._forward_composed_rows(noisy_mag)
            return self._forward_composed(noisy_mag)
        if torch.is_grad_enabled() and (self.training or noisy_mag.requires_grad):
            # training step (fullsubnet/trainer.py:56-63): autograd graph with the LSTM layers
            # (forward + BPTT) on the HIP kernels, see fullsubnet_amd/train.py
            from .train import forward_train
            if not noisy_mag.is_cuda:
                raise _lib.FsnError("noisy_mag must live on a ROCm device; this path has no CPU implementation")
            return forward_train(self, noisy_mag)
        x = noisy_mag.contiguous()
        L = _lib.lib()
        out = torch.empty((batch_size, 2, num_freqs, num_frames), dtype=torch.float32, device=x.device)
        ws = _lib.workspace(L.fsn_fullsubnet_workspace_bytes(ctypes.byref(self._cfg), batch_size, num_frames),
                            x.device)
        _lib.check(L.fsn_fullsubnet_forward(ctypes.byref(self._cfg), self.packed_weights().data_ptr(),
                                            _lib.dev_ptr(x, "noisy_mag"), batch_size, num_frames,
                                            _lib.dev_ptr(out), ws.data_ptr(), ws.numel(), _lib.stream_ptr(x.device)))
        if batch_size > 1 and self.num_groups_in_drop_band > 1:
            # model.py:114-119 drops bands for any B > 1; rows of the sub-band model are independent
            # and the norm statistics are taken before the drop, so selecting afterwards is identical.
            out = drop_band(out, num_groups=self.num_groups_in_drop_band)
        return out

    @torch.no_grad()
    def fullband_output(self, noisy_mag):
        """The intermediate ``fb_output`` of fullsubnet/model.py:95 (look-ahead pad -> norm -> fb_model):
        noisy_mag [B, 1, F, T] -> [B, F, T + look_ahead].  Stage-level parity checks; fused configurations only."""
        assert noisy_mag.dim() == 4
        B, C, F, T = noisy_mag.size()
        assert C == 1 and F == self.num_freqs
        if not self._fused:
            raise _lib.FsnError("fullband_output is built for the fused configurations (shipped TOMLs)")
        x = noisy_mag.contiguous()
        L = _lib.lib()
        out = torch.empty((B, F, T + self.look_ahead), dtype=torch.float32, device=x.device)
        ws = _lib.workspace(L.fsn_fullsubnet_workspace_bytes(ctypes.byref(self._cfg), B, T), x.device)
        _lib.check(L.fsn_fullsubnet_fullband(ctypes.byref(self._cfg), self.packed_weights().data_ptr(),
                                             _lib.dev_ptr(x, "noisy_mag"), B, T, _lib.dev_ptr(out), ws.data_ptr(),
                                             ws.numel(), _lib.stream_ptr(x.device)))
        return out

    @torch.no_grad()
    def forward_rows(self, noisy_mag, row_begin, row_end):
        """The sub-band model on the rows [row_begin, row_end) of the B*F flattened (b, f) sequences that
        fullsubnet/model.py:121-128 feeds the sub-band LSTM - the multi-GPU partition of the path.
        noisy_mag [B, 1, F, T] is the whole batch (only the utterances the slice touches are read; their full-band
        model and norm statistics are computed whole).  Returns [row_end - row_begin, 2, T]: row n = b F + f of the
        compressed cIRM, i.e. ``forward(noisy_mag)[b, :, f, :]`` without the band dropping of quirk Q1.  Inference,
        fused configurations only."""
        assert noisy_mag.dim() == 4
        batch_size, num_channels, num_freqs, num_frames = noisy_mag.size()
        assert num_channels == 1 and num_freqs == self.num_freqs
        if not self._fused:
            raise _lib.FsnError("forward_rows is built for the fused configurations (shipped TOMLs); shard composed "
                                "configurations by utterance (parallel.enhance_sharded)")
        if not 0 <= row_begin < row_end <= batch_size * num_freqs:
            raise _lib.FsnError(f"row range [{row_begin}, {row_end}) is not inside the {batch_size * num_freqs} "
                                f"sub-band rows of the batch")
        x = noisy_mag.contiguous()
        L = _lib.lib()
        out = torch.empty((row_end - row_begin, 2, num_frames), dtype=torch.float32, device=x.device)
        ws = _lib.workspace(L.fsn_fullsubnet_rows_workspace_bytes(ctypes.byref(self._cfg), batch_size, num_frames,
                                                                  row_begin, row_end), x.device)
        _lib.check(L.fsn_fullsubnet_forward_rows(ctypes.byref(self._cfg), self.packed_weights().data_ptr(),
                                                 _lib.dev_ptr(x, "noisy_mag"), batch_size, num_frames, row_begin,
                                                 row_end, _lib.dev_ptr(out), ws.data_ptr(), ws.numel(),
                                                 _lib.stream_ptr(x.device)))
        return out

    @torch.no_grad()
    def forward_row_sharded(self, noisy_mag, group=None):
        """``forward`` with the batch x frequency rows sharded over a torch.distributed group: every rank holds the
        whole ``noisy_mag``, runs ``forward_rows`` on its contiguous share of the B*F rows and ONE all-gather
        (RCCL over xGMI with backend "nccl") re-assembles the full-band mask [B, 2, F, T] on every rank.  Full masks
        for every sample (no band dropping)."""
        import torch.distributed as dist
        from .parallel import gather_shards, shard_bounds
        B, _, F, T = noisy_mag.shape
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if world == 1:
            rows = self.forward_rows(noisy_mag, 0, B * F)
        else:
            lo, hi = shard_bounds(B * F, dist.get_rank(group), world)
            local = (self.forward_rows(noisy_mag, lo, hi) if hi > lo
                     else noisy_mag.new_empty((0, 2, T)))
            rows = gather_shards(local, B * F, group=group)
        return rows.view(B, F, 2, T).permute(0, 2, 1, 3).contiguous()

    def _composed_rows_ok(self, noisy_mag):
        """Whether the composed configuration's INFERENCE forward runs with its glue on the library as well
        (``_forward_composed_rows``): no full-band neighbours, one of the two shipped norms, a sub-band window of up to 32
        columns, and a batch drop_band accepts (model.py:114-119; anything else takes the tensor algebra and fails there
        exactly like the reference)."""
        B = noisy_mag.shape[0]
        g = self.num_groups_in_drop_band
        return (getattr(self, "composed_rows", True) and noisy_mag.is_cuda and noisy_mag.dtype == torch.float32
                and not (torch.is_grad_enabled() and (noisy_mag.requires_grad or any(p.requires_grad for p in self.parameters())))
                and self.fb_num_neighbors == 0 and self.norm_type in _lib.NORM_TYPES and 2 * self.sb_num_neighbors + 2 <= 32
                and (B == 1 or g <= 1 or B > g) and noisy_mag.shape[3] + self.look_ahead >= 1)

    def _forward_composed_rows(self, noisy_mag):
        """``_forward_composed`` in inference with every tensor between the two SequenceModel blocks in the time-major,
        zero-padded layout the LSTM / GRU / Linear entries take, written by the library's own glue kernels (the ones of the
        fused training graph, train_glue_kernels.hip: look-ahead pad + norm -> full-band block -> unfold ++ full-band output,
        norm, drop_band -> sub-band block -> reshape + look-ahead slice; fullsubnet/model.py:84-135): a GRU FullSubNet, other
        hidden sizes or output activations run without a tensor-algebra kernel of the host framework between the blocks
        (an activation other than ReLU is one elementwise launch on the block's output)."""
        L = _lib.lib()
        dev = noisy_mag.device
        B, _, F, T = noisy_mag.shape
        Tp = T + self.look_ahead
        dims = _lib.TrainDims(B, F, T, self.look_ahead, self.sb_num_neighbors, self.num_groups_in_drop_band,
                              _lib.NORM_TYPES[self.norm_type])
        dp = ctypes.byref(dims)
        fs, rows = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(L.fsn_train_rows(dp, ctypes.byref(fs), ctypes.byref(rows)))
        Fs, R = fs.value, rows.value
        Bp, Fp, Rp = (B + 15) // 16 * 16, (F + 15) // 16 * 16, (R + 15) // 16 * 16
        st = _lib.stream_ptr(dev)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        mag = noisy_mag.detach().reshape(B, F, T)
        mag = mag if mag.is_contiguous() else mag.contiguous()
        gws = _lib.workspace(L.fsn_train_glue_workspace_bytes(dp), dev)
        x_tm, mag_tm = new(Tp, Bp, Fp), new(Tp, Bp, Fp)
        _lib.check(L.fsn_train_fb_input(dp, _lib.dev_ptr(mag, "noisy_mag"), _lib.dev_ptr(x_tm), _lib.dev_ptr(mag_tm), Bp, Fp,
                                        gws.data_ptr(), gws.numel(), st))
        fb_out = self.fb_model.forward_time_major(x_tm, B, rows_out=True)            # [Tp, Bp, F]
        fb_out = fb_out if fb_out.is_contiguous() else fb_out.contiguous()
        sb_in, den = new(Tp, Rp, 32), new(L.fsn_train_den_elems(dp, Rp))
        _lib.check(L.fsn_train_sb_input(dp, _lib.dev_ptr(mag_tm), _lib.dev_ptr(fb_out), F, Bp, Fp, _lib.dev_ptr(sb_in), Rp,
                                        _lib.dev_ptr(den), gws.data_ptr(), gws.numel(), st))
        y2 = self.sb_model.forward_time_major(sb_in, R, rows_out=True)               # [Tp, Rp, 2]
        y2 = y2 if y2.is_contiguous() else y2.contiguous()
        mask = new(B, 2, Fs, T)
        _lib.check(L.fsn_train_mask_out(dp, _lib.dev_ptr(y2), Rp, _lib.dev_ptr(mask), st))
        return mask

    def _forward_composed(self, noisy_mag):
        """fullsubnet/model.py:72-136 operation by operation, for the configurations the fused kernels
        are not specialised for: the two SequenceModel blocks run on the HIP LSTM / GRU / GEMM kernels
        (inference or autograd), norms / unfold / concat / drop_band are tensor algebra."""
        from .base_model import BaseModel, look_ahead_pad
        norm = BaseModel().norm_wrapper(self.norm_type)
        x = look_ahead_pad(noisy_mag, self.look_ahead)
        B, C, F, T = x.size()
        fb_output = self.fb_model(norm(x).reshape(B, C * F, T)).reshape(B, 1, F, T)
        fb_unfolded = BaseModel.freq_unfold(fb_output, self.fb_num_neighbors).reshape(
            B, F, self.fb_num_neighbors * 2 + 1, T)
        noisy_unfolded = BaseModel.freq_unfold(x, self.sb_num_neighbors).reshape(B, F, self.sb_num_neighbors * 2 + 1, T)
        sb_input = norm(torch.cat([noisy_unfolded, fb_unfolded], dim=2))
        if B > 1 and self.num_groups_in_drop_band > 1:  # model.py:114-119 (quirk Q1: also in eval mode)
            sb_input = drop_band(sb_input.permute(0, 2, 1, 3), num_groups=self.num_groups_in_drop_band)
            F = sb_input.shape[2]
            sb_input = sb_input.permute(0, 2, 1, 3)
        K = (self.sb_num_neighbors * 2 + 1) + (self.fb_num_neighbors * 2 + 1)
        sb_mask = self.sb_model(sb_input.reshape(B * F, K, T))
        output = sb_mask.reshape(B, F, 2, T).permute(0, 2, 1, 3).contiguous()
        return output[:, :, :, self.look_ahead:]

    @torch.no_grad()
    def enhance(self, noisy, n_fft=512, hop_length=256, return_crm=False):
        """Whole path of inferencer.py:130-145 in one call: noisy [B, L] -> enhanced [B, L]
        (full-band masks for every sample, i.e. B independent utterances)."""
        from .acoustics.feature import hann_window
        assert noisy.dim() == 2
        y = noisy.contiguous()
        B, Ls = y.shape
        if not self._fused:  # composed configuration: the same stages as separate calls, no band dropping
            from .acoustics.feature import istft, stft
            from .acoustics.mask import decompress_cIRM
            mag, _, re, im = stft(y, n_fft, hop_length, n_fft, return_phase=False)
            groups, self.num_groups_in_drop_band = self.num_groups_in_drop_band, 1
            try:
                x4 = mag.unsqueeze(1)
                crm = self._forward_composed_rows(x4) if self._composed_rows_ok(x4) else self._forward_composed(x4)
            finally:
                self.num_groups_in_drop_band = groups
            m = decompress_cIRM(crm.permute(0, 2, 3, 1))
            out = istft((m[..., 0] * re - m[..., 1] * im, m[..., 1] * re + m[..., 0] * im), n_fft, hop_length, n_fft,
                        length=Ls, input_type="real_imag")
            return (out, crm) if return_crm else out
        L = _lib.lib()
        out = torch.empty_like(y)
        T = 1 + Ls // hop_length
        crm = torch.empty((B, 2, self.num_freqs, T), dtype=torch.float32, device=y.device) if return_crm else None
        ws = _lib.workspace(L.fsn_enhance_workspace_bytes(ctypes.byref(self._cfg), B, Ls, n_fft, hop_length), y.device)
        _lib.check(L.fsn_enhance(ctypes.byref(self._cfg), self.packed_weights().data_ptr(),
                                 _lib.dev_ptr(hann_window(n_fft, y.device)), _lib.dev_ptr(y, "noisy"), B, Ls, n_fft,
                                 hop_length, _lib.dev_ptr(out), _lib.dev_ptr(crm, allow_none=True), ws.data_ptr(),
                                 ws.numel(), _lib.stream_ptr(y.device)))
        return (out, crm) if return_crm else out
