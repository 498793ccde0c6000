"""Drop-in ``Model`` for recipes/dns_interspeech_2020/improved_fullsubnet/model.py (BASELINE config 5):
waveform in / waveform out, magnitude compression ``|X| ** fdrc``, a full-band LSTM on the first
F - 1 bins and *banded* sub-band LSTMs (finer-to-coarser: section i moves over its band in steps of
``num_center_freqs[i]`` bins and predicts that many complex mask bins per unit), the mask applied
without decompression.

STFT / iSTFT (any n_fft / hop: 512 / 128 at 16 kHz, 960 / 480 at 48 kHz) and every LSTM / Linear block
run on libfsn_hip.so; in inference (``Model._forward_kernels``, round 5) so do power law, last-bin slice, norms, the sections'
inputs and the mask products - no tensor-algebra launch of the host framework is left in a call; the tensor-algebra forward
below it is what autograd records in training and what the unit-sharded and odd configurations run.  Parameter names follow the reference (``fb_model.*``, ``sb_model.sb_models.{i}.*``).
"""
import torch
import torch.nn as nn
from torch.nn import functional

from .acoustics.feature import istft, stft
from .base_model import _hip_norm
from .sequence_model import SequenceModel as _SequenceModel

EPSILON = float(torch.finfo(torch.float32).eps)


def _round_up16(x):
    return (x + 15) // 16 * 16
_UNFOLD_INDEX = {}  # (band, centre, neighbours, bins, device) -> gather index of SubbandModel._freq_unfold


class SequenceModel(_SequenceModel):
    """improved_fullsubnet/model.py:25-121 (time-major nn.LSTM there; same parameters and results)."""

    def __init__(self, input_size, output_size, hidden_size, num_layers, bidirectional, sequence_model="GRU",
                 output_activate_function="Tanh", num_groups=4, mogrify_steps=5, dropout=0.0):
        if dropout:
            raise NotImplementedError("dropout between LSTM layers is not built (every use in the reference is 0)")
        super().__init__(input_size, output_size, hidden_size, num_layers, bidirectional, sequence_model,
                         output_activate_function)


class BaseModel(nn.Module):
    """improved_fullsubnet/model.py:124-216: note the eps of the offline norms is fp32 epsilon here."""

    @staticmethod
    def offline_laplace_norm(input, return_mu=False):
        if not return_mu:
            y = _hip_norm("offline_laplace_norm", input, eps=EPSILON)  # fsn_norm outside autograd, on the GPU
            if y is not None:
                return y
        mu = torch.mean(input, dim=list(range(1, input.dim())), keepdim=True)
        normed = input / (mu + EPSILON)
        return (normed, mu) if return_mu else normed

    @staticmethod
    def cumulative_laplace_norm(input):
        y = _hip_norm("cumulative_laplace_norm", input)
        if y is not None:
            return y
        B, C, F, T = input.size()
        x = input.reshape(B * C, F, T)
        cum = torch.cumsum(torch.sum(x, dim=1), dim=-1)
        count = torch.arange(F, F * T + 1, F, dtype=x.dtype, device=x.device).reshape(1, T)
        mean = (cum / count).reshape(B * C, 1, T)
        return (x / (mean + EPSILON)).reshape(B, C, F, T)

    @staticmethod
    def offline_gaussian_norm(input):
        y = _hip_norm("offline_gaussian_norm", input, eps=EPSILON)
        if y is not None:
            return y
        dims = list(range(1, input.dim()))
        mu = torch.mean(input, dim=dims, keepdim=True)
        std = torch.std(input, dim=dims, keepdim=True)
        return (input - mu) / (std + EPSILON)

    def norm_wrapper(self, norm_type: str):
        norms = {"offline_laplace_norm": self.offline_laplace_norm,
                 "cumulative_laplace_norm": self.cumulative_laplace_norm,
                 "offline_gaussian_norm": self.offline_gaussian_norm}
        if norm_type not in norms:
            raise NotImplementedError("You must set up a type of Norm. "
                                      "e.g. offline_laplace_norm, cumulative_laplace_norm, forgetting_norm, etc.")
        return norms[norm_type]


class SubBandSequenceWrapper(SequenceModel):
    """model.py:219-245: [B, N, 1, F_sub, T] -> [B, 2, N * centre, T]."""

    def forward(self, subband_input):
        batch_size, num_subband_units, num_channels, num_subband_freqs, num_frames = subband_input.shape
        assert num_channels == 1
        output = subband_input.reshape(batch_size * num_subband_units, num_subband_freqs, num_frames)
        output = super().forward(output)
        output = output.reshape(batch_size, num_subband_units, 2, -1, num_frames)
        output = output.permute(0, 2, 1, 3, 4).contiguous()
        return output.reshape(batch_size, 2, -1, num_frames)


class SubbandModel(BaseModel):
    def __init__(self, freq_cutoffs, sb_num_center_freqs, sb_num_neighbor_freqs, fb_num_center_freqs,
                 fb_num_neighbor_freqs, sequence_model, hidden_size, activate_function=False,
                 norm_type="offline_laplace_norm"):
        super().__init__()
        if len(freq_cutoffs) < 1:
            raise ValueError("the banded sub-band model needs at least two sections (one cut-off)")
        self.sb_models = nn.ModuleList([
            SubBandSequenceWrapper(input_size=(sc + sn * 2) + (fc + fn * 2), output_size=sc * 2,
                                   hidden_size=hidden_size, num_layers=2, sequence_model=sequence_model,
                                   bidirectional=False, output_activate_function=activate_function)
            for sc, sn, fc, fn in zip(sb_num_center_freqs, sb_num_neighbor_freqs, fb_num_center_freqs,
                                      fb_num_neighbor_freqs)])
        self.freq_cutoffs = freq_cutoffs
        self.sb_num_center_freqs = sb_num_center_freqs
        self.sb_num_neighbor_freqs = sb_num_neighbor_freqs
        self.fb_num_center_freqs = fb_num_center_freqs
        self.fb_num_neighbor_freqs = fb_num_neighbor_freqs
        self.norm = self.norm_wrapper(norm_type)
        self.norm_type = norm_type

    @staticmethod
    def _freq_unfold(input, lower_cutoff_freq=0, upper_cutoff_freq=20, num_center_freqs=1, num_neighbor_freqs=15):
        """model.py:315-400: unit u of the band [lower, upper) sees the bins
        lower + u c - n ... lower + (u + 1) c + n - 1 (c centre bins, n neighbours on each side), reflected
        at the two ends of the spectrum: [B, 1, F, T] -> [B, N = (upper - lower) / c, 1, c + 2 n, T]."""
        batch_size, num_channels, num_freqs, num_frames = input.shape
        assert num_channels == 1, "Only mono audio is supported."
        if (upper_cutoff_freq - lower_cutoff_freq) % num_center_freqs != 0:
            raise ValueError(
                f"The number of center frequencies should be divisible by the subband freqency interval. "
                f"Got {num_center_freqs=}, {upper_cutoff_freq=}, and {lower_cutoff_freq=}. "
                f"The subband freqency interval is {upper_cutoff_freq-lower_cutoff_freq}.")
        n, c = num_neighbor_freqs, num_center_freqs
        if lower_cutoff_freq != 0 and lower_cutoff_freq - n < 0:
            raise ValueError("an inner band needs num_neighbor_freqs bins below its lower cut-off")
        if upper_cutoff_freq != num_freqs and lower_cutoff_freq != 0 and upper_cutoff_freq + n > num_freqs:
            raise ValueError("an inner band needs num_neighbor_freqs bins above its upper cut-off")
        units = (upper_cutoff_freq - lower_cutoff_freq) // c
        dev = input.device
        key = (lower_cutoff_freq, upper_cutoff_freq, c, n, num_freqs, str(dev))
        idx = _UNFOLD_INDEX.get(key)
        if idx is None:  # built once per band and device (a dozen tiny launches otherwise, per section and call)
            idx = (lower_cutoff_freq - n + c * torch.arange(units, device=dev).reshape(units, 1)
                   + torch.arange(c + 2 * n, device=dev).reshape(1, -1))
            idx = idx.abs()
            idx = torch.where(idx > num_freqs - 1, 2 * (num_freqs - 1) - idx, idx)
            _UNFOLD_INDEX[key] = idx
        out = input[:, 0][:, idx, :]  # [B, N, c + 2n, T]
        return out.unsqueeze(2).contiguous()

    def _band(self, sb_idx, num_freqs):
        lower = 0 if sb_idx == 0 else self.freq_cutoffs[sb_idx - 1]
        upper = num_freqs if sb_idx == len(self.sb_models) - 1 else self.freq_cutoffs[sb_idx]
        return lower, upper

    def num_units(self, num_freqs):
        """Sub-band units of every section for a ``num_freqs``-bin input (model.py:315-400)."""
        return [(hi - lo) // c for (lo, hi), c in
                zip((self._band(i, num_freqs) for i in range(len(self.sb_models))), self.sb_num_center_freqs)]

    def _section_input(self, noisy_input, fb_output, sb_idx, units=None):
        """The normalised input of one section's sequence model [B, n, 1, F_sub, T] (model.py:402-440).  ``units = (lo,
        hi)`` restricts it to that range of the section's units; the norm statistics are always taken over the whole
        section, as in the reference.  None: more ranks than units, nothing of this section here."""
        lower, upper = self._band(sb_idx, noisy_input.size(2))
        noisy_subband = self._freq_unfold(noisy_input, lower, upper, self.sb_num_center_freqs[sb_idx],
                                          self.sb_num_neighbor_freqs[sb_idx])
        fb_subband = self._freq_unfold(fb_output, lower, upper, self.fb_num_center_freqs[sb_idx],
                                       self.fb_num_neighbor_freqs[sb_idx])
        sb_model_input = self.norm(torch.cat([noisy_subband, fb_subband], dim=-2))
        if units is not None:
            lo, hi = units
            if hi <= lo:
                return None
            sb_model_input = sb_model_input[:, lo:hi].contiguous()
        return sb_model_input

    def _section_prepared(self, noisy_input, fb_output, sb_idx, units=None):
        """The same input as ``_section_input``, written by fsn_improved_section_input straight into the layout the LSTM
        entries take - (h [T, Np, Ip] time-major, zero-padded, rows) - without forming the unfolded tensor: three launches
        instead of nine per section.  Inference on the GPU with the offline Laplace norm (the model's default); None when
        that does not apply (the caller then takes ``_section_input``) or when this rank owns no unit of the section."""
        from . import _lib
        from .sequence_model import _round_up
        if (self.norm_type != "offline_laplace_norm" or torch.is_grad_enabled() or not noisy_input.is_cuda
                or noisy_input.dtype != torch.float32):
            return None
        B, _, F, T = noisy_input.shape
        lower, upper = self._band(sb_idx, F)
        sc, sn = self.sb_num_center_freqs[sb_idx], self.sb_num_neighbor_freqs[sb_idx]
        fc, fn = self.fb_num_center_freqs[sb_idx], self.fb_num_neighbor_freqs[sb_idx]
        if (upper - lower) % sc or (upper - lower) % fc or (upper - lower) // sc != (upper - lower) // fc:
            return None  # the reference's own error path (model.py:341-346)
        n_units = (upper - lower) // sc
        lo, hi = (0, n_units) if units is None else units
        if hi <= lo:
            return None
        width = sc + 2 * sn + fc + 2 * fn
        rows = B * (hi - lo)
        Np, Ip = _round_up(rows, 16), _round_up(width, 16)
        if Ip > 240 or Np > 65535:  # beyond the kernel's tile (include/fsn_hip.h): through the unfolded tensor
            return None
        x = noisy_input.reshape(B, F, T).contiguous()
        f = fb_output.reshape(B, F, T).contiguous()
        h = torch.empty((T, Np, Ip), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        ws = _lib.workspace(L.fsn_improved_section_input_workspace_bytes(B, F), x.device)
        _lib.check(L.fsn_improved_section_input(
            _lib.dev_ptr(x, "noisy"), _lib.dev_ptr(f, "fb_output"), B, F, T, lower, upper, sc, sn, fc, fn, lo, hi, EPSILON,
            _lib.dev_ptr(h), Np, Ip, ws.data_ptr(), ws.numel(), _lib.stream_ptr(x.device)))
        return h, rows

    @staticmethod
    def _wrap_output(o, B, n_units):
        """[B n, 2 c, T] of the sequence model -> [B, 2, n c, T] (SubBandSequenceWrapper.forward, model.py:238-245)."""
        T = o.shape[-1]
        o = o.reshape(B, n_units, 2, -1, T).permute(0, 2, 1, 3, 4).contiguous()
        return o.reshape(B, 2, -1, T)

    def _section(self, noisy_input, fb_output, sb_idx, units=None):
        """One section (model.py:402-449)."""
        sb_model_input = self._section_input(noisy_input, fb_output, sb_idx, units)
        if sb_model_input is None:
            return noisy_input.new_zeros((noisy_input.size(0), 2, 0, noisy_input.size(-1)))
        return self.sb_models[sb_idx](sb_model_input)

    def _run_sections(self, noisy_input, fb_output, units, rows_out=False):
        """All sections, ``units[i]`` = None or the unit range of section i -> list of [B, 2, n_i c_i, T]; ``rows_out``
        (inference on prepared inputs only): the sections' outputs as their output layers wrote them, [T, Np_i, 2 c_i]
        time-major - what fsn_improved_mask_apply takes."""
        num = len(self.sb_models)
        if torch.is_grad_enabled() or not noisy_input.is_cuda:
            assert not rows_out
            return [self._section(noisy_input, fb_output, i, units[i]) for i in range(num)]
        # inference: the sections are independent two-layer stacks over the same frames.  When together they fill the
        # chip's workgroup sets (batches around 32 at 48 kHz) they run as ONE persistent launch of the group kernel with
        # a weight set per section (sequence_model.multi_forward -> fsn_lstm2_forward_multi)
        from .sequence_model import multi_forward, multi_plan, to_rows
        B, T = noisy_input.size(0), noisy_input.size(-1)
        n_units = self.num_units(noisy_input.size(2))
        span = [n_units[i] if units[i] is None else max(units[i][1] - units[i][0], 0) for i in range(num)]
        live = [i for i in range(num) if span[i] > 0]
        widths = [(sc + 2 * sn) + (fc + 2 * fn) for sc, sn, fc, fn in
                  zip(self.sb_num_center_freqs, self.sb_num_neighbor_freqs, self.fb_num_center_freqs, self.fb_num_neighbor_freqs)]
        if live and multi_plan([self.sb_models[i] for i in live], [(B * span[i], widths[i], T) for i in live]):
            prepared = []
            for i in live:
                p = self._section_prepared(noisy_input, fb_output, i, units[i])
                if p is None:  # another norm: through the unfolded tensor
                    x = self._section_input(noisy_input, fb_output, i, units[i]).reshape(B * span[i], widths[i], T)
                    h = torch.zeros((T, _round_up16(B * span[i]), _round_up16(widths[i])), dtype=torch.float32, device=x.device)
                    h[:, :B * span[i], :widths[i]] = x.permute(2, 0, 1)
                    p = (h, B * span[i])
                prepared.append(p)
            outs = multi_forward([self.sb_models[i] for i in live], prepared=prepared, rows_out=rows_out)
            if rows_out:  # [T, Np, 2 c] as the output layers wrote them, per section (None: no unit of it here)
                return [outs[live.index(i)] if i in live else None for i in range(num)]
            result = []
            for i in range(num):
                if i not in live:
                    result.append(noisy_input.new_zeros((B, 2, 0, T)))
                    continue
                o = outs[live.index(i)].reshape(B, span[i], 2, -1, T).permute(0, 2, 1, 3, 4).contiguous()
                result.append(o.reshape(B, 2, -1, T))
            return result
        # otherwise each one is a chain of small dependent launches (B x units rows only): they run concurrently on side
        # streams and join on the caller's.  Up to 64 rows per section (one or two utterances) every section is ONE
        # launch of the persistent chain kernel, of which the chip holds two large ones or a large and a small one at
        # a time (the gate of fsn_api.hip): two streams, large and small sections alternating, and the inputs of all
        # sections prepared before the first recurrence is issued (the host's launch time would otherwise sit
        # between the sections' starts)
        main = torch.cuda.current_stream(noisy_input.device)
        rows = [B * span[i] for i in range(num)]
        nstreams = 2 if max(rows) <= 64 else num
        if getattr(self, "_streams", None) is None or len(self._streams) != num:
            self._streams = [torch.cuda.Stream(noisy_input.device) for _ in range(num)]
        order = sorted(range(num), key=lambda i: -rows[i])
        stream_of = {i: self._streams[k % nstreams] for k, i in enumerate(order)}
        for st in self._streams[:nstreams]:
            st.wait_stream(main)
        inputs = {}
        for i in order:
            with torch.cuda.stream(stream_of[i]):
                if span[i] <= 0:
                    inputs[i] = None
                    continue
                inputs[i] = self._section_prepared(noisy_input, fb_output, i, units[i])
                if inputs[i] is None:  # wider than fsn_improved_section_input's tile / another norm: through the unfolded tensor
                    inputs[i] = self._section_input(noisy_input, fb_output, i, units[i])
                    if rows_out:  # ... into the entries' own layout, as the one-launch path above does
                        inputs[i] = (to_rows(inputs[i].reshape(B * span[i], widths[i], T)), B * span[i])
        subband_output = [None] * num
        for i in order:
            with torch.cuda.stream(stream_of[i]):
                if inputs[i] is None:
                    if rows_out:
                        subband_output[i] = None
                        continue
                    out = noisy_input.new_zeros((B, 2, 0, T))
                elif isinstance(inputs[i], tuple) and rows_out:
                    out = self.sb_models[i].forward_time_major(*inputs[i], rows_out=True)
                elif isinstance(inputs[i], tuple):
                    out = self._wrap_output(self.sb_models[i].forward_time_major(*inputs[i]), B, span[i])
                else:
                    out = self.sb_models[i](inputs[i])
            out.record_stream(main)
            subband_output[i] = out
        for st in self._streams[:nstreams]:
            main.wait_stream(st)
        return subband_output

    def forward_units(self, noisy_input, fb_output, rank, world):
        """Frequency-axis shard (BASELINE config 5): rank ``rank`` of ``world`` runs only its contiguous share of
        every section's units.  Returns one unit-major tensor ``[n_local_i, B, 2, c_i, T]`` per section (the
        layout ``parallel.gather_ragged`` re-assembles along the unit axis)."""
        from .parallel import shard_bounds
        B, _, num_freqs, T = noisy_input.shape
        units = [shard_bounds(n, rank, world) for n in self.num_units(num_freqs)]
        outs = self._run_sections(noisy_input, fb_output, units)
        return [o.reshape(B, 2, hi - lo, c, T).permute(2, 0, 1, 3, 4).contiguous()
                for o, (lo, hi), c in zip(outs, units, self.sb_num_center_freqs)]

    @staticmethod
    def assemble_units(full):
        """Unit-major section outputs ``[n_i, B, 2, c_i, T]`` (all units) -> the mask ``[B, 2, F - 1, T]``."""
        parts = []
        for o in full:
            n, B, _, c, T = o.shape
            parts.append(o.permute(1, 2, 0, 3, 4).reshape(B, 2, n * c, T))
        return torch.cat(parts, dim=-2)

    def forward(self, noisy_input, fb_output, unit_group=None):
        """model.py:402-449.  With ``unit_group`` (a torch.distributed process group, or True for the default one)
        the sub-band units are sharded across that group's ranks - every rank holds the whole ``noisy_input`` /
        ``fb_output``, runs its share of the units and one all-gather re-assembles the mask on every rank."""
        batch_size, num_channels, num_freqs, num_frames = noisy_input.size()
        assert num_channels == 1, "Only mono audio is supported."
        if unit_group is not None and unit_group is not False:
            if torch.is_grad_enabled():
                raise RuntimeError("the sub-band unit shard is an inference layout (its all-gather is not differentiable): "
                                   "call it under torch.no_grad(); training shards the batch (DDP)")
            import torch.distributed as dist
            from .parallel import gather_ragged
            group = None if unit_group is True else unit_group
            world = dist.get_world_size(group)
            if world > 1:
                local = self.forward_units(noisy_input, fb_output, dist.get_rank(group), world)
                return self.assemble_units(gather_ragged(local, self.num_units(num_freqs), group=group))
        return torch.cat(self._run_sections(noisy_input, fb_output, [None] * len(self.sb_models)), dim=-2)


class Model(BaseModel):
    def __init__(self, n_fft=512, hop_length=128, win_length=512, fdrc=0.5, num_freqs=257, freq_cutoffs=[20, 80],
                 sb_num_center_freqs=[1, 4, 8], sb_num_neighbor_freqs=[15, 15, 15], fb_num_center_freqs=[1, 4, 8],
                 fb_num_neighbor_freqs=[15, 15, 15], fb_hidden_size=512, sb_hidden_size=384, sequence_model="LSTM",
                 fb_output_activate_function=False, sb_output_activate_function=False,
                 norm_type="offline_laplace_norm"):
        super().__init__()
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.win_length = win_length
        self.fdrc = fdrc
        self.fb_model = SequenceModel(input_size=num_freqs - 1, output_size=num_freqs - 1, hidden_size=fb_hidden_size,
                                      num_layers=2, bidirectional=False, sequence_model=sequence_model,
                                      output_activate_function=fb_output_activate_function)
        self.sb_model = SubbandModel(freq_cutoffs=freq_cutoffs, sb_num_center_freqs=sb_num_center_freqs,
                                     sb_num_neighbor_freqs=sb_num_neighbor_freqs,
                                     fb_num_center_freqs=fb_num_center_freqs,
                                     fb_num_neighbor_freqs=fb_num_neighbor_freqs, hidden_size=sb_hidden_size,
                                     sequence_model=sequence_model, activate_function=sb_output_activate_function)
        self.norm = self.norm_wrapper(norm_type)

    def _persistent_chunk(self, B, T):
        """The largest batch <= B whose band sections run as one persistent launch (sequence_model.multi_plan), or None
        when B itself does / no batch in reach does."""
        from .sequence_model import multi_plan
        key = (B, T)
        cache = self.__dict__.setdefault("_chunk_cache", {})
        if key not in cache:
            sb = self.sb_model
            F = self.fb_model.input_size
            n_units = sb.num_units(F)
            widths = [(sc + 2 * sn) + (fc + 2 * fn) for sc, sn, fc, fn in
                      zip(sb.sb_num_center_freqs, sb.sb_num_neighbor_freqs, sb.fb_num_center_freqs, sb.fb_num_neighbor_freqs)]
            models = list(sb.sb_models)

            def fits(b):
                return multi_plan(models, [(b * n, w, T) for n, w in zip(n_units, widths)])

            best = None
            if not fits(B):
                for b in range(min(B - 1, 256), 7, -1):
                    if fits(b):
                        best = b
                        break
            cache[key] = best
        return cache[key]

    def _glue_on_kernels(self, y, unit_group):
        """Inference on the GPU with the configuration of the shipped use (fdrc 0.5 or 1, offline Laplace norm, LSTM blocks):
        the forward below without a single tensor-algebra launch of the host framework (round 5)."""
        sb = self.sb_model
        return (getattr(self, "glue_kernels", True) and unit_group is None and y.is_cuda and not torch.is_grad_enabled()
                and y.dtype == torch.float32 and self.fdrc in (0.5, 1.0) and sb.norm_type == "offline_laplace_norm"
                and self.norm == self.offline_laplace_norm and self.fb_model.cell == "LSTM"
                and all(m.cell == "LSTM" and m.output_size for m in sb.sb_models)
                and max(2 * c for c in sb.sb_num_center_freqs) <= 480)

    def _forward_kernels(self, y, mag, real, imag):
        """model.py:541-591 with every step between the transforms and the LSTM / Linear entries on kernels of the library:
        fsn_improved_front (mag ** fdrc, last bin left out), fsn_norm, fsn_bft_to_rows / fsn_rows_to_bft around the
        full-band model, fsn_improved_section_input per section, fsn_improved_mask_apply (the sections' outputs into the
        two masked planes, last bin zero).  Bit-identical to the tensor-algebra forward (tests/test_gpu_family.py)."""
        import ctypes
        from . import _lib
        from .sequence_model import from_rows, to_rows
        L = _lib.lib()
        dev = y.device
        st = _lib.stream_ptr(dev)
        B, F, T = mag.shape
        Fm = F - 1
        noisy_mag = torch.empty((B, 1, Fm, T), dtype=torch.float32, device=dev)
        _lib.check(L.fsn_improved_front(_lib.dev_ptr(mag, "mag"), B, F, T, 1 if self.fdrc == 0.5 else 0, _lib.dev_ptr(noisy_mag), st))
        fb_in = self.norm(noisy_mag).reshape(B, Fm, T)                                  # fsn_norm (contiguous: no copy)
        fb_rows = self.fb_model.forward_time_major(to_rows(fb_in), B, rows_out=True)     # [T, Np, Fm]
        fb_output = from_rows(fb_rows, B).reshape(B, 1, Fm, T)
        sb = self.sb_model
        outs = sb._run_sections(noisy_mag, fb_output, [None] * len(sb.sb_models), rows_out=True)
        secs = (_lib.MaskSection * len(outs))()
        keep, n = [], 0
        n_units = sb.num_units(Fm)
        for i, o in enumerate(outs):
            if o is None:
                continue
            assert o.dim() == 3, "the sections hand over [T, Np, 2 c] rows here"
            if o.stride(2) != 1 or o.stride(0) != o.shape[1] * o.stride(1):
                o = o.contiguous()
            keep.append(o)
            q = secs[n]
            q.o, q.Np, q.ld = o.data_ptr(), o.shape[1], o.stride(1)
            q.lower, q.units, q.center = sb._band(i, Fm)[0], n_units[i], sb.sb_num_center_freqs[i]
            n += 1
        er, ei = torch.empty_like(real), torch.empty_like(imag)
        _lib.check(L.fsn_improved_mask_apply(n, ctypes.byref(secs), _lib.dev_ptr(real, "real"), _lib.dev_ptr(imag, "imag"), B, F, T,
                                             _lib.dev_ptr(er), _lib.dev_ptr(ei), st))
        enhanced = istft((er, ei), self.n_fft, self.hop_length, self.win_length, length=y.size(-1), input_type="real_imag")
        return enhanced.unsqueeze(1)

    def forward(self, y, unit_group=None):
        """model.py:541-591: y [B, L] or [B, 1, L] -> enhanced [B, 1, L].  ``unit_group``: shard the sub-band units
        over that process group (every rank gets the same ``y`` and returns the same result; for fewer utterances
        than GPUs - otherwise shard the utterances, ``parallel.enhance_sharded``)."""
        ndim = y.dim()
        assert ndim in (2, 3), "Input must be 2D (B, T) or 3D tensor (B, 1, T)"
        if ndim == 3:
            assert y.size(1) == 1, "Input must be 2D (B, T) or 3D tensor (B, 1, T)"
            y = y.squeeze(1)
        if unit_group is None and y.is_cuda and not torch.is_grad_enabled():
            # the model has no cross-utterance term: a batch beyond what ONE persistent launch of the band sections holds
            # (32 utterances at 48 kHz) runs as chunks that do fit - 64 utterances: 55 ms as wavefronts, 2 x 21.5 as chunks
            c = self._persistent_chunk(y.size(0), 1 + y.size(-1) // self.hop_length)
            if c and y.size(0) > c:
                return torch.cat([self.forward(y[i:i + c]) for i in range(0, y.size(0), c)], dim=0)
        mag, _, real, imag = stft(y, self.n_fft, self.hop_length, self.win_length, return_phase=False)  # [B, F, T] each
        if self._glue_on_kernels(y, unit_group):
            return self._forward_kernels(y, mag, real, imag)
        noisy_mag = mag.unsqueeze(1) ** self.fdrc
        noisy_mag = noisy_mag[..., :-1, :]  # the last bin is left out (model.py:566) and masked with 0 below
        B, _, Fm, T = noisy_mag.shape
        fb_output = self.fb_model(self.norm(noisy_mag).reshape(B, Fm, T)).reshape(B, 1, Fm, T)
        cRM = self.sb_model(noisy_mag, fb_output, unit_group=unit_group)  # [B, 2, F - 1, T]
        cRM = functional.pad(cRM, (0, 0, 0, 1), mode="constant", value=0.0)
        # model.py:576-577: the mask multiplies real and imaginary parts separately (no complex product)
        enhanced = istft((cRM[:, 0] * real, cRM[:, 1] * imag), self.n_fft, self.hop_length, self.win_length,
                         length=y.size(-1), input_type="real_imag")
        return enhanced.unsqueeze(1)
