"""Chunked / frame-by-frame FullSubNet enhancement with carried state (the real-time use the model is
built for: causal LSTMs, cumulative norm, ``look_ahead`` frames of latency).

The reference only ships the offline loop (recipes/dns_interspeech_2020/inferencer.py:130-145 on a whole
utterance); ``StreamingEnhancer`` produces the same samples incrementally:

    enh = StreamingEnhancer(model)            # model.norm_type == "cumulative_laplace_norm"
    for chunk in chunks:                      # any chunk lengths
        out.append(enh.process(chunk))        # samples that are final so far (may be empty)
    out.append(enh.flush())                   # the rest; concatenation == offline result

What is carried between calls: the last input samples (STFT overlap), the running sums of the two
cumulative Laplace norms (base_model.py:221-251) and (h, c) of the four LSTM layers (one device blob
advanced by ``fsn_fullsubnet_stream_step``: the whole model on k more frames in one C call), the spectra
waiting for their mask (``look_ahead`` frames) and the last enhanced frame (overlap-add).  Every stage is
the same kernel as in the offline path, applied to the new frames only; tests/test_gpu_streaming.py checks
chunked == whole-utterance.  Configurations outside the fused kernels (other hidden sizes, neighbours ...)
run the same steps from the per-layer stateful entry (``fsn_lstm_layer_forward_state``) and tensor algebra.
"""
import ctypes

import torch

from . import _lib
from .acoustics.feature import istft, stft
from .acoustics.mask import decompress_cIRM
from .base_model import EPSILON, BaseModel
from .sequence_model import _round_up, linear_infer


def _packed_layers(block):
    """MFMA-fragment copies of the block's (padded) layer weights, rebuilt only when a parameter changed."""
    layers, fc = block._inference_weights()
    key = block._padded_key  # (data_ptr, version, device) of every parameter
    cached = getattr(block, "_stream_packed", None)
    if cached is None or cached[0] != key:
        L = _lib.lib()
        packed = []
        for w_ih, w_hh, b_ih, b_hh in layers:
            I, H = w_ih.shape[1], w_hh.shape[1]
            buf = _lib.workspace(L.fsn_lstm_layer_packed_bytes(I, H), w_ih.device)
            _lib.check(L.fsn_lstm_layer_pack(_lib.dev_ptr(w_ih), _lib.dev_ptr(w_hh), _lib.dev_ptr(b_ih), _lib.dev_ptr(b_hh),
                                             I, H, buf.data_ptr(), buf.numel(), _lib.stream_ptr(w_ih.device)))
            packed.append((buf, I, H))
        cached = (key, packed)
        block._stream_packed = cached
    return cached[1], fc


def _block_forward_state(block, x, state):
    """SequenceModel.forward for k more frames.  x [B, F, k]; state: list of (h, c) [Np, Hp] per layer
    (updated in place).  Returns [B, O, k]."""
    L = _lib.lib()
    B, F, k = x.shape
    H, Hp = block.hidden_size, _round_up(block.hidden_size, 64)
    Np, Ip = _round_up(B, 16), _round_up(F, 16)
    h = torch.zeros((k, Np, Ip), dtype=torch.float32, device=x.device)
    h[:, :B, :F] = x.permute(2, 0, 1)
    if block.cell == "GRU":  # sequence_model.py:59-66: the carried state is h alone (fsn_gru_layer_forward_state)
        layers, fc = block._inference_weights()
        for (w_ih, w_hh, b_ih, b_hh), (hs, _) in zip(layers, state):
            I, Hl = w_ih.shape[1], w_hh.shape[1]
            hseq = torch.empty((k, Np, Hl), dtype=torch.float32, device=x.device)
            ws = _lib.workspace(L.fsn_gru_layer_fwd_workspace_bytes(k, Np, I, Hl), x.device)
            _lib.check(L.fsn_gru_layer_forward_state(
                _lib.dev_ptr(h, "x"), h.shape[2], _lib.dev_ptr(w_ih), _lib.dev_ptr(w_hh), _lib.dev_ptr(b_ih),
                _lib.dev_ptr(b_hh), k, Np, I, Hl, _lib.dev_ptr(hseq), _lib.dev_ptr(hs), ws.data_ptr(), ws.numel(),
                _lib.stream_ptr(x.device)))
            h = hseq
    else:
        packed, fc = _packed_layers(block)
        for (buf, I, Hl), (hs, cs) in zip(packed, state):
            hseq = torch.empty((k, Np, Hl), dtype=torch.float32, device=x.device)
            ws = _lib.workspace(L.fsn_lstm_layer_state_workspace_bytes(k, Np, Hl), x.device)
            _lib.check(L.fsn_lstm_layer_forward_state(
                _lib.dev_ptr(h, "x"), h.shape[2], buf.data_ptr(), k, Np, I, Hl, _lib.dev_ptr(hseq), _lib.dev_ptr(hs),
                _lib.dev_ptr(cs), ws.data_ptr(), ws.numel(), _lib.stream_ptr(x.device)))
            h = hseq
    relu = block.output_activate_function == "ReLU"
    if fc is not None:
        o = linear_infer(h.reshape(k * Np, Hp), fc[0], fc[1], relu).reshape(k, Np, block.output_size)[:, :B]
    else:
        o, relu = h[:, :B, :H], False
    if block.output_activate_function and not relu:
        o = block.activate_function(o)
    return o.permute(1, 2, 0)


def _new_state(block, batch, device):
    Np, Hp = _round_up(batch, 16), _round_up(block.hidden_size, 64)
    return [(torch.zeros((Np, Hp), dtype=torch.float32, device=device),
             torch.zeros((Np, Hp), dtype=torch.float32, device=device)) for _ in range(block.num_layers)]


class StreamingEnhancer:
    def __init__(self, model, batch_size=1, n_fft=512, hop_length=256):
        if model.norm_type != "cumulative_laplace_norm":
            raise ValueError("streaming needs a causal norm: build the model with norm_type = 'cumulative_laplace_norm' "
                             "(fullsubnet/train_cumulativeLaplaceNorm.toml)")
        if hop_length * 2 != n_fft:
            raise NotImplementedError("streaming overlap-add is written for hop = n_fft / 2 (every FullSubNet TOML)")
        self.model = model.eval()
        self.B = batch_size
        self.n_fft, self.hop = n_fft, hop_length
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise _lib.FsnError("StreamingEnhancer needs the model on a ROCm device; this path has no CPU implementation")
        self.reset()

    # ------------------------------------------------------------------------------------------
    def reset(self):
        m, dev, B = self.model, self.device, self.B
        self._buf = torch.zeros((B, 0), dtype=torch.float32, device=dev)  # samples from global index _buf0 on
        self._buf0 = 0
        self._n_in = 0          # samples received
        self._t_next = 0        # next STFT frame to compute
        self._tau = 0           # model steps done (input frames incl. the look-ahead zeros at the end)
        if m._fused:  # the whole frame step is one C call (fsn_fullsubnet_stream_step) on one state blob
            nbytes = _lib.lib().fsn_fullsubnet_stream_state_bytes(ctypes.byref(m._cfg), B)
            if nbytes == 0:
                raise _lib.FsnError(_lib.lib().fsn_last_error().decode())
            self._state = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        else:         # composed configurations: per-block stateful layers + tensor algebra
            self._fb_state = _new_state(m.fb_model, B, dev)
            self._sb_state = _new_state(m.sb_model, B * m.num_freqs, dev)
            self._fb_sum = torch.zeros((B,), dtype=torch.float32, device=dev)
            self._sb_sum = torch.zeros((B * m.num_freqs,), dtype=torch.float32, device=dev)
        self._spec = []         # (re, im) [B, F, 1] of frames waiting for their mask, oldest first
        self._m_next = 0        # next output frame to mask
        self._prev = None       # enhanced (re, im) of frame _m_next - 1 (overlap-add partner)
        self._n_out = 0         # samples emitted
        self._closed = False

    # ---- model on k more (already look-ahead-ordered) input frames -------------------------------
    @torch.no_grad()
    def _model_steps(self, mag):
        """mag [B, F, k] -> compressed cIRM of model steps tau .. tau + k - 1, [B, 2, F, k]."""
        m = self.model
        B, F, k = mag.shape
        if m._fused:
            L = _lib.lib()
            x = mag.contiguous()
            out = torch.empty((B, 2, F, k), dtype=torch.float32, device=x.device)
            ws = _lib.workspace(L.fsn_fullsubnet_stream_workspace_bytes(ctypes.byref(m._cfg), B, k), x.device)
            _lib.check(L.fsn_fullsubnet_stream_step(
                ctypes.byref(m._cfg), m.packed_weights().data_ptr(), self._state.data_ptr(), self._state.numel(),
                self._tau, _lib.dev_ptr(x, "mag"), B, k, _lib.dev_ptr(out), ws.data_ptr(), ws.numel(),
                _lib.stream_ptr(x.device)))
            self._tau += k
            return out
        t = torch.arange(self._tau + 1, self._tau + k + 1, dtype=torch.float32, device=mag.device)  # frames so far
        # full-band cumulative Laplace norm (base_model.py:221-251) continued from the carried sum
        cum = self._fb_sum[:, None] + torch.cumsum(mag.sum(dim=1), dim=-1)
        fb_in = mag / ((cum / (F * t))[:, None, :] + EPSILON)
        self._fb_sum = cum[:, -1].contiguous()
        fb_out = _block_forward_state(m.fb_model, fb_in, self._fb_state)  # [B, F, k]
        noisy_unf = BaseModel.freq_unfold(mag.unsqueeze(1), m.sb_num_neighbors).reshape(B, F, -1, k)
        fb_unf = BaseModel.freq_unfold(fb_out.unsqueeze(1), m.fb_num_neighbors).reshape(B, F, -1, k)
        sb_in = torch.cat([noisy_unf, fb_unf], dim=2).reshape(B * F, -1, k)  # dim 1 of [B, F, C, T] is folded
        C = sb_in.shape[1]
        cum = self._sb_sum[:, None] + torch.cumsum(sb_in.sum(dim=1), dim=-1)
        sb_in = sb_in / ((cum / (C * t))[:, None, :] + EPSILON)
        self._sb_sum = cum[:, -1].contiguous()
        mask = _block_forward_state(m.sb_model, sb_in, self._sb_state)  # [B F, 2, k]
        self._tau += k
        return mask.reshape(B, F, 2, k).permute(0, 2, 1, 3)

    # ---- STFT of the frames that have all their samples -------------------------------------------
    def _new_frames(self, final):
        hop, half = self.hop, self.n_fft // 2
        if self._n_in <= half:  # frame 0 mirrors sample `half` on its left side
            return None
        # frame t >= 1 covers samples t hop - half .. t hop + half - 1; the last frame of the utterance
        # (T - 1 = n // hop) always needs the right-edge reflection and is only known at the end
        t_end = self._n_in // hop if final else (self._n_in - half) // hop
        if t_end < self._t_next:
            return None
        j0 = max(self._t_next - 2, 0)  # two frames of history keep the segment longer than the reflect pad
        a = j0 * hop - self._buf0
        b = (self._n_in if final else max(t_end * hop + half, half + 1)) - self._buf0
        seg = self._buf[:, a:b].contiguous()
        mag, _, re, im = stft(seg, self.n_fft, hop, self.n_fft, return_phase=False)
        lo, hi = self._t_next - j0, t_end - j0 + 1
        out = tuple(x[:, :, lo:hi].contiguous() for x in (mag, re, im))
        self._t_next = t_end + 1
        keep_from = max(self._t_next - 2, 0) * hop
        if keep_from > self._buf0:
            self._buf = self._buf[:, keep_from - self._buf0:].contiguous()
            self._buf0 = keep_from
        return out

    # ---- masks -> enhanced spectra -> overlap-add --------------------------------------------------
    def _emit(self, crm, total_length=None):
        """crm [B, 2, F, k]: masks of output frames _m_next .. _m_next + k - 1."""
        k = crm.shape[-1]
        re = torch.cat([s[0] for s in self._spec[:k]], dim=-1)
        im = torch.cat([s[1] for s in self._spec[:k]], dim=-1)
        del self._spec[:k]
        dm = decompress_cIRM(crm.permute(0, 2, 3, 1).contiguous())
        er = dm[..., 0] * re - dm[..., 1] * im
        ei = dm[..., 1] * re + dm[..., 0] * im
        first = self._m_next if self._prev is None else self._m_next - 1
        if self._prev is not None:
            er = torch.cat([self._prev[0], er], dim=-1)
            ei = torch.cat([self._prev[1], ei], dim=-1)
        self._m_next += k
        self._prev = (er[..., -1:].contiguous(), ei[..., -1:].contiguous())
        n_frames = er.shape[-1]
        length = (n_frames - 1) * self.hop if total_length is None else total_length - first * self.hop
        if length <= 0:
            return torch.zeros((self.B, 0), dtype=torch.float32, device=self.device)
        y = istft((er.contiguous(), ei.contiguous()), self.n_fft, self.hop, self.n_fft, length=length,
                  input_type="real_imag")
        skip = self._n_out - first * self.hop  # samples of this segment already emitted (none in steady state)
        y = y[:, skip:]
        self._n_out += y.shape[1]
        return y

    @torch.no_grad()
    def process(self, chunk):
        """chunk [B, n] (any n >= 0) -> enhanced samples that are final after this chunk, [B, n_ready]."""
        if self._closed:
            raise RuntimeError("StreamingEnhancer: flush() was called; reset() before a new stream")
        chunk = chunk.to(self.device, torch.float32)
        assert chunk.dim() == 2 and chunk.shape[0] == self.B
        self._buf = torch.cat([self._buf, chunk], dim=1)
        self._n_in += chunk.shape[1]
        frames = self._new_frames(final=False)
        if frames is None:
            return torch.zeros((self.B, 0), dtype=torch.float32, device=self.device)
        return self._advance(frames)

    def _advance(self, frames, tail_zeros=0, total_length=None):
        mag, re, im = frames
        k = mag.shape[-1]
        for i in range(k):
            self._spec.append((re[..., i:i + 1], im[..., i:i + 1]))
        if tail_zeros:
            mag = torch.cat([mag, mag.new_zeros(mag.shape[0], mag.shape[1], tail_zeros)], dim=-1)
        la = self.model.look_ahead
        tau0 = self._tau
        crm = self._model_steps(mag)  # model steps tau0 .. : step s is output frame s - la
        drop = max(la - tau0, 0)
        crm = crm[..., drop:]
        if crm.shape[-1] == 0:
            return torch.zeros((self.B, 0), dtype=torch.float32, device=self.device)
        return self._emit(crm.contiguous(), total_length)

    @torch.no_grad()
    def flush(self):
        """End of stream: the remaining frames (right-edge reflect padding), the ``look_ahead`` zero
        frames of fullsubnet/model.py:85 and the tail of the overlap-add.  Returns [B, rest]."""
        if self._closed:
            raise RuntimeError("StreamingEnhancer: already flushed")
        self._closed = True
        if self._n_in <= self.n_fft // 2:
            raise _lib.FsnError("stream shorter than n_fft / 2 + 1 samples (reflect padding), like torch.stft")
        frames = self._new_frames(final=True)
        la = self.model.look_ahead
        if frames is None:
            mag0 = torch.zeros((self.B, self.model.num_freqs, 0), dtype=torch.float32, device=self.device)
            frames = (mag0, mag0, mag0)
        return self._advance(frames, tail_zeros=la, total_length=self._n_in)
