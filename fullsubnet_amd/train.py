"""Training step of the FullSubNet recipe (recipes/dns_interspeech_2020/fullsubnet/trainer.py:33-76).

SURVEY §8 row A16: the four LSTM layers (99.9 % of the FLOPs of the step, forward and backward) run
on libfsn_hip.so through ``LstmLayerFunction`` (forward with saved activations + back-propagation
through time), the two output layers through ``LinearFunction``, the loss through ``mse_loss`` and
gradient clipping + Adam through ``optim.ClipAdam``.  For the shipped configuration (LSTM, offline Laplace norm,
``fb_num_neighbors = 0``) the glue between them - look-ahead pad, norms, the sub-band input (freq_unfold ++ full-band
output, normalised, band-dropped) forward and backward, the mask's reshape, the cIRM target - runs on
hand-written kernels too (``FullSubNetTrainFunction``, csrc/train_glue_kernels.hip): one autograd node for the whole
model, no tensor-algebra kernel of the host framework in the step.  Other configurations take the same graph with the
glue as autograd-tracked torch tensor algebra, so every gradient of the reference's graph is produced either way.
"""
import ctypes

import torch
import torch.nn.functional as functional

from . import _lib
from .acoustics.feature import drop_band, stft
from .acoustics.mask import build_complex_ideal_ratio_mask
from .optim import ClipAdam


class LstmLayerFunction(torch.autograd.Function):
    """One nn.LSTM layer (unidirectional, h0 = c0 = 0) on time-major input x [T, N, I] -> [T, N, H]."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        L = _lib.lib()
        T, N, I = x.shape
        H = w_hh.shape[1]
        Np, Ip = (N + 15) // 16 * 16, (I + 15) // 16 * 16
        xp = x
        if Np != N or Ip != I or not x.is_contiguous():
            xp = torch.zeros((T, Np, Ip), dtype=torch.float32, device=x.device)
            xp[:, :N, :I] = x
        w_ih_c, w_hh_c = w_ih.detach().contiguous(), w_hh.detach().contiguous()
        hseq = torch.empty((T, Np, H), dtype=torch.float32, device=x.device)
        save = _lib.workspace(L.fsn_lstm_layer_save_bytes(T, Np, H), x.device)
        ws = _lib.workspace(L.fsn_lstm_layer_fwd_workspace_bytes(T, Np, I, H), x.device)
        _lib.check(L.fsn_lstm_layer_forward(
            _lib.dev_ptr(xp, "x"), Ip, _lib.dev_ptr(w_ih_c, "w_ih"), _lib.dev_ptr(w_hh_c, "w_hh"),
            _lib.dev_ptr(b_ih.detach().contiguous(), "b_ih"), _lib.dev_ptr(b_hh.detach().contiguous(), "b_hh"),
            T, Np, I, H, _lib.dev_ptr(hseq), save.data_ptr(), save.numel(), ws.data_ptr(), ws.numel(),
            _lib.stream_ptr(x.device)))
        ctx.save_for_backward(xp, w_ih_c, w_hh_c, hseq, save)
        ctx.dims = (T, N, I, H, Np, Ip)
        return hseq[:, :N]

    @staticmethod
    def backward(ctx, dh):
        L = _lib.lib()
        xp, w_ih, w_hh, hseq, save = ctx.saved_tensors
        T, N, I, H, Np, Ip = ctx.dims
        dhp = dh
        if Np != N or not dh.is_contiguous():
            dhp = torch.zeros((T, Np, H), dtype=torch.float32, device=dh.device)
            dhp[:, :N] = dh
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty((T, Np, Ip), dtype=torch.float32, device=dh.device) if need_dx else None
        dw_ih = torch.empty_like(w_ih)
        dw_hh = torch.empty_like(w_hh)
        db = torch.empty((4 * H,), dtype=torch.float32, device=dh.device)
        ws = _lib.workspace(L.fsn_lstm_layer_bwd_workspace_bytes(T, Np, I, H), dh.device)
        _lib.check(L.fsn_lstm_layer_backward(
            _lib.dev_ptr(dhp, "dh"), _lib.dev_ptr(xp), Ip, _lib.dev_ptr(w_ih), _lib.dev_ptr(w_hh), T, Np, I, H,
            _lib.dev_ptr(hseq), save.data_ptr(), _lib.dev_ptr(dx, allow_none=True), Ip, _lib.dev_ptr(dw_ih),
            _lib.dev_ptr(dw_hh), _lib.dev_ptr(db), ws.data_ptr(), ws.numel(), _lib.stream_ptr(dh.device)))
        return (dx[:, :N, :I] if need_dx else None), dw_ih, dw_hh, db, db.clone()


# "f16" / "bf16": the arithmetic of torch.autocast (16-bit matrix-core operands, fp32 accumulation, everything stored in fp32);
# "+s16" (Model.train_saves = "16", _lib.ARITH_SAVES16): the activated gates that BPTT re-reads saved in that 16-bit type too -
# what the vendor LSTM keeps in its reserve space under autocast - half the save traffic of the two persistent launches
TRAIN_ARITH = ("f32", "f16", "bf16", "f16+s16", "bf16+s16")


def train_arith_of(model):
    """The `arith` string of a model's training step: Model.train_arithmetic ("f32" | "f16" | "bf16"; Trainer sets it from
    use_amp) and Model.train_saves ("32" | "16", default "16" under a 16-bit arithmetic: tests/test_gpu_amp.py holds that
    step to the reference's own fp16-autocast step)."""
    arith = getattr(model, "train_arithmetic", "f32")
    if arith in ("f16", "bf16") and str(getattr(model, "train_saves", DEFAULT_TRAIN_SAVES)) == "16":
        arith += "+s16"
    return arith


DEFAULT_TRAIN_SAVES = "16"


class Lstm2Function(torch.autograd.Function):
    """nn.LSTM(num_layers=2) (unidirectional, h0 = c0 = 0, both layers H wide) on time-major x [T, N, I] -> [T, N, H]:
    fsn_lstm2_forward_train (the full-band shape - H = 512, up to 64 rows - and the sub-band shape - H = 384, 96+ row
    tiles - each as ONE persistent launch for both layers and all steps) and fsn_lstm2_backward (the sub-band shape's
    back-propagation through time as one persistent launch as well).  `arith`: "f32", or "f16" / "bf16" = the
    arithmetic of torch.autocast (fullsubnet/trainer.py:56): 16-bit matrix-core operands, fp32 accumulation, on the
    sub-band shape's kernels; the gradient coming in is then expected to carry the caller's loss scale (GradScaler)."""

    @staticmethod
    def forward(ctx, x, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1, arith="f32"):
        L = _lib.lib()
        if arith not in TRAIN_ARITH:
            raise _lib.FsnError(f"training arithmetic {arith!r}: one of {TRAIN_ARITH}")
        ctx.arith = _lib.ARITH[arith]
        T, N, I = x.shape
        H = w_hh0.shape[1]
        Np, Ip = (N + 15) // 16 * 16, (I + 15) // 16 * 16
        xp = x
        if Np != N or Ip != I or not x.is_contiguous():
            xp = torch.zeros((T, Np, Ip), dtype=torch.float32, device=x.device)
            xp[:, :N, :I] = x
        ws_ = [t.detach().contiguous() for t in (w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1)]
        hseq0 = torch.empty((T, Np, H), dtype=torch.float32, device=x.device)
        hseq1 = torch.empty((T, Np, H), dtype=torch.float32, device=x.device)
        nsave = L.fsn_lstm_layer_save_bytes(T, Np, H)
        save0, save1 = _lib.workspace(nsave, x.device), _lib.workspace(nsave, x.device)
        ws = _lib.workspace(L.fsn_lstm2_train_workspace_bytes(T, Np, I, H, ctx.arith), x.device)
        _lib.check(L.fsn_lstm2_forward_train(
            _lib.dev_ptr(xp, "x"), Ip, *[_lib.dev_ptr(t) for t in ws_], T, Np, I, H, _lib.dev_ptr(hseq0),
            _lib.dev_ptr(hseq1), save0.data_ptr(), save1.data_ptr(), nsave, ws.data_ptr(), ws.numel(), ctx.arith,
            _lib.stream_ptr(x.device)))
        ctx.save_for_backward(xp, ws_[0], ws_[1], ws_[4], ws_[5], hseq0, hseq1, save0, save1)
        ctx.dims = (T, N, I, H, Np, Ip)
        return hseq1[:, :N]

    @staticmethod
    def backward(ctx, dh):
        L = _lib.lib()
        xp, w_ih0, w_hh0, w_ih1, w_hh1, hseq0, hseq1, save0, save1 = ctx.saved_tensors
        T, N, I, H, Np, Ip = ctx.dims
        dev = dh.device
        dhp = dh
        if Np != N or not dh.is_contiguous():
            dhp = torch.zeros((T, Np, H), dtype=torch.float32, device=dev)
            dhp[:, :N] = dh
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty((T, Np, Ip), dtype=torch.float32, device=dev) if need_dx else None
        dw_ih0, dw_hh0, dw_ih1, dw_hh1 = (torch.empty_like(w) for w in (w_ih0, w_hh0, w_ih1, w_hh1))
        db0 = torch.empty((4 * H,), dtype=torch.float32, device=dev)
        db1 = torch.empty((4 * H,), dtype=torch.float32, device=dev)
        ws = _lib.workspace(L.fsn_lstm2_bwd_workspace_bytes(T, Np, I, H, ctx.arith), dev)
        _lib.check(L.fsn_lstm2_backward(
            _lib.dev_ptr(dhp, "dh"), _lib.dev_ptr(xp), Ip, _lib.dev_ptr(w_ih0), _lib.dev_ptr(w_hh0), _lib.dev_ptr(w_ih1),
            _lib.dev_ptr(w_hh1), T, Np, I, H, _lib.dev_ptr(hseq0), _lib.dev_ptr(hseq1), save0.data_ptr(), save1.data_ptr(),
            _lib.dev_ptr(dx, allow_none=True), Ip, _lib.dev_ptr(dw_ih0), _lib.dev_ptr(dw_hh0), _lib.dev_ptr(db0),
            _lib.dev_ptr(dw_ih1), _lib.dev_ptr(dw_hh1), _lib.dev_ptr(db1), ws.data_ptr(), ws.numel(), ctx.arith,
            _lib.stream_ptr(dev)))
        return ((dx[:, :N, :I] if need_dx else None), dw_ih0, dw_hh0, db0, db0.clone(), dw_ih1, dw_hh1, db1, db1.clone(),
                None)


class GruLayerFunction(torch.autograd.Function):
    """One nn.GRU layer (unidirectional, h0 = 0) on time-major input x [T, N, I] -> [T, N, H]
    (fsn_gru_layer_forward with saved r, z, n, hn + fsn_gru_layer_backward)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        L = _lib.lib()
        T, N, I = x.shape
        H = w_hh.shape[1]
        Np, Ip = (N + 15) // 16 * 16, (I + 15) // 16 * 16
        xp = x
        if Np != N or Ip != I or not x.is_contiguous():
            xp = torch.zeros((T, Np, Ip), dtype=torch.float32, device=x.device)
            xp[:, :N, :I] = x
        w_ih_c, w_hh_c = w_ih.detach().contiguous(), w_hh.detach().contiguous()
        hseq = torch.empty((T, Np, H), dtype=torch.float32, device=x.device)
        save = _lib.workspace(L.fsn_gru_layer_save_bytes(T, Np, H), x.device)
        ws = _lib.workspace(L.fsn_gru_layer_fwd_workspace_bytes(T, Np, I, H), x.device)
        _lib.check(L.fsn_gru_layer_forward(
            _lib.dev_ptr(xp, "x"), Ip, _lib.dev_ptr(w_ih_c, "w_ih"), _lib.dev_ptr(w_hh_c, "w_hh"),
            _lib.dev_ptr(b_ih.detach().contiguous(), "b_ih"), _lib.dev_ptr(b_hh.detach().contiguous(), "b_hh"),
            T, Np, I, H, _lib.dev_ptr(hseq), save.data_ptr(), save.numel(), ws.data_ptr(), ws.numel(),
            _lib.stream_ptr(x.device)))
        ctx.save_for_backward(xp, w_ih_c, w_hh_c, hseq, save)
        ctx.dims = (T, N, I, H, Np, Ip)
        return hseq[:, :N]

    @staticmethod
    def backward(ctx, dh):
        L = _lib.lib()
        xp, w_ih, w_hh, hseq, save = ctx.saved_tensors
        T, N, I, H, Np, Ip = ctx.dims
        dhp = dh
        if Np != N or not dh.is_contiguous():
            dhp = torch.zeros((T, Np, H), dtype=torch.float32, device=dh.device)
            dhp[:, :N] = dh
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty((T, Np, Ip), dtype=torch.float32, device=dh.device) if need_dx else None
        dw_ih = torch.empty_like(w_ih)
        dw_hh = torch.empty_like(w_hh)
        db_ih = torch.empty((3 * H,), dtype=torch.float32, device=dh.device)
        db_hh = torch.empty((3 * H,), dtype=torch.float32, device=dh.device)
        ws = _lib.workspace(L.fsn_gru_layer_bwd_workspace_bytes(T, Np, I, H), dh.device)
        _lib.check(L.fsn_gru_layer_backward(
            _lib.dev_ptr(dhp, "dh"), _lib.dev_ptr(xp), Ip, _lib.dev_ptr(w_ih), _lib.dev_ptr(w_hh), T, Np, I, H,
            _lib.dev_ptr(hseq), save.data_ptr(), _lib.dev_ptr(dx, allow_none=True), Ip, _lib.dev_ptr(dw_ih),
            _lib.dev_ptr(dw_hh), _lib.dev_ptr(db_ih), _lib.dev_ptr(db_hh), ws.data_ptr(), ws.numel(),
            _lib.stream_ptr(dh.device)))
        return (dx[:, :N, :I] if need_dx else None), dw_ih, dw_hh, db_ih, db_hh


class LinearFunction(torch.autograd.Function):
    """nn.Linear (+ optional ReLU) on x [..., I] -> [..., O] through fsn_linear_forward / _backward."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        L = _lib.lib()
        lead, I, O = x.shape[:-1], x.shape[-1], w.shape[0]
        R = x.numel() // I
        Ip = (I + 15) // 16 * 16
        x2 = x.reshape(R, I)
        if Ip != I or not x2.is_contiguous():
            xp = torch.zeros((R, Ip), dtype=torch.float32, device=x.device)
            xp[:, :I] = x2
        else:
            xp = x2
        wc = w.detach().contiguous()
        y = torch.empty((R, O), dtype=torch.float32, device=x.device)
        ws = _lib.workspace(L.fsn_linear_workspace_bytes(R, I, O), x.device)
        _lib.check(L.fsn_linear_forward(_lib.dev_ptr(xp, "x"), Ip, _lib.dev_ptr(wc, "w"),
                                        _lib.dev_ptr(b.detach().contiguous(), "b"), R, I, O, 1 if relu else 0,
                                        _lib.dev_ptr(y), ws.data_ptr(), ws.numel(), _lib.stream_ptr(x.device)))
        ctx.save_for_backward(xp, wc, y if relu else None)
        ctx.dims = (lead, R, I, O, Ip, relu)
        return y.reshape(*lead, O)

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        xp, w, y = ctx.saved_tensors
        lead, R, I, O, Ip, relu = ctx.dims
        Op = (O + 15) // 16 * 16
        dy2 = dy.reshape(R, O)
        if relu:
            dy2 = dy2 * (y > 0)
        dyp = torch.zeros((R, Op), dtype=torch.float32, device=dy.device)
        dyp[:, :O] = dy2
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty((R, Ip), dtype=torch.float32, device=dy.device) if need_dx else None
        dw = torch.empty_like(w)
        db = torch.empty((O,), dtype=torch.float32, device=dy.device)
        ws = _lib.workspace(L.fsn_linear_workspace_bytes(R, I, O), dy.device)
        _lib.check(L.fsn_linear_backward(_lib.dev_ptr(dyp, "dy"), Op, _lib.dev_ptr(xp), Ip, _lib.dev_ptr(w), R, I, O,
                                         _lib.dev_ptr(dx, allow_none=True), Ip, _lib.dev_ptr(dw), _lib.dev_ptr(db),
                                         ws.data_ptr(), ws.numel(), _lib.stream_ptr(dy.device)))
        return (dx[:, :I].reshape(*lead, I) if need_dx else None), dw, db, None


LSTM2_CHUNK_ROWS = 2048  # one persistent launch of the group kernels: 32 clusters of 64 rows (two workgroups per CU)
LSTM2_MIN_PIECE_ROWS = 1536  # ... and the fewest rows the library plans one for (96 row tiles)


def lstm2_train_chunks(T, N, I, H, pad_to_32=True, pad_small=False):
    """Rows per piece, number of pieces - or None - for a two-layer LSTM stack with MORE rows than one persistent launch of
    the training kernels holds (fsn_lstm2_forward_train / fsn_lstm2_backward: the group kernels take 96 - 128 row tiles in
    whole 64-row clusters).  The rows of a stack are independent sequences (sequence_model.py:52-58), so N rows run as
    equal pieces of whole clusters; the weight gradients of the pieces add up in autograd.  Fast FullSubNet's bottleneck
    (fast_fullsubnet/model.py:66-74; train_shrinkSize2.toml:52: 72 utterances x 64 bands = 4608 rows) is 3 x 1536."""
    if I > 32 or N < 1:
        return None
    L = _lib.lib()
    n0 = -(-N // LSTM2_CHUNK_ROWS)
    for n in (n0, n0 + 1):
        rows = (-(-N // n) + 63) // 64 * 64
        # e.g. 20 utterances x 128 bins = 2560 rows: two pieces of 1536 (the second one 2/3 full); ``pad_small``: fewer rows
        # than the smallest launch as ONE zero-padded piece - what pays under the 16-bit arithmetic (8 utterances: 32 ms step
        # by step in fp32 against 11), not in fp32 (a launch of 24 clusters costs what it costs whatever it holds)
        if N > LSTM2_MIN_PIECE_ROWS or pad_small:
            rows = max(rows, LSTM2_MIN_PIECE_ROWS)
        if L.fsn_lstm2_train_is_persistent(T, rows, 32 if pad_to_32 else I, H) == 1:
            return rows, n
    return None


def rows_to_pieces(x, rows, n):
    """x [T, N, W] time-major (contiguous, on the GPU) -> [n, T, rows, W]: piece k holds rows [k rows, (k + 1) rows), zeros
    beyond N (fsn_train_rows_pieces: no kernel of the host framework in the step)."""
    T, N, W = x.shape
    out = torch.empty((n, T, rows, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().fsn_train_rows_pieces(_lib.dev_ptr(x, "rows"), _lib.dev_ptr(out), T, N, W, rows, n, 1, _lib.stream_ptr(x.device)))
    return out


def pieces_to_rows(pieces, N):
    """The inverse: pieces [n, T, rows, W] (ONE contiguous tensor) -> [T, N, W] (rows beyond N dropped)."""
    n, T, rows, W = pieces.shape
    out = torch.empty((T, N, W), dtype=torch.float32, device=pieces.device)
    _lib.check(_lib.lib().fsn_train_rows_pieces(_lib.dev_ptr(pieces, "pieces"), _lib.dev_ptr(out), T, N, W, rows, n, 0,
                                                _lib.stream_ptr(pieces.device)))
    return out


def lstm2_rows_chunked(x_tn, params, arith, rows, n):
    """x_tn [T, N, I] (I <= 32) -> [T, N, H] through ``n`` Lstm2Function calls of ``rows`` rows each; the input is zero-padded
    to 32 columns (the group kernels' two K chunks; W_ih0 gets zero columns) and to n * rows rows."""
    import torch.nn.functional as F
    T, N, I = x_tn.shape
    w_ih0, rest = params[0], params[1:]
    xp = F.pad(x_tn, (0, 32 - I, 0, n * rows - N))
    w_ih0p = F.pad(w_ih0, (0, 32 - I))
    outs = [Lstm2Function.apply(xp[:, k * rows:(k + 1) * rows], w_ih0p, *rest, arith) for k in range(n)]
    return torch.cat(outs, dim=1)[:, :N]


def lstm_stack(x_tn, lstm, arith="f32"):
    """Two stacked layers of an nn.LSTM parameter container on time-major x [T, N, I]."""
    h = x_tn
    if lstm.num_layers == 2 and lstm.weight_hh_l0.shape == lstm.weight_hh_l1.shape:
        return Lstm2Function.apply(h, *[getattr(lstm, f"{n}_l{k}") for k in (0, 1)
                                        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")], arith)
    for k in range(lstm.num_layers):
        h = LstmLayerFunction.apply(h, getattr(lstm, f"weight_ih_l{k}"), getattr(lstm, f"weight_hh_l{k}"),
                                    getattr(lstm, f"bias_ih_l{k}"), getattr(lstm, f"bias_hh_l{k}"))
    return h


def _freq_unfold(x, n):
    """audio_zen/model/base_model.py:14-46 (same ops)."""
    B, C, F, T = x.shape
    if n <= 0:
        return x.permute(0, 2, 1, 3).reshape(B, F, C, 1, T)
    out = x.reshape(B * C, 1, F, T)
    out = functional.pad(out, [0, 0, n, n], mode="reflect")
    out = functional.unfold(out, kernel_size=(2 * n + 1, T))
    out = out.reshape(B, C, 2 * n + 1, T, F)
    return out.permute(0, 4, 1, 2, 3).contiguous()


def _norm(x, norm_type):
    """offline / cumulative Laplace norm, base_model.py:204-251."""
    if norm_type == "offline_laplace_norm":
        mu = torch.mean(x, dim=list(range(1, x.dim())), keepdim=True)
        return x / (mu + 1e-5)
    B, C, F, T = x.shape
    xr = x.reshape(B * C, F, T)
    cum = torch.cumsum(torch.sum(xr, dim=1), dim=-1)
    count = torch.arange(F, F * T + 1, F, dtype=x.dtype, device=x.device).reshape(1, T)
    mean = (cum / count).reshape(B * C, 1, T)
    return (xr / (mean + torch.finfo(torch.float32).eps)).reshape(B, C, F, T)


_INDEX_CACHE = {}


def _sb_indices(B, F, n, groups, device):
    """Row list of the sub-band model after drop_band (feature.py:327-345 order: group g holds samples
    g::G at bins g:F':G), the reflect-padded window bins of every row (base_model.py:31-44) and the
    number of (unit, window row) pairs that hit each bin (for the analytic mean of the unfolded tensor)."""
    key = (B, F, n, groups, str(device))
    if key not in _INDEX_CACHE:
        f_all = torch.arange(F)
        k = torch.arange(-n, n + 1)
        src = (f_all[:, None] + k[None, :]).abs()
        src = torch.where(src >= F, 2 * (F - 1) - src, src)  # [F, 2n+1]
        mult = torch.bincount(src.reshape(-1), minlength=F).to(torch.float32)
        if B > 1 and groups > 1:
            Fd = F - F % groups
            rb = torch.cat([torch.arange(g, B, groups).repeat_interleave(len(range(g, Fd, groups))) for g in range(groups)])
            rf = torch.cat([torch.arange(g, Fd, groups).repeat(len(range(g, B, groups))) for g in range(groups)])
            Fs = Fd // groups
        else:
            rb, rf, Fs = torch.arange(B).repeat_interleave(F), f_all.repeat(B), F
        _INDEX_CACHE[key] = (rb.to(device), rf.to(device), src[rf].to(device), mult.to(device), Fs)
    return _INDEX_CACHE[key]


class SubbandInputOffline(torch.autograd.Function):
    """fullsubnet/model.py:98-125 for norm_type = offline_laplace_norm without materialising the
    unfolded tensor twice: freq_unfold(noisy, n) ++ fb_output, divided by the per-utterance mean of
    the FULL [F, 2n+2, T'] tensor (computed analytically from per-bin sums), restricted to the rows
    drop_band keeps.  Gradients to fb_output and, when asked for, to the noisy magnitude (directly and through the
    mean)."""

    @staticmethod
    def forward(ctx, x, fb_out, n, groups):
        B, _, F, Tp = x.shape
        rb, rf, win_idx, mult, Fs = _sb_indices(B, F, n, groups, x.device)
        raw = torch.cat([x[rb[:, None], 0, win_idx, :], fb_out[rb, 0, rf, :][:, None, :]], dim=1)  # [R, 2n+2, Tp]
        count = float(F * (2 * n + 2) * Tp)
        mu = ((x[:, 0].sum(dim=2) * mult[None, :]).sum(dim=1) + fb_out.sum(dim=(1, 2, 3))) / count  # [B]
        den = (mu + 1e-5)[rb]  # [R]
        ctx.save_for_backward(raw, den, rb, rf, win_idx, mult)
        ctx.dims = (B, F, Tp, Fs, count)
        return raw / den[:, None, None]

    @staticmethod
    def backward(ctx, dy):
        raw, den, rb, rf, win_idx, mult = ctx.saved_tensors
        B, F, Tp, Fs, count = ctx.dims
        d_fb = torch.zeros((B, 1, F, Tp), dtype=dy.dtype, device=dy.device)
        d_fb[rb, 0, rf, :] = dy[:, -1, :] / den[:, None]
        # d mu_b = - sum_{rows of b} dy * raw / (mu + eps)^2 ; every fb_out element has weight 1/count in mu
        per_row = (dy * raw).sum(dim=(1, 2)) / (den * den)
        d_mu = torch.zeros((B,), dtype=dy.dtype, device=dy.device)
        d_mu[rb[::Fs]] = -per_row.reshape(-1, Fs).sum(dim=1)  # a sample's rows are one contiguous block
        d_fb += (d_mu / count)[:, None, None, None]
        d_x = None
        if ctx.needs_input_grad[0]:
            # x[b, f] feeds every kept row whose window holds bin f (directly) and the mean (with its multiplicity)
            d_x = torch.zeros((B, F, Tp), dtype=dy.dtype, device=dy.device)
            d_x.index_put_((rb[:, None].expand_as(win_idx), win_idx), dy[:, :-1, :] / den[:, None, None], accumulate=True)
            d_x += (d_mu / count)[:, None, None] * mult[None, :, None]
            d_x = d_x[:, None]
        return d_x, d_fb, None, None


# FullSubNetTrainFunction.backward: the sub-band weight-gradient products on a second stream beside the full-band model's
# backward (False: everything on the caller's stream, in order)
OVERLAP_WEIGHT_PRODUCTS = True
_SIDE = {}


def _side_stream(device, which=0):
    key = (str(device), which)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device)
    return _SIDE[key]


_ONE = {}


def _one(device):
    """A device scalar 1.0 (made once per device: no fill kernel per step)."""
    key = str(device)
    if key not in _ONE:
        _ONE[key] = torch.ones((), dtype=torch.float32, device=device)
    return _ONE[key]


def _dup(t, out=None):
    """A second buffer with the same values through the library (b_ih and b_hh receive the same gradient, and the fused
    optimizer clips gradients in place: they must not share storage)."""
    out = torch.empty_like(t) if out is None else out
    _lib.check(_lib.lib().fsn_scale_by_scalar(_lib.dev_ptr(t), _lib.dev_ptr(_one(t.device)), _lib.dev_ptr(out), t.numel(),
                                              _lib.stream_ptr(t.device)))
    return out


class FullSubNetTrainFunction(torch.autograd.Function):
    """fullsubnet/model.py:72-136 as ONE autograd node for the shipped configuration: noisy_mag [B, 1, F, T] ->
    compressed mask [B, 2, Fs, T] (band-dropped, look-ahead frames removed).  Forward: fsn_train_fb_input ->
    fsn_lstm2_forward_train + fsn_linear_forward (full-band model, ReLU) -> fsn_train_sb_input -> fsn_lstm2_forward_train +
    fsn_linear_forward (sub-band model) -> fsn_train_mask_out; backward: the mirror image.  Every tensor in between lives
    in the time-major, zero-padded layouts the LSTM entries take, written by the kernels themselves (torch.empty only)."""

    @staticmethod
    def forward(ctx, noisy_mag, look_ahead, nb, groups, arith, norm_type, *params):
        L = _lib.lib()
        if arith not in TRAIN_ARITH:
            raise _lib.FsnError(f"training arithmetic {arith!r}: one of {TRAIN_ARITH}")
        if norm_type not in _lib.NORM_TYPES:
            raise _lib.FsnError(f"fused training graph: norm_type {norm_type!r}: one of {sorted(_lib.NORM_TYPES)}")
        ar = _lib.ARITH[arith]
        dev = noisy_mag.device
        B, _, F, T = noisy_mag.shape
        # drop_band's own check (feature.py:317-319), reached for every batch of more than one utterance (model.py:114)
        assert B == 1 or B > groups, (
            f"Batch size = {B}, num_groups = {groups}. The batch size should larger than the num_groups.")
        Tp = T + look_ahead
        dims = _lib.TrainDims(B, F, T, look_ahead, nb, groups, _lib.NORM_TYPES[norm_type])
        dp = ctypes.byref(dims)
        fs, rows = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(L.fsn_train_rows(dp, ctypes.byref(fs), ctypes.byref(rows)))
        Fs, R = fs.value, rows.value
        Bp, Fp, Rp = (B + 15) // 16 * 16, (F + 15) // 16 * 16, (R + 15) // 16 * 16
        p = [t.detach().contiguous() for t in params]
        fb, fb_fc, sb, sb_fc = p[0:8], p[8:10], p[10:18], p[18:20]
        Hf, Hs, Is = fb[1].shape[1], sb[1].shape[1], 2 * nb + 2
        st = _lib.stream_ptr(dev)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        mag = noisy_mag.detach().reshape(B, F, T)
        if not mag.is_contiguous():
            mag = mag.contiguous()
        gws = _lib.workspace(L.fsn_train_glue_workspace_bytes(dp), dev)
        x_tm, mag_tm = new(Tp, Bp, Fp), new(Tp, Bp, Fp)
        _lib.check(L.fsn_train_fb_input(dp, _lib.dev_ptr(mag, "noisy_mag"), _lib.dev_ptr(x_tm), _lib.dev_ptr(mag_tm), Bp, Fp,
                                        gws.data_ptr(), gws.numel(), st))

        def lstm2(x, ldx, w, N, I, H):
            h0, h1 = new(Tp, N, H), new(Tp, N, H)
            nsave = L.fsn_lstm_layer_save_bytes(Tp, N, H)
            s0, s1 = _lib.workspace(nsave, dev), _lib.workspace(nsave, dev)
            ws = _lib.workspace(L.fsn_lstm2_train_workspace_bytes(Tp, N, I, H, ar), dev)
            _lib.check(L.fsn_lstm2_forward_train(_lib.dev_ptr(x), ldx, *[_lib.dev_ptr(t) for t in w], Tp, N, I, H, _lib.dev_ptr(h0),
                                                 _lib.dev_ptr(h1), s0.data_ptr(), s1.data_ptr(), nsave, ws.data_ptr(), ws.numel(), ar, st))
            return h0, h1, s0, s1

        def linear(x, ldx, w, b, rows_, I, O, relu, out=None):
            y = new(rows_, O) if out is None else out
            ws = _lib.workspace(L.fsn_linear_workspace_bytes(rows_, I, O), dev)
            _lib.check(L.fsn_linear_forward(_lib.dev_ptr(x), ldx, _lib.dev_ptr(w), _lib.dev_ptr(b), rows_, I, O, relu, _lib.dev_ptr(y),
                                            ws.data_ptr(), ws.numel(), st))
            return y

        fh0, fh1, fs0, fs1 = lstm2(x_tm, Fp, fb, Bp, F, Hf)
        fb_out = linear(fh1, Hf, fb_fc[0], fb_fc[1], Tp * Bp, Hf, F, 1)          # [Tp Bp, F], ReLU
        sb_in, den = new(Tp, Rp, 32), new(L.fsn_train_den_elems(dp, Rp))  # divisors: per utterance / per unit and frame
        _lib.check(L.fsn_train_sb_input(dp, _lib.dev_ptr(mag_tm), _lib.dev_ptr(fb_out), F, Bp, Fp, _lib.dev_ptr(sb_in), Rp,
                                        _lib.dev_ptr(den), gws.data_ptr(), gws.numel(), st))
        # The sub-band rows are independent sequences (model.py:121-128).  ONE persistent launch per direction holds 2048 of
        # them (config 3's per-rank batch of 16; the kernels of the 16-bit arithmetic exist for such launches only): a larger
        # batch - the shipped TOMLs say 32 and 48 per process (train.toml:52, train_cumulativeLaplaceNorm.toml:52) - runs as
        # equal pieces of whole clusters, each through the same two entries; the pieces' weight gradients add up below
        pieces = None
        if not (L.fsn_lstm2_train_is_persistent(Tp, Rp, Is, Hs) == 1 and Rp % 64 == 0 and Rp <= LSTM2_CHUNK_ROWS):
            # (a launch with a few left-over row tiles beside it - 17 utterances: 2176 rows - is persistent too, but its
            # left-over rows advance step by step: 41 ms against 22 as two pieces)
            pieces = lstm2_train_chunks(Tp, Rp, Is, Hs, pad_to_32=False, pad_small=ar != _lib.ARITH["f32"] and Rp >= 256)
        if pieces is None:
            xs, y2 = [sb_in], new(Tp * Rp, 2)
            y2_out = [y2]
        else:
            xs = list(rows_to_pieces(sb_in, *pieces))                             # n x [Tp, rows, 32]
            y2_pieces = new(pieces[1], Tp, pieces[0], 2)
            y2_out = [y2_pieces[k].view(Tp * pieces[0], 2) for k in range(pieces[1])]
        sb_saved = []
        for xk, yk in zip(xs, y2_out):
            Nk = xk.shape[1]
            h0, h1, s0, s1 = lstm2(xk, 32, sb, Nk, Is, Hs)
            linear(h1, Hs, sb_fc[0], sb_fc[1], Tp * Nk, Hs, 2, 0, out=yk)          # [Tp Nk, 2]
            sb_saved += [xk, h0, h1, s0, s1]
        if pieces is not None:
            y2 = pieces_to_rows(y2_pieces, Rp)                                     # [Tp, Rp, 2]
        mask = new(B, 2, Fs, T)
        _lib.check(L.fsn_train_mask_out(dp, _lib.dev_ptr(y2), Rp, _lib.dev_ptr(mask), st))
        ctx.save_for_backward(x_tm, fh0, fh1, fs0, fs1, fb_out, sb_in, den, gws, *sb_saved, *p)
        ctx.meta = (dims, ar, Tp, Bp, Fp, Rp, Hf, Hs, Is, F, pieces, len(xs))
        return mask

    @staticmethod
    def backward(ctx, d_mask):
        L = _lib.lib()
        dims, ar, Tp, Bp, Fp, Rp, Hf, Hs, Is, F, pieces, n_sb = ctx.meta
        x_tm, fh0, fh1, fs0, fs1, fb_out, sb_in, den, gws = ctx.saved_tensors[:9]
        sb_saved = ctx.saved_tensors[9:9 + 5 * n_sb]
        p = ctx.saved_tensors[9 + 5 * n_sb:]
        fb, fb_fc, sb, sb_fc = p[0:8], p[8:10], p[10:18], p[18:20]
        dp = ctypes.byref(dims)
        dev = d_mask.device
        st = _lib.stream_ptr(dev)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        dm = d_mask if d_mask.is_contiguous() else d_mask.contiguous()

        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if OVERLAP_WEIGHT_PRODUCTS else None
        third = _side_stream(dev, 1) if OVERLAP_WEIGHT_PRODUCTS else None
        keep = []  # what the side stream still reads: alive until the join below

        def linear_bwd(dy, lddy, x, ldx, w, rows_, I, O, beside=None):
            dx, dw, db = new(rows_, ldx), torch.empty_like(w), new(O)
            ws = _lib.workspace(L.fsn_linear_workspace_bytes(rows_, I, O), dev)
            head = (_lib.dev_ptr(dy), lddy, _lib.dev_ptr(x), ldx, _lib.dev_ptr(w), rows_, I, O)
            if beside is not None:  # the input gradient here (the chain waits for it), the parameter gradients beside it
                _lib.check(L.fsn_linear_backward(*head, _lib.dev_ptr(dx), ldx, None, None, ws.data_ptr(), ws.numel(), st))
                ws2 = _lib.workspace(L.fsn_linear_workspace_bytes(rows_, I, O), dev)
                beside.wait_stream(main)
                with torch.cuda.stream(beside):
                    _lib.check(L.fsn_linear_backward(*head, None, ldx, _lib.dev_ptr(dw), _lib.dev_ptr(db), ws2.data_ptr(), ws2.numel(),
                                                     _lib.stream_ptr(dev)))
                keep.extend((ws2, dy))
                return dx, dw, db
            _lib.check(L.fsn_linear_backward(*head, _lib.dev_ptr(dx), ldx, _lib.dev_ptr(dw), _lib.dev_ptr(db), ws.data_ptr(), ws.numel(), st))
            return dx, dw, db

        def lstm2_bwd(dh, x, ldx, w, h0, h1, s0, s1, N, I, H, need_dx, beside=False, dx_out=None):
            dx = (new(Tp, N, ldx) if dx_out is None else dx_out) if need_dx else None
            dw = [torch.empty_like(w[k]) for k in (0, 1, 4, 5)]
            db0, db1 = new(4 * H), new(4 * H)
            ws = _lib.workspace(L.fsn_lstm2_bwd_workspace_bytes(Tp, N, I, H, ar), dev)
            args = (_lib.dev_ptr(dh), _lib.dev_ptr(x), ldx, _lib.dev_ptr(w[0]), _lib.dev_ptr(w[1]), _lib.dev_ptr(w[4]), _lib.dev_ptr(w[5]),
                    Tp, N, I, H, _lib.dev_ptr(h0), _lib.dev_ptr(h1), s0.data_ptr(), s1.data_ptr(), _lib.dev_ptr(dx, allow_none=True), ldx,
                    _lib.dev_ptr(dw[0]), _lib.dev_ptr(dw[1]), _lib.dev_ptr(db0), _lib.dev_ptr(dw[2]), _lib.dev_ptr(dw[3]), _lib.dev_ptr(db1),
                    ws.data_ptr(), ws.numel(), ar)
            if beside and side is not None:
                # Only dx is waited for (the full-band model's backward hangs on it): back-propagation through time and dx
                # here, the weight- and bias-gradient products on a second stream BESIDE dx and what follows - the
                # full-band chain is a latency-bound launch on 96 CUs that leaves the chip to them (fsn_lstm2_backward_phase)
                db0b, db1b = torch.empty_like(db0), torch.empty_like(db1)  # (allocated on the caller's stream, like all outputs)
                side.wait_stream(main)
                with torch.cuda.stream(side):  # what the products need besides the gate gradients: while part 1 runs
                    _lib.check(L.fsn_lstm2_backward_phase(*args, 8, _lib.stream_ptr(dev)))
                _lib.check(L.fsn_lstm2_backward_phase(*args, 1, st))
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    _lib.check(L.fsn_lstm2_backward_phase(*args, 2 + 16, _lib.stream_ptr(dev)))
                    _dup(db0, db0b), _dup(db1, db1b)
                _lib.check(L.fsn_lstm2_backward_phase(*args, 4, st))
                keep.extend((ws, dh, dx))
                return dx, [dw[0], dw[1], db0, db0b, dw[2], dw[3], db1, db1b]
            _lib.check(L.fsn_lstm2_backward(*args, st))
            return dx, [dw[0], dw[1], db0, _dup(db0), dw[2], dw[3], db1, _dup(db1)]

        dy2 = new(Tp * Rp, 16)
        _lib.check(L.fsn_train_mask_grad(dp, _lib.dev_ptr(dm, "d_mask"), _lib.dev_ptr(dy2), Rp, 16, st))
        dy_parts = [dy2] if pieces is None else list(rows_to_pieces(dy2.view(Tp, Rp, 16), *pieces))
        dx_pieces = None if pieces is None else new(pieces[1], Tp, pieces[0], 32)
        dx_sb, g_parts, fc_parts = None, [], []
        for k in range(n_sb):
            xk, sh0, sh1, ss0, ss1 = sb_saved[5 * k:5 * k + 5]
            Nk = xk.shape[1]
            dsh1, d_sfw_k, d_sfb_k = linear_bwd(dy_parts[k].view(Tp * Nk, 16), 16, sh1, Hs, sb_fc[0], Tp * Nk, Hs, 2)  # (its parameter gradients beside the BPTT launch: measured, slows that launch by more)
            dx_sb, g_k = lstm2_bwd(dsh1, xk, 32, sb, sh0, sh1, ss0, ss1, Nk, Is, Hs, True, beside=True,
                                   dx_out=None if pieces is None else dx_pieces[k])
            g_parts.append(g_k), fc_parts.append((d_sfw_k, d_sfb_k))
        if pieces is not None:
            dx_sb = pieces_to_rows(dx_pieces, Rp)
        d_fb = new(Tp * Bp, Fp)
        _lib.check(L.fsn_train_sb_input_backward(dp, _lib.dev_ptr(dx_sb), _lib.dev_ptr(sb_in), Rp, _lib.dev_ptr(den), _lib.dev_ptr(fb_out),
                                                 F, Bp, _lib.dev_ptr(d_fb), Fp, gws.data_ptr(), gws.numel(), st))
        dfh1, d_ffw, d_ffb = linear_bwd(d_fb, Fp, fh1, Hf, fb_fc[0], Tp * Bp, Hf, F, beside=third)  # the chain below only needs dfh1
        _, g_fb = lstm2_bwd(dfh1, x_tm, Fp, fb, fh0, fh1, fs0, fs1, Bp, F, Hf, False)
        if side is not None:
            main.wait_stream(side)  # the sub-band weight gradients: everything after this call sees them
            main.wait_stream(third)
            keep.clear()
        g_sb, (d_sfw, d_sfb) = g_parts[0], fc_parts[0]
        for k in range(1, n_sb):  # the pieces' parameter gradients add up (fixed order)
            g_sb = [a + b for a, b in zip(g_sb, g_parts[k])]
            d_sfw, d_sfb = d_sfw + fc_parts[k][0], d_sfb + fc_parts[k][1]
        return (None, None, None, None, None, None, *g_fb, d_ffw, d_ffb, *g_sb, d_sfw, d_sfb)


def fused_train_supported(model, noisy_mag):
    """The configurations FullSubNetTrainFunction is built for: the shipped FullSubNet TOMLs (LSTM, offline or cumulative
    Laplace norm - fullsubnet/train.toml:82, train_cumulativeLaplaceNorm.toml:82 -, no full-band neighbours, ReLU / linear
    output layers), a ROCm input that does not itself need a gradient."""
    return (getattr(model, "_fused", False) and model.norm_type in _lib.NORM_TYPES and noisy_mag.is_cuda
            and not noisy_mag.requires_grad and 2 * model.sb_num_neighbors + 2 <= 32
            and model.fb_model.sequence_model.num_layers == 2 and model.sb_model.sequence_model.num_layers == 2
            and getattr(model, "fused_training_graph", True))


def _fused_params(model):
    out = []
    for blk in (model.fb_model, model.sb_model):
        lstm = blk.sequence_model
        out += [getattr(lstm, f"{n}_l{k}") for k in (0, 1) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        out += [blk.fc_output_layer.weight, blk.fc_output_layer.bias]
    return out


def forward_train(model, noisy_mag):
    """fullsubnet/model.py:72-136 under autograd (drop_band included), LSTMs on the HIP kernels.
    noisy_mag [B, 1, F, T] -> [B, 2, F // g, T]."""
    arith = train_arith_of(model)
    # drop_band's own check (feature.py:317-319), reached for every batch of more than one utterance (model.py:114): the
    # reference's forward fails for 1 < B <= num_groups, and so does every graph here
    assert noisy_mag.shape[0] == 1 or noisy_mag.shape[0] > model.num_groups_in_drop_band, (
        f"Batch size = {noisy_mag.shape[0]}, num_groups = {model.num_groups_in_drop_band}. "
        "The batch size should larger than the num_groups.")
    if fused_train_supported(model, noisy_mag):  # the whole model as one autograd node on the library's kernels
        return FullSubNetTrainFunction.apply(noisy_mag, model.look_ahead, model.sb_num_neighbors, model.num_groups_in_drop_band,
                                             arith, model.norm_type, *_fused_params(model))
    x = functional.pad(noisy_mag, [0, model.look_ahead])
    B, C, F, Tp = x.shape
    fb_in = _norm(x, model.norm_type).reshape(B, F, Tp)
    h = lstm_stack(fb_in.permute(2, 0, 1), model.fb_model.sequence_model, arith)  # [Tp, B, Hf]
    fc = model.fb_model.fc_output_layer
    fb_out = LinearFunction.apply(h, fc.weight, fc.bias, True)  # ReLU(h W^T + b): [Tp, B, F]
    fb_out = fb_out.permute(1, 2, 0).reshape(B, 1, F, Tp)
    n = model.sb_num_neighbors
    if model.norm_type == "offline_laplace_norm":
        sb_in = SubbandInputOffline.apply(x, fb_out, n, model.num_groups_in_drop_band)
        Fs = sb_in.shape[0] // B
    else:
        sb_in = torch.cat([_freq_unfold(x, n).reshape(B, F, 2 * n + 1, Tp),
                           _freq_unfold(fb_out, 0).reshape(B, F, 1, Tp)], dim=2)
        sb_in = _norm(sb_in, model.norm_type)
        Fs = F
        if B > 1:
            sb_in = drop_band(sb_in.permute(0, 2, 1, 3), num_groups=model.num_groups_in_drop_band)
            Fs = sb_in.shape[2]
            sb_in = sb_in.permute(0, 2, 1, 3)
        sb_in = sb_in.reshape(B * Fs, 2 * n + 2, Tp)
    h = lstm_stack(sb_in.permute(2, 0, 1), model.sb_model.sequence_model, arith)  # [Tp, B Fs, Hs]
    fc = model.sb_model.fc_output_layer
    mask = LinearFunction.apply(h, fc.weight, fc.bias, False)  # [Tp, B Fs, 2]
    mask = mask.permute(1, 2, 0).reshape(B, Fs, 2, Tp).permute(0, 2, 1, 3).contiguous()
    return mask[:, :, :, model.look_ahead:]


class MseLossFunction(torch.autograd.Function):
    """torch.nn.MSELoss() (audio_zen/loss.py:4) through fsn_mse_loss: the loss and d loss / d input in one pass."""

    @staticmethod
    def forward(ctx, input, target):
        L = _lib.lib()
        if input.shape != target.shape:
            raise _lib.FsnError(f"mse_loss: shapes differ {tuple(input.shape)} vs {tuple(target.shape)}")
        x, y = input.contiguous(), target.contiguous()
        n = x.numel()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ws = _lib.workspace(L.fsn_mse_loss_workspace_bytes(n), x.device)
        _lib.check(L.fsn_mse_loss(_lib.dev_ptr(x, "input"), _lib.dev_ptr(y, "target"), n, _lib.dev_ptr(loss),
                                  _lib.dev_ptr(grad, allow_none=True), ws.data_ptr(), ws.numel(),
                                  _lib.stream_ptr(x.device)))
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (grad,) = ctx.saved_tensors
        if dloss.is_cuda and dloss.dtype == torch.float32 and dloss.numel() == 1:  # grad * dloss (a device scalar) on the library
            out = torch.empty_like(grad)
            _lib.check(_lib.lib().fsn_scale_by_scalar(_lib.dev_ptr(grad), _lib.dev_ptr(dloss.reshape(1).contiguous()), _lib.dev_ptr(out),
                                                      grad.numel(), _lib.stream_ptr(grad.device)))
            return out, None
        return grad * dloss, None


def mse_loss(input, target):
    return MseLossFunction.apply(input, target)


def train_step(model, optimizer, noisy, clean, n_fft=512, hop_length=256, win_length=512, clip_grad_norm_value=10.0,
               loss_function=None, scaler=None):
    """One iteration of Trainer._train_epoch (fullsubnet/trainer.py:41-71).  Returns the (unscaled) loss tensor
    (call .item() to synchronise like the reference does).

    fp32 by default (use_amp = false).  With `scaler` (a torch.amp.GradScaler, the reference's own object,
    trainer.py:63-69) and `model.train_arithmetic` in ("f16", "bf16") the step is the reference's autocast step: the
    transforms and the cIRM target stay fp32 (outside the autocast context there too, trainer.py:46-54), the LSTM
    products take 16-bit operands, the loss is scaled before backward, and scaler.step / scaler.update skip the
    update and back the scale off when a gradient is not finite."""
    optimizer.zero_grad()
    noisy_mag, _, noisy_real, noisy_imag = stft(noisy, n_fft, hop_length, win_length, return_phase=False)
    _, _, clean_real, clean_imag = stft(clean, n_fft, hop_length, win_length, return_phase=False)
    inner = model.module if hasattr(model, "module") else model
    # fullsubnet/trainer.py:53 drops bands of the target like the model does of its input; the sibling trainers
    # (fast_fullsubnet/trainer.py:33-76, fullband_baseline/trainer.py) have no band dropping: their models carry no
    # `num_groups_in_drop_band` and the target stays whole
    groups = getattr(inner, "num_groups_in_drop_band", None)
    x_in = noisy_mag.unsqueeze(1)
    if loss_function is None and groups is not None and inner.training and fused_train_supported(inner, x_in):
        # the target straight into the prediction's layout [B, 2, Fs, T] (mask.py:7-44 + drop_band, one kernel): the mean
        # squared error does not care about the element order as long as both sides share it
        B, F, T = noisy_mag.shape
        assert B > groups, (  # drop_band's own check (feature.py:322-323): the reference's step fails the same way
            f"Batch size = {B}, num_groups = {groups}. The batch size should larger than the num_groups.")
        dims = _lib.TrainDims(B, F, T, inner.look_ahead, inner.sb_num_neighbors, groups, _lib.NORM_TYPES[inner.norm_type])
        crm = model(x_in)
        cirm = torch.empty_like(crm)
        _lib.check(_lib.lib().fsn_train_cirm_target(ctypes.byref(dims), *[_lib.dev_ptr(t.contiguous()) for t in
                                                                       (noisy_real, noisy_imag, clean_real, clean_imag)],
                                                    _lib.dev_ptr(cirm), _lib.stream_ptr(cirm.device)))
        loss = mse_loss(crm, cirm)
    else:
        loss_function = loss_function or (lambda target, pred: mse_loss(pred, target))
        cirm = build_complex_ideal_ratio_mask(noisy_real, noisy_imag, clean_real, clean_imag)  # [B, F, T, 2]
        if groups is not None:
            cirm = drop_band(cirm.permute(0, 3, 1, 2), groups).permute(0, 2, 3, 1)
        crm = model(x_in).permute(0, 2, 3, 1)
        loss = loss_function(cirm, crm)
    if scaler is not None and scaler.is_enabled():
        scaler.scale(loss).backward()
        if isinstance(optimizer, ClipAdam):
            # unscale + clip + Adam fused: GradScaler hands the scale to the optimizer (`_step_supports_amp_scaling`),
            # which divides by it before the norm, and skips the update itself when the norm is not finite
            for group in optimizer.param_groups:
                group["clip_grad_norm_value"] = clip_grad_norm_value
            scaler.step(optimizer)
        else:  # the reference's sequence, trainer.py:65-69
            scaler.unscale_(optimizer)
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad_norm_value)
            scaler.step(optimizer)
        scaler.update()
        return loss.detach()
    torch.autograd.backward(loss, grad_tensors=_one(loss.device))  # loss.backward() without its ones_like fill kernel
    if isinstance(optimizer, ClipAdam):  # clip + Adam fused (two launches)
        for group in optimizer.param_groups:
            group["clip_grad_norm_value"] = clip_grad_norm_value
        optimizer.step()
    else:
        # a stock optimizer (the reference's torch.optim.Adam, train.py:55-59) has no skip of its own: a poisoned step - a
        # persistent kernel that ran out of time under the trainer's "defer" policy turns every gradient into NaN - must not
        # reach the parameters and moments (one host read of the norm; the reference syncs on loss.item() every step anyway)
        total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad_norm_value)
        if bool(torch.isfinite(total_norm)):
            optimizer.step()
    return loss
