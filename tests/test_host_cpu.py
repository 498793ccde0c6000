"""CPU-side checks: the C-ABI library loads and exports every symbol include/fsn_hip.h declares,
the ctypes table matches the header, and the host-side mirror keeps the reference's surface.
No GPU compute is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fsn_hip.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fsn_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    syms = header_symbols()
    for must in ("fsn_stft", "fsn_istft", "fsn_fullsubnet_forward", "fsn_fullsubnet_pack", "fsn_enhance",
                 "fsn_decompress_cirm", "fsn_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from fullsubnet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for s in header_symbols():
        assert hasattr(handle, s), f"{s} declared in include/fsn_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == header_symbols(), "ctypes table and header disagree"
    L = _lib.lib()
    assert L.fsn_version() >= 100
    names = _lib.profile_stage_names()
    assert "sb_rec_l1" in names and len(names) == L.fsn_profile_num_stages()


def test_argument_validation_without_a_gpu():
    from fullsubnet_amd import _lib
    L = _lib.lib()
    bad = _lib.Cfg(257, 2, 15, 512, 100, 0)  # unsupported sub-band hidden size
    assert L.fsn_fullsubnet_packed_bytes(ctypes.byref(bad)) == 0
    assert b"sb_hidden" in L.fsn_last_error()
    ok = _lib.Cfg(257, 2, 15, 512, 384, 0)
    nbytes = L.fsn_fullsubnet_packed_bytes(ctypes.byref(ok))
    assert nbytes >= 5_637_635 * 4  # at least the parameter count of the model
    ws2 = L.fsn_enhance_workspace_bytes(ctypes.byref(ok), 2, 16000, 512, 256)
    ws64 = L.fsn_enhance_workspace_bytes(ctypes.byref(ok), 64, 48000, 512, 256)
    assert 0 < ws2 < ws64 < 64 << 30
    assert L.fsn_enhance_workspace_bytes(ctypes.byref(ok), 2, 16000, 400, 100) == 0  # unsupported FFT
    with pytest.raises(_lib.FsnError):
        _lib.dev_ptr(torch.zeros(4), "x")  # CPU tensor is rejected, no fallback
    # row-range form: a slice only pays for the utterances it touches; an aligned full range is the plain forward
    full = L.fsn_fullsubnet_workspace_bytes(ctypes.byref(ok), 8, 100)
    assert L.fsn_fullsubnet_rows_workspace_bytes(ctypes.byref(ok), 8, 100, 0, 8 * 257) == full
    one = L.fsn_fullsubnet_workspace_bytes(ctypes.byref(ok), 1, 100)
    assert L.fsn_fullsubnet_rows_workspace_bytes(ctypes.byref(ok), 8, 100, 3 * 257, 4 * 257) == one
    # (not "< full": eight utterances run on the group kernel, whose workspace holds no gate buffer for its rows)
    two = L.fsn_fullsubnet_workspace_bytes(ctypes.byref(ok), 2, 100)
    assert one < L.fsn_fullsubnet_rows_workspace_bytes(ctypes.byref(ok), 8, 100, 3 * 257 - 1, 4 * 257) <= two
    for lo, hi in ((5, 5), (-1, 10), (0, 8 * 257 + 1)):
        assert L.fsn_fullsubnet_rows_workspace_bytes(ctypes.byref(ok), 8, 100, lo, hi) == 0
        assert b"row range" in L.fsn_last_error()


def test_two_layer_training_entries_size_queries_without_a_gpu():
    """fsn_lstm2_forward_train / fsn_lstm2_backward / fsn_lstm2_forward: the workspace queries answer without a device
    (whatever kernel the shape will run on there) and cover at least what the layer-by-layer path needs."""
    from fullsubnet_amd import _lib
    L = _lib.lib()
    for T, N, I, H in [(193, 2064, 32, 384), (193, 16, 257, 512), (10, 48, 64, 512), (5, 4128, 32, 384)]:
        fwd = L.fsn_lstm2_train_workspace_bytes(T, N, I, H, 0)
        bwd = L.fsn_lstm2_bwd_workspace_bytes(T, N, I, H, 0)
        assert fwd > 0 and bwd >= T * N * 4 * H * 4  # one layer's gate gradients at least (layer by layer)
        assert L.fsn_lstm2_fwd_workspace_bytes(T, N, I, H, H) >= L.fsn_lstm_layer_fwd_workspace_bytes(T, N, I, H) // 2
    assert L.fsn_lstm2_forward_is_persistent(100, 2048, 16, 16, 384, 384) in (0, 1)
    assert L.fsn_lstm2_forward_is_persistent(100, 2048, 64, 64, 384, 384) == 0   # more than 32 input columns
    assert L.fsn_lstm2_forward_is_persistent(100, 2048, 16, 16, 384, 512) == 0   # unequal widths
    assert L.fsn_lstm2_forward_is_persistent(100, 2048, 16, 48, 384, 384) == 0   # x rows wider than two K chunks


def test_gru_many_row_plan_and_workspace_without_a_gpu():
    """fsn_gru_layer_is_persistent / fsn_gru_layer_fwd_workspace_bytes (ABI 117) answer without a device (256 CUs assumed): the
    GRU takes the persistent many-row kernels for H = 384, at least 9/8 x CUs row tiles and a narrow row-major input or the layer
    above an equally wide one; the workspace then holds the four-gate matrices and the left-over rows' compact copies on top of
    the step form's buffers, and never less than the step form needs."""
    from fullsubnet_amd import _lib
    L = _lib.lib()
    T, H = 190, 384
    per = lambda N, I, ldx, Hh=H: L.fsn_gru_layer_is_persistent(T, N, I, ldx, Hh)
    assert per(16448, 32, 32) == 1 and per(16448, 384, 384) == 1 and per(16448, 12, 16) == 1 and per(16448, 20, 48) == 1
    assert per(16448, 40, 48) == 0        # more than 32 input columns and not the stacked form
    assert per(16448, 384, 400) == 0      # the stacked form reads the hidden sequence as it lies (ldx = H)
    assert per(16448, 32, 32, 320) == 0 and per(16448, 32, 32, 512) == 0
    assert per(16 * 287, 32, 32) == 0 and per(16 * 288, 32, 32) == 1   # from 9/8 x 256 row tiles on
    assert per(16448 + 8, 32, 32) == 0    # rows in whole 16-row tiles
    step = lambda N, I: 4 * (3 * H * ((I + 15) // 16 * 16) + 3 * H * H + 3 * H + T * N * 3 * H)
    for N, I in [(16448, 32), (16448, 384), (4608, 12), (272, 32), (16, 257)]:
        ws = L.fsn_gru_layer_fwd_workspace_bytes(T, N, I, H)
        assert ws >= step(N, I), (N, I, ws)
        if per(N, I, (I + 15) // 16 * 16):
            assert ws >= step(N, I) + 4 * 2 * (4 * H * ((I + 15) // 16 * 16) + 4 * H * H), (N, I, ws)
    assert L.fsn_gru_layer_fwd_workspace_bytes(0, 16, 32, H) == 0


def test_fast_fullsubnet_glue_queries_without_a_gpu():
    """fsn_fast_low_rate_frames = the length real_time_downsampling produces (fast_fullsubnet/model.py:108-129: frame 0,
    then blocks of `shrink` frames, a shorter last block kept) for every (T, shrink); the workspace query covers the two
    down-sampled sources + one mean per utterance; bad sizes answer 0."""
    from fullsubnet_amd import _lib
    from fullsubnet_amd.fast_fullsubnet import Model
    L = _lib.lib()
    m = Model.__new__(Model)  # the glue methods only read shrink_size
    for shrink in (1, 2, 3, 4, 7):
        m.shrink_size = shrink
        for T in range(2, 40):
            want = m.real_time_downsampling(torch.zeros(1, 1, 1, T)).shape[-1]
            assert L.fsn_fast_low_rate_frames(T, shrink) == want, (T, shrink)
            up = m.real_time_upsampling(torch.zeros(1, 1, 1, want), target_len=T)
            assert up.shape[-1] == T  # every frame t finds its low-rate frame t // shrink
            assert (T - 1) // shrink < want
    Ts = L.fsn_fast_low_rate_frames(190, 2)
    assert L.fsn_fast_glue_workspace_bytes(190, 256, 64, 2) >= 2 * Ts * 256 * 64 * 4 + 256 * 4
    for bad in ((1, 4, 64, 2), (10, 0, 64, 2), (10, 4, 0, 2), (10, 4, 64, 0)):
        assert L.fsn_fast_glue_workspace_bytes(*bad) == 0
    assert L.fsn_fast_low_rate_frames(1, 2) == 0 and L.fsn_fast_low_rate_frames(10, 0) == 0


def test_model_surface_matches_reference_state_dict():
    from fullsubnet_amd import Model, _lib
    from oracle.fullsubnet_oracle import make_params
    m = Model(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
              fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
              sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=2,
              weight_init=True)
    ref = make_params(seed=0)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())  # same names, same order as the reference's Model
    for k, v in ref.items():
        assert tuple(sd[k].shape) == v.shape, k
    assert sorted(_lib.STATE_KEYS) == sorted(ref.keys())
    assert sum(p.numel() for p in m.parameters()) == 5_637_635
    m.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()}, strict=True)
    # orthogonal init ran (base_model.py:416-421)
    m2 = Model(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
               fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
               sb_model_hidden_size=384, weight_init=True)
    w = m2.sb_model.sequence_model.weight_hh_l0.detach()
    assert torch.allclose(w.T @ w, torch.eye(384), atol=1e-4)
    with pytest.raises(Exception):
        m(torch.zeros(1, 1, 257, 10))  # CPU tensor: there is no fallback path


def test_drop_band_mirror_matches_golden(golden_dir):
    from fullsubnet_amd import drop_band
    z = np.load(os.path.join(golden_dir, "elementwise.npz"))
    np.testing.assert_array_equal(drop_band(torch.from_numpy(z["x"]), 2).numpy(), z["drop2"])
    np.testing.assert_array_equal(drop_band(torch.from_numpy(z["x"]), 3).numpy(), z["drop3"])


def test_drop_band_after_the_model_equals_drop_band_before(golden_dir):
    """Host logic of Model.forward for B > 1: selecting rows of the full mask afterwards is what the
    reference computes by dropping bands before the sub-band model (quirk Q1)."""
    import ast
    from oracle import fullsubnet_oracle as O
    z = np.load(os.path.join(golden_dir, "fsn_dropband_b4.npz"))
    meta = ast.literal_eval(str(z["meta"]))
    params = O.make_params(seed=meta["seed_w"], gain=meta["gain"], mask_gain=meta["mask_gain"])
    full = O.fullsubnet_forward(z["mag"][:, None], params, num_groups_in_drop_band=1)
    sel = O.drop_band(full, 2)
    assert np.abs(sel - z["crm"]).max() <= 1e-4


def test_subband_input_function_matches_unfold_cat_norm_dropband():
    """fullsubnet_amd.train.SubbandInputOffline (gather + analytic mean, custom backward) against
    the reference's op sequence unfold -> cat -> offline norm -> drop_band under autograd (CPU)."""
    from fullsubnet_amd.train import SubbandInputOffline, _freq_unfold, _norm
    from fullsubnet_amd.acoustics.feature import drop_band
    torch.manual_seed(0)
    for B, F, Tp, n, G in [(4, 33, 6, 5, 2), (1, 17, 5, 3, 2), (6, 20, 4, 4, 3)]:
        x = torch.rand(B, 1, F, Tp, requires_grad=True)
        x2 = x.detach().clone().requires_grad_(True)
        fb = torch.rand(B, 1, F, Tp, requires_grad=True)
        fb2 = fb.detach().clone().requires_grad_(True)
        a = SubbandInputOffline.apply(x, fb, n, G)
        ref = torch.cat([_freq_unfold(x2, n).reshape(B, F, 2 * n + 1, Tp), _freq_unfold(fb2, 0).reshape(B, F, 1, Tp)], 2)
        ref = _norm(ref, "offline_laplace_norm")
        if B > 1:
            ref = drop_band(ref.permute(0, 2, 1, 3), G).permute(0, 2, 1, 3)
        ref = ref.reshape(-1, 2 * n + 2, Tp)
        assert a.shape == ref.shape
        assert torch.allclose(a, ref, rtol=1e-5, atol=1e-6)
        w = torch.randn_like(ref)
        (a * w).sum().backward()
        (ref * w).sum().backward()
        assert torch.allclose(fb.grad, fb2.grad, rtol=1e-4, atol=1e-6), (B, F)
        assert torch.allclose(x.grad, x2.grad, rtol=1e-4, atol=1e-6), (B, F)  # the input gradient as well


def test_leftover_step_kernel_fits_next_to_persistent_kernel(tmp_path):
    """The left-over sub-band tiles run as lstm_step1_kernel launches CONCURRENTLY with the persistent
    lstm_rec_kernel.  That only happens if both fit on a CU together: 12 persistent waves =
    3 per SIMD x their register count, plus one step wave, within the 512-entry register file, and both
    LDS allocations within 160 KB.  A compiler or source change that breaks this costs ~1.4 ms per batch
    silently (the step kernels then queue behind the 32 ms kernel), so the budget is checked on the
    compiled code object's metadata."""
    import re
    import shutil
    import subprocess
    from fullsubnet_amd import build as b
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "lstm.s"
    flags = [f for f in b.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", os.path.join(b.CSRC, "lstm_kernels.hip"), "-o", str(out)],
                   check=True, capture_output=True)
    meta = {}
    for blk in out.read_text().split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                      for k in ("vgpr_count", "group_segment_fixed_size", "vgpr_spill_count")}
    step = next(v for k, v in meta.items() if "lstm_step1_kernel" in k)
    recs = {k: v for k, v in meta.items() if "lstm_rec_kernelILi384ELi4ELi2E" in k}
    assert len(recs) == 2 and step["vgpr_spill_count"] == 0
    gran = lambda n: (n + 7) // 8 * 8  # VGPR allocation granule on gfx950
    for name, rec in recs.items():
        assert rec["vgpr_spill_count"] == 0, name
        assert 3 * gran(rec["vgpr_count"]) + gran(step["vgpr_count"]) <= 512, (name, rec, step)
    # dynamic LDS of the persistent kernel: 16 RT (H + 4) floats for h + the double-buffered layer-0 input tile
    lds_rec = 4 * (64 * 388) + 4 * (2 * 64 * 36)
    assert lds_rec + 3 * step["group_segment_fixed_size"] <= 160 * 1024
    # the last layer with its input projection inside (lstm_rec_x_kernel): capped at 152 registers by attribute;
    # what does not fit is spilled OUTSIDE the time loop (loop-invariant values, reloaded once per step)
    recx = next(v for k, v in meta.items() if "lstm_rec_x_kernelILi384ELi4ELi2E" in k)
    assert 3 * gran(recx["vgpr_count"]) + gran(step["vgpr_count"]) <= 512, (recx, step)
    assert recx["vgpr_spill_count"] <= 8, recx
    lds_x = 4 * (64 * 388 + 2 * 384 + 2 * 24 * 256)  # h + output weights + two ring stages of 24 fragments
    assert lds_x + step["group_segment_fixed_size"] <= 156 * 1024  # one step workgroup per CU, with headroom


def test_built_library_holds_the_same_budget():
    """The same contract on the kernels INSIDE libfsn_hip.so - what actually travels to the GPU box (a stale or
    differently built library once cost 4.5 ms per batch unnoticed: its persistent kernels had 168 registers)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("so_kernel_resources",
                                                  os.path.join(ROOT, "tools", "so_kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.READELF):
        pytest.skip("llvm-readelf not available")
    ks = mod.kernels()
    step = next(v for k, v in ks.items() if "lstm_step1_kernel" in k)
    recs = {k: v for k, v in ks.items() if "lstm_rec_kernelILi384ELi4ELi2E" in k}
    assert len(recs) == 2
    gran = lambda n: (n + 7) // 8 * 8
    for name, rec in recs.items():
        assert rec["vgpr_spill_count"] == 0 and rec["private_segment_fixed_size"] == 0, (name, rec)
        assert 3 * gran(rec["vgpr_count"]) + gran(step["vgpr_count"]) <= 512, (name, rec, step)
    recx = next(v for k, v in ks.items() if "lstm_rec_x_kernelILi384ELi4ELi2E" in k)
    assert 3 * gran(recx["vgpr_count"]) + gran(step["vgpr_count"]) <= 512 and recx["vgpr_spill_count"] <= 8, recx
    # the GRU's left-over tiles beside the same kernels run as a four-gate cell (FSN_REC_GRU = 1 << 20 in their OPT / ABL
    # parameter; gru_step1_kernel): every such instantiation x the step workgroup inside a SIMD's registers and a CU's LDS
    gstep = next(v for k, v in ks.items() if "gru_step1_kernel" in k)
    assert gstep["vgpr_spill_count"] == 0 and gstep["group_segment_fixed_size"] <= 8 * 1024, gstep
    gru = {k: v for k, v in ks.items() if re.search(r"lstm_rec_(in|x)_kernelILi384ELi\dELi2ELi(\d+)E", k)
           and int(re.search(r"lstm_rec_(?:in|x)_kernelILi384ELi\dELi2ELi(\d+)E", k).group(1)) & (1 << 20)}
    assert len(gru) == 9, sorted(gru)  # rec_in: 2 - 4 row tiles x one / two input chunks; rec_x: 2 - 4 row tiles (hseq out)
    for name, rec in gru.items():
        assert 3 * gran(rec["vgpr_count"]) + gran(gstep["vgpr_count"]) <= 512 and rec["vgpr_spill_count"] <= 8, (name, rec, gstep)


def test_no_kernel_rewrites_store_data_inside_the_measured_unsafe_distance():
    """gfx950 store-data hazard (tools/probe_store_hazard.hip, profiles/r06_store_hazard.md): a vector instruction that writes
    the data registers of a 12 / 16-byte-per-lane store 1 - 2 issue slots behind it (1 behind a buffer store with an SGPR
    soffset - the case hipcc's hazard recogniser exempts, and what corrupted lstm2_g16_bwd_kernel's gate gradients in round 5)
    puts the new value into memory for the last lane quads.  No kernel of the shipped library may contain such a pair; the
    stand-alone probe's failing pairs and a build of the BPTT kernel without fsn_hold_store_data must both be caught."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(ROOT, "tools", "check_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        pytest.skip("llvm-objdump not available")
    hits = mod.scan(window=8)
    assert hits, "the scan found no store at all: the disassembly step is broken"
    assert not mod.unsafe(hits), mod.unsafe(hits)[:4]
    # the rule itself, on hand-made hits: (kernel, store, distance, writer)
    mk = lambda d, w: ("k", "buffer_store_dwordx4 v[38:41], v134, s[68:71], s0 offen", d, w)
    assert mod.unsafe([mk(1, "v_pk_add_f32 v[38:39], v[38:39], v[50:51]"), mk(2, "v_mov_b32_dpp v38, v40"), mk(3, "v_mov_b32_e32 v38, v1"),
                       mk(1, "ds_read_b128 v[38:41], v138")]) == [mk(1, "v_pk_add_f32 v[38:39], v[38:39], v[50:51]"), mk(2, "v_mov_b32_dpp v38, v40")]


def test_ring_fills_of_the_persistent_kernel_take_scalar_addresses():
    """Round 5: lstm_rec_x_kernel's ring fills cost 0.65 ms per batch while each LDS-DMA fragment formed its
    address in vector registers inside a loop the compiler could not unroll.  In the shipped library every fill of the time
    loop must be the scalar-base form (`global_load_lds_dwordx4 v, s[..]`); the one loop-form fill left is the prologue's."""
    import importlib.util
    import re
    import subprocess
    import tempfile
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(ROOT, "tools", "check_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        pytest.skip("llvm-objdump not available")
    forms = {}
    with tempfile.TemporaryDirectory() as tmp:
        for k, blob in enumerate(mod.code_objects(mod.SO)):
            path = os.path.join(tmp, f"co{k}.elf")
            with open(path, "wb") as f:
                f.write(blob)
            dis = subprocess.run([mod.OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
            kernel = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    kernel = m.group(1)
                elif kernel and "lstm_rec_x_kernel" in kernel and "global_load_lds_dwordx4" in line:
                    v = forms.setdefault(kernel, [0, 0])
                    v[0 if re.search(r"\boff\b", line) else 1] += 1
    # 2 - 4 row tiles per workgroup x (fused output layer | hidden sequence out), + the GRU as a four-gate cell (hidden sequence out)
    assert len(forms) == 9, sorted(forms)
    for kernel, (vector_form, scalar_form) in forms.items():
        assert vector_form <= 1 and scalar_form >= 4, (kernel, vector_form, scalar_form)


def test_subband_multiplicity_closed_form():
    """offline_den_kernel (elementwise_kernels.hip) weighs bin f by m[f] = number of (unit, row) pairs of
    freq_unfold that read it; the kernel's closed form against the brute-force count over the reflect map."""
    def brute(F, nb):
        m = [0] * F
        for u in range(F):
            for k in range(-nb, nb + 1):
                j = abs(u + k)
                j = 2 * (F - 1) - j if j >= F else j
                m[j] += 1
        return m

    def closed(F, nb):
        out = []
        for f in range(F):
            direct = min(nb, f) - max(-nb, f - (F - 1)) + 1
            low = max(0, min(F - 1, nb - f) + 1) if f >= 1 else 0
            high = max(0, F - max(0, 2 * (F - 1) - f - nb)) if f <= F - 2 else 0
            out.append(direct + low + high)
        return out

    for F, nb in [(257, 15), (257, 0), (64, 5), (481, 15), (33, 15), (20, 7)]:
        assert brute(F, nb) == closed(F, nb), (F, nb)
    m = closed(257, 15)
    assert m[0] == m[256] == 16 and m[1] == 32 and m[100] == 31 and sum(m) == 31 * 257


def test_gate_admission_rule_of_persistent_launches():
    """The rule of fsn_api.hip's PersistLaunch on hypothetical sets, no device needed: launches of different
    streams run side by side only when every workgroup of every kernel is placeable in any dispatch order - the sum of
    chip fractions stays below the smallest CU fill that could refuse a workgroup of any kernel of the set."""
    import ctypes
    from fullsubnet_amd import _lib
    L = _lib.lib()

    def fits(*launches):  # (workgroups, resident per CU) on 256 CUs
        n = len(launches)
        fr = (ctypes.c_double * n)(*[wg / (occ * 256.0) for wg, occ in launches])
        oc = (ctypes.c_int * n)(*[occ for _, occ in launches])
        return L.fsn_debug_persist_set_fits(n, fr, oc)

    chain2, chain4, chain1 = (192, 2), (192, 4), (192, 1)  # H = 384 chain kernel: 2 / 1 / 4 row tiles
    group, group_small = (448, 2), (160, 2)                # group kernel (two resident per CU): 28 and 10 clusters
    assert fits(chain2) == 1 and fits(group) == 1 and fits((512, 2)) == 1  # alone: always (the plan checked the grid)
    assert fits(chain2, chain2) == 1          # 384 half-CU workgroups on 512 half-CU slots: no fragmentation possible
    assert fits(chain2, chain4) == 1
    assert fits(chain4, chain4) == 1
    assert fits(chain2, chain2, chain4) == 0  # 0.94 of the chip in two sizes: a state with every CU > 1/2 full exists
    assert fits(chain2, chain4, chain4) == 0  # 0.75 = the fill that can refuse a half-CU workgroup: refused (conservative)
    assert fits(chain1, chain4) == 0          # a one-per-CU kernel shares with nothing
    assert fits(group, group) == 0            # 1.75 chips' worth
    assert fits((448, 4), (448, 4)) == 1      # the same grids at four per CU: 896 quarter-CU workgroups on 1024 slots
    assert fits(group_small, group_small) == 1
    assert fits(group, chain2) == 0
    # more launches never make a refused set admissible
    assert fits(chain2, chain2, chain4, chain4) == 0


def test_weight_gradient_scratch_bound_covers_every_plan():
    """fsn_launch_gemm_tn splits K by one of two plans (256 x 128 tiles, or 192 x 192 tiles with every split on one XCD),
    depending on shape and K; every caller sizes ONE scratch buffer per shape by the bound of
    fsn_gemm_tn_workspace_bytes.  A plan beyond that bound was a memory overrun in round 3 (the GRU backward's 2H x H and
    H x H products): the bound must cover the plan for every shape and every K."""
    import ctypes
    from fullsubnet_amd import _lib
    L = _lib.lib()
    worst = 0.0
    for M in (64, 192, 257, 384, 512, 768, 1024, 1152, 1536, 2048, 3072):
        for Nc in (2, 16, 32, 33, 64, 128, 192, 257, 384, 416, 512, 768):
            for K in (16, 100, 2000, 8192, 9504, 65536, 402480, 3000000):
                for arith in (_lib.ARITH["f32"], _lib.ARITH["f16"], _lib.ARITH["bf16"]):
                    splits, bound = ctypes.c_int(0), ctypes.c_long(0)
                    assert L.fsn_debug_tn_plan(M, Nc, K, arith, ctypes.byref(splits), ctypes.byref(bound)) == 0
                    assert 1 <= splits.value <= bound.value, (M, Nc, K, arith, splits.value, bound.value)
                    worst = max(worst, splits.value / bound.value)
    assert worst == 1.0  # the bound is attained somewhere: it is not a loose over-estimate


def test_batches_are_covered_exactly_by_the_chunks_they_run_as():
    """fsn_enhance / fsn_fullsubnet_forward run a batch as one or several calls of the model core (whole rounds of the
    persistent kernels plus a remainder split by a cost model, include/fsn_hip.h: fsn_debug_core_chunks).  Whatever
    plans the device at hand offers, the chunks must cover the batch exactly once, largest rounds first, and never be
    more than the contract allows (80 calls for 4096 utterances)."""
    import ctypes
    from fullsubnet_amd import _lib
    L = _lib.lib()
    cfg = _lib.Cfg(257, 2, 15, 512, 384, 0, 0)
    sizes = (ctypes.c_int * 80)()
    for B in list(range(1, 140)) + [192, 255, 256, 257, 1000, 4096]:
        n = L.fsn_debug_core_chunks(ctypes.byref(cfg), B, sizes, 80)
        got = [sizes[i] for i in range(n)]
        assert 1 <= n <= 80 and all(g >= 1 for g in got) and sum(got) == B, (B, got)
        assert got == sorted(got, reverse=True), (B, got)
    assert L.fsn_debug_core_chunks(ctypes.byref(cfg), 0, sizes, 80) == -1
    assert L.fsn_debug_core_chunks(ctypes.byref(cfg), 8, sizes, 4) == -1
