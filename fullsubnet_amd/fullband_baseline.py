"""Full-band baseline model of the DNS-INTERSPEECH-2020 recipes (BASELINE config 1) on libfsn_hip.so.

Mirror of ``recipes/dns_interspeech_2020/fullband_baseline/model.py:8-68`` - constructor keywords,
``state_dict()`` keys (``fullband_model.sequence_model.*``, ``fullband_model.fc_output_layer.*``) and the
``[B, 1, F, T] -> [B, 2, F, T]`` contract are the reference's; the three LSTM layers and the 2F-wide output
layer run on the HIP LSTM / GEMM kernels through :class:`fullsubnet_amd.sequence_model.SequenceModel`.
"""
from .base_model import BaseModel, look_ahead_pad
from .sequence_model import SequenceModel

_NUM_LAYERS = 3  # fixed by the reference (model.py:33)


class Model(BaseModel):
    def __init__(self, num_freqs, hidden_size, sequence_model, output_activate_function, look_ahead,
                 norm_type="offline_laplace_norm", weight_init=True):
        super().__init__()
        self.look_ahead = look_ahead
        self.norm = self.norm_wrapper(norm_type)
        # one block: F magnitudes in, real and imaginary mask of every bin out
        self.fullband_model = SequenceModel(num_freqs, 2 * num_freqs, hidden_size, _NUM_LAYERS, False,
                                            sequence_model, output_activate_function)
        if weight_init:
            self.apply(self.weight_init)

    def forward(self, noisy_mag):
        if noisy_mag.dim() != 4 or noisy_mag.shape[1] != 1:
            raise AssertionError(f"{self.__class__.__name__} takes the mag feature as inputs ([B, 1, F, T]).")
        frames_in = look_ahead_pad(noisy_mag, self.look_ahead)  # the model may peek `look_ahead` frames ahead
        n_batch, _, n_bins, n_frames = frames_in.shape
        block_in = self.norm(frames_in).reshape(n_batch, n_bins, n_frames)
        mask = self.fullband_model(block_in).reshape(n_batch, 2, n_bins, n_frames)
        return mask[..., self.look_ahead:]  # frame t was produced at step t + look_ahead
