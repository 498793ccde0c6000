"""Drop-in ``Model`` for recipes/dns_interspeech_2020/fullband_baseline/model.py:8-68 (BASELINE
config 1): look-ahead pad -> norm -> 3-layer LSTM + Linear(2F) on the HIP LSTM / GEMM kernels."""
from .base_model import BaseModel, look_ahead_pad
from .sequence_model import SequenceModel


class Model(BaseModel):
    def __init__(self, num_freqs, hidden_size, sequence_model, output_activate_function, look_ahead,
                 norm_type="offline_laplace_norm", weight_init=True):
        super().__init__()
        self.fullband_model = SequenceModel(input_size=num_freqs, output_size=num_freqs * 2, hidden_size=hidden_size,
                                            num_layers=3, bidirectional=False, sequence_model=sequence_model,
                                            output_activate_function=output_activate_function)
        self.look_ahead = look_ahead
        self.norm = self.norm_wrapper(norm_type)
        if weight_init:
            self.apply(self.weight_init)

    def forward(self, noisy_mag):
        """noisy_mag [B, 1, F, T] -> [B, 2, F, T]."""
        assert noisy_mag.dim() == 4
        noisy_mag = look_ahead_pad(noisy_mag, self.look_ahead)
        batch_size, num_channels, num_freqs, num_frames = noisy_mag.size()
        assert num_channels == 1, f"{self.__class__.__name__} takes the mag feature as inputs."
        x = self.norm(noisy_mag).reshape(batch_size, num_channels * num_freqs, num_frames)
        output = self.fullband_model(x).reshape(batch_size, 2, num_freqs, num_frames)
        return output[:, :, :, self.look_ahead:]
