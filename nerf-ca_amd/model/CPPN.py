"""Static attenuation field -- drop-in for the reference's ``model/CPPN.py``.

Same constructor dictionary, attributes, methods and state-dict keys (model/CPPN.py:6-180);
``forward`` runs the fused HIP point kernel instead of a chain of ``nn.Linear`` calls.
"""
import torch

from ._field import FieldBase


class CPPN(FieldBase):
    def __init__(self, model_definition: dict) -> None:
        super().__init__()
        self._setup_common(model_definition)
        self.input_features = self._enc_features
        self._build_layers(self.input_features)
        self._bind(time_dim=0, phases=0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """f32[..., 3] -> f32[..., 1]  (model/CPPN.py:88-110)."""
        return self._points(x)

    def pos_enc(self, values, pos_enc_basis, type):
        """model/CPPN.py:112-135 (``type`` is unused there as well)."""
        return self._encode(values, pos_enc_basis)

    def windowed_pos_enc(self, pos_enc_basis, type):
        return self._window_vector(pos_enc_basis)
