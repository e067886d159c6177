"""rgb_experiment_amd — MI355X-native message-passing hot path behind the rgb-experiment plugin
surface (reference rgb_experiment/__init__.py:1-4)."""
from .data import Data
from .initial_params import InitialParameters
from .itexperiments import compare_pred_label, experiment, label_propagation, normalized_adjacency, test
from .rd2pd import RD2PD

__version__ = "0.1.0"
__all__ = ["experiment", "test", "compare_pred_label", "label_propagation", "normalized_adjacency",
           "InitialParameters", "RD2PD", "Data"]
