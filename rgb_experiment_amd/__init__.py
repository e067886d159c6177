"""rgb_experiment_amd — MI355X-native message-passing hot path behind the rgb-experiment plugin
surface (reference rgb_experiment/__init__.py:1-4)."""
from .data import Data
from .initial_params import InitialParameters
from .itexperiments import compare_pred_label, experiment, label_propagation, normalized_adjacency, test
from .rd2pd import RD2PD



def visualize_feature(x, y=None, pics_root=None, pic_name="pic1.png", title=None):
    """Name kept for import compatibility (reference rgb_experiment/__init__.py:4, visualize_feature.py:11-15: a PCA
    scatter plot through sklearn + matplotlib). Reporting is outside the hot-path scope of this build."""
    raise NotImplementedError("visualize_feature (PCA plot) is a reporting feature outside the MI355X hot-path scope; "
                              "use the reference's visualize_feature.py on the returned embeddings")


__version__ = "0.1.0"
__all__ = ["experiment", "test", "compare_pred_label", "label_propagation", "normalized_adjacency",
           "InitialParameters", "RD2PD", "Data", "visualize_feature"]
