"""Defaults of the reference (initial_params.py:6-42), restricted to the hot-path models."""
import os

_PKG = os.path.dirname(os.path.realpath(__file__))


class InitialParameters:
    default_data_path = os.environ.get("RGB_DATA_ROOT", os.path.join(os.path.dirname(_PKG), "data"))
    default_pics_path = os.path.join(os.path.dirname(_PKG), "pics")

    model_names = ["MLP", "GCN", "GraphSAGE", "GAT", "APPNPStack", "GraphSAGE2", "PTA", "DAGNN", "SGC", "GIN"]
    # reference initial_params.py:24-35
    default_init_params = [
        {"num_layers": 3, "hidden_unit": 64, "dropout_rate": 0.5},
        {"num_layers": 2, "hidden_unit": 64, "dropout_rate": 0.5},
        {"num_layers": 2, "hidden_unit": 64, "dropout_rate": 0.5},
        {"num_layers": 2, "hidden_unit": 8, "dropout_rate": 0.5, "heads": 8},
        {"hidden_unit": 64, "dropout_rate": 0.5, "alpha": 0.1, "K": 10},
        {"num_layers": 2, "hidden_unit": 64, "dropout_rate": 0.5},
        {"nhid": 64, "dropout": 0, "epsilon": 100, "mode": 2, "K": 10, "alpha": 0.1},
        {"hidden_dim": 64, "K": 10, "dropout_rate": 0.5},
        {"K": 2},
        {"num_layers": 2, "hidden_unit": 64, "dropout_rate": 0.5},
    ]

    # reference initial_params.py:42
    default_cs_param = {"num_correction_layers": 50, "correction_alpha": 0.8, "num_smoothing_layers": 50,
                        "smoothing_alpha": 0.8, "autoscale": True}

    @classmethod
    def defaults_for(cls, model_name):
        for name, params in zip(cls.model_names, cls.default_init_params):
            if name.lower() == model_name.lower():
                return dict(params)
        raise KeyError(model_name)
