"""classification_part/config/resnetRGB: the same TrainingConfiguration fixed to the RGB ResNet50
(`TrainingConfiguration(load_pretrained_weights=...)`, classification_part/training.py:108-109)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "resnet"))
import importlib.util

_spec = importlib.util.spec_from_file_location("_resnet_config", os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                             "..", "resnet", "config_file.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)


class TrainingConfiguration(_mod.TrainingConfiguration):
    def __init__(self, load_pretrained_weights=True):
        super(TrainingConfiguration, self).__init__(deconv=False, archi="resnet_rgb",
                                                    load_pretrained_weights=load_pretrained_weights)
