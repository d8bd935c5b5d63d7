"""Keras-2.2.4-shaped facade over the MI355X engine: `from jpeg_detection_resnet_ssd_amd.keras.layers import
Conv2D, ...` mirrors the `keras.*` imports of the reference's builders and training scripts."""
from . import (backend, callbacks, initializers, layers, losses, metrics, models, optimizers, regularizers,  # noqa: F401
               utils)
