"""MI355X-native (gfx950) implementation of the ResNet50-DCT + SSD300 training hot path of
Shulk97/JPEG_detection_Resnet_SSD, behind the reference's Keras-style surface."""
__version__ = "0.1.0"
