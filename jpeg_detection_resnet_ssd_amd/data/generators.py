"""Synthetic stand-in for DataGeneratorDCT (localisation_part/data_generator/object_detection_2d_data_generator_dct_j2d.py:71):
same `generate(...)` / `get_dataset_size()` surface and the same emission contract (inputs list + encoded labels,
NHWC numpy arrays, Y / CbCr or Y / Cb / Cr with `deconv=True`), fed by synthetic JPEG-DCT images instead of Pascal VOC
(the dataset, PIL/cv2 augmentation and jpeg2dct are out of scope and absent here)."""
import numpy as np

from . import synthetic_dct as sd


class SyntheticDataGeneratorDCT(object):
    def __init__(self, n_images=256, seed=1234, load_images_into_memory=False, hdf5_dataset_path=None, **kwargs):
        self.n_images = int(n_images)
        self.seed = int(seed)

    def parse_xml(self, *args, **kwargs):
        return None

    def get_dataset_size(self):
        return self.n_images

    def generate(self, batch_size=32, shuffle=True, transformations=(), label_encoder=None,
                 returns=("processed_images", "encoded_labels"), keep_images_without_gt=False, deconv=False,
                 fast=True, **kwargs):
        step = 0
        pool = max(1, self.n_images // batch_size)
        while True:
            s = self.seed + (step % pool)
            x = (sd.fast_dct_batch(batch_size, seed=s, split_chroma=deconv) if fast
                 else sd.dct_batch(batch_size, seed=s, split_chroma=deconv))
            gt = sd.random_ground_truth(batch_size, seed=s)
            y = label_encoder(gt) if label_encoder is not None else gt
            step += 1
            yield x, y
