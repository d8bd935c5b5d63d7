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
        self.image_ids = ["%06d" % i for i in range(self.n_images)]
        self.eval_neutral = None          # no 'difficult' annotations in the synthetic set
        self._labels = None

    @property
    def labels(self):
        """Per-image ground truth (class, xmin, ymin, xmax, ymax) of the evaluation view of the dataset."""
        if self._labels is None:
            self._labels = [sd.random_ground_truth(1, seed=self.seed * 7 + i)[0] for i in range(self.n_images)]
        return self._labels

    def parse_xml(self, *args, **kwargs):
        return None

    def get_dataset_size(self):
        return self.n_images

    def generate(self, batch_size=32, shuffle=True, transformations=(), label_encoder=None,
                 returns=("processed_images", "encoded_labels"), keep_images_without_gt=False, deconv=False,
                 fast=True, **kwargs):
        if "image_ids" in returns or "inverse_transform" in returns or "processed_labels" in returns:
            for item in self._generate_indexed(batch_size, label_encoder, returns, deconv):
                yield item
            return
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

    def _generate_indexed(self, batch_size, label_encoder, returns, deconv):
        """Evaluation view: image i is the same whatever the batch size, ids / labels travel with the batch, output
        order as in the reference's generator (processed_images, encoded_labels, processed_labels, image_ids,
        inverse_transform; object_detection_2d_data_generator_dct_j2d.py:1196-1206).  Wraps around the dataset."""
        current = 0
        while True:
            idx = [(current + k) % self.n_images for k in range(batch_size)]
            current = (current + batch_size) % self.n_images
            parts = [sd.dct_batch(1, seed=self.seed * 7 + i, split_chroma=deconv) for i in idx]
            x = [np.concatenate([p[j] for p in parts], axis=0) for j in range(len(parts[0]))]
            gt = [self.labels[i] for i in idx]
            ret = []
            if "processed_images" in returns:
                ret.append(x)
            if "encoded_labels" in returns:
                ret.append(label_encoder(gt) if label_encoder is not None else None)
            if "processed_labels" in returns:
                ret.append(gt)
            if "image_ids" in returns:
                ret.append([self.image_ids[i] for i in idx])
            if "inverse_transform" in returns:
                ret.append([[] for _ in idx])
            yield ret
