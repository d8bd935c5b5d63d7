"""Training configuration for the ResNet50-DCT classifiers: same class name, constructor and property surface as
classification_part/config/resnet/config_file.py:25-260 (TrainingConfiguration(deconv, archi, load_pretrained_weights)),
wired to ResNet50Custom(archi=...) -- the committed reference config calls ResNet50RGB(archi=...), which swallows
`archi` and cannot consume the [Y, CbCr] batches its own generators emit; the README's DCT experiments
need ResNet50Custom, so that is what `--archi <dct archi>` builds here.

Data: ImageNet + jpeg2dct are absent, so the generators are synthetic JPEG-DCT batches with one-hot labels (same
emission contract as DCTGeneratorJPEG2DCT / DCTGeneratorJPEG2DCTDeconv, vgg_jpeg_keras/generators/generators.py:39-353).
`prepare_horovod(hvd)` reproduces the reference's data-parallel scaling rules (:121-150) with `hvd` = the RCCL adapter
of training.py."""
from os import environ
from os.path import join

import numpy as np

from jpeg_detection_resnet_ssd_amd.data import synthetic_dct as sd
from jpeg_detection_resnet_ssd_amd.keras.callbacks import (CSVLogger, EarlyStopping, ModelCheckpoint, ReduceLROnPlateau,
                                                           TerminateOnNaN)
from jpeg_detection_resnet_ssd_amd.keras.losses import categorical_crossentropy
from jpeg_detection_resnet_ssd_amd.keras.metrics import top_k_categorical_accuracy
from jpeg_detection_resnet_ssd_amd.keras.optimizers import SGD
from jpeg_detection_resnet_ssd_amd.keras.utils import Sequence
from jpeg_detection_resnet_ssd_amd.vgg_jpeg_keras.networks.resnet_dct import ResNet50Custom, ResNet50RGB


def _top_k_accuracy(k):
    def _func(y_true, y_pred):
        return top_k_categorical_accuracy(y_true, y_pred, k)
    return _func


class SyntheticDCTClassificationGenerator(Sequence):
    """Batches ([Y, CbCr] or [Y, Cb, Cr], one-hot labels) shaped like DCTGeneratorJPEG2DCT(Deconv) output at 224x224."""

    def __init__(self, batch_size, deconv, num_classes=1000, n_batches=64, seed=0):
        self.batch_size, self.deconv, self.num_classes, self.n_batches, self.seed = batch_size, deconv, num_classes, n_batches, seed

    def __len__(self):
        return self.n_batches

    def __getitem__(self, index):
        x = sd.fast_dct_batch(self.batch_size, seed=self.seed + index, grid=28, split_chroma=self.deconv)
        rng = np.random.default_rng(self.seed + 7 * index + 1)
        y = np.zeros((self.batch_size, self.num_classes), dtype=np.float32)
        y[np.arange(self.batch_size), rng.integers(0, self.num_classes, self.batch_size)] = 1.0
        return x, y


class SyntheticRGBGenerator(Sequence):
    def __init__(self, batch_size, num_classes=1000, n_batches=64, seed=0):
        self.batch_size, self.num_classes, self.n_batches, self.seed = batch_size, num_classes, n_batches, seed

    def __len__(self):
        return self.n_batches

    def __getitem__(self, index):
        rng = np.random.default_rng(self.seed + index)
        x = rng.integers(0, 256, size=(self.batch_size, 224, 224, 3)).astype(np.float32)
        y = np.zeros((self.batch_size, self.num_classes), dtype=np.float32)
        y[np.arange(self.batch_size), rng.integers(0, self.num_classes, self.batch_size)] = 1.0
        return [x], y


class TrainingConfiguration(object):
    def __init__(self, deconv=False, archi="late_concat_rfa_thinner", load_pretrained_weights=True):
        self.description = ""
        self.deconv = deconv
        self.archi = archi
        self._workers = 4
        self._multiprocessing = True
        self._gpus = 1
        self._project_name = archi
        self._workspace = "thomasC"
        self.num_classes = 1000
        self.img_size = (224, 224)
        self._weights = None
        weights = "imagenet" if load_pretrained_weights else None
        if archi == "resnet_rgb":
            self._network = ResNet50RGB(weights=weights, archi=archi)
        else:
            self._network = ResNet50Custom(weights=weights, archi=archi)
        self._epochs = 120
        self._batch_size = 256
        self.batch_size_divider = 4
        self._steps_per_epoch = 5000
        self._validation_steps = 50000 // self._batch_size
        self.optimizer_parameters = {"lr": 0.1, "momentum": 0.9, "decay": 0.0001, "nesterov": True}
        self._optimizer = SGD(**self.optimizer_parameters)
        self._loss = categorical_crossentropy
        self._metrics = [_top_k_accuracy(1), _top_k_accuracy(5)]
        self.train_directory = join(environ.get("DATASET_PATH_TRAIN", ""), "imagenet/train")
        self.validation_directory = join(environ.get("DATASET_PATH_VAL", ""), "imagenet/validation")
        self.index_file = join(environ.get("PROJECT_PATH", ""), "data/imagenet_class_index.json")
        self.model_checkpoint = None
        self.csv_logger = None
        self.terminate_on_nan = TerminateOnNaN()
        self.early_stopping = EarlyStopping(monitor="val_loss", min_delta=0, patience=10)
        self.reduce_lr_on_plateau = ReduceLROnPlateau(patience=5, verbose=1)
        self._callbacks = [self.reduce_lr_on_plateau, self.early_stopping, self.terminate_on_nan]
        self._horovod = None
        self._train_generator = None
        self._validation_generator = None
        self._evaluator = None

    # -- callbacks -------------------------------------------------------------------------------
    def add_csv_logger(self, output_path, filename="results.csv", separator=",", append=True):
        if self.horovod is not None and self.horovod.rank() != 0:
            return
        self.csv_logger = CSVLogger(filename=join(output_path, filename), separator=separator, append=append)
        self._callbacks.append(self.csv_logger)

    def add_model_checkpoint(self, output_path, verbose=1, save_best_only=True):
        if self.horovod is not None and self.horovod.rank() != 0:
            return
        self.model_checkpoint = ModelCheckpoint(
            filepath=join(environ["EXPERIMENTS_OUTPUT_DIRECTORY"], "epoch-{epoch:02d}_loss-{loss:.4f}_val_loss-{val_loss:.4f}.h5"),
            verbose=verbose, save_best_only=save_best_only)
        self._callbacks.append(self.model_checkpoint)

    def prepare_horovod(self, hvd):
        """Scaling rules of the reference's Horovod path (config_file.py:121-150): lr *= size / 4, per-rank batch
        = 256 // 4, steps //= size // 4, validation steps = 3 * steps // size; broadcast from rank 0, metric
        averaging and the 5-epoch learning-rate warm-up."""
        self._horovod = hvd
        self._callbacks = [hvd.callbacks.BroadcastGlobalVariablesCallback(0), hvd.callbacks.MetricAverageCallback(),
                           hvd.callbacks.LearningRateWarmupCallback(warmup_epochs=5, verbose=1),
                           ReduceLROnPlateau(patience=5, verbose=1), self.terminate_on_nan, self.early_stopping]
        self.optimizer_parameters["lr"] = self.optimizer_parameters["lr"] * hvd.size() / self.batch_size_divider
        self._optimizer = hvd.DistributedOptimizer(SGD(**self.optimizer_parameters))
        self._batch_size = self._batch_size // self.batch_size_divider
        self._steps_per_epoch = self._steps_per_epoch // max(1, hvd.size() // self.batch_size_divider)
        self._validation_steps = 3 * self._validation_steps // hvd.size()

    def prepare_for_inference(self):
        pass

    def prepare_evaluator(self):
        self._evaluator = None

    def prepare_testing_generator(self):
        pass

    def prepare_training_generators(self):
        rank = self.horovod.rank() if self.horovod is not None else 0
        if self.archi == "resnet_rgb":
            self._train_generator = SyntheticRGBGenerator(self._batch_size, self.num_classes, seed=1000 * rank)
            self._validation_generator = SyntheticRGBGenerator(self._batch_size, self.num_classes, n_batches=8, seed=999983)
        else:
            self._train_generator = SyntheticDCTClassificationGenerator(self._batch_size, self.deconv, self.num_classes,
                                                                        seed=1000 * rank)
            self._validation_generator = SyntheticDCTClassificationGenerator(self._batch_size, self.deconv, self.num_classes,
                                                                             n_batches=8, seed=999983)

    # -- properties (template_keras/config/template_config.py:10-121) ----------------------------------
    workers = property(lambda self: self._workers)
    multiprocessing = property(lambda self: self._multiprocessing)
    gpus = property(lambda self: self._gpus)
    project_name = property(lambda self: self._project_name)
    workspace = property(lambda self: self._workspace)
    network = property(lambda self: self._network)
    epochs = property(lambda self: self._epochs)
    batch_size = property(lambda self: self._batch_size)
    steps_per_epoch = property(lambda self: self._steps_per_epoch)
    validation_steps = property(lambda self: self._validation_steps)
    optimizer = property(lambda self: self._optimizer)
    loss = property(lambda self: self._loss)
    metrics = property(lambda self: self._metrics)
    callbacks = property(lambda self: self._callbacks)
    horovod = property(lambda self: self._horovod)
    train_generator = property(lambda self: self._train_generator)
    validation_generator = property(lambda self: self._validation_generator)
    test_generator = property(lambda self: None)
    evaluator = property(lambda self: self._evaluator)

    @property
    def weights(self):
        return self._weights

    @weights.setter
    def weights(self, value):
        self._weights = value
