"""keras.utils bits the reference imports."""


class Sequence(object):
    """keras.utils.Sequence protocol (classification_part/template_keras/generators/template_generator.py:8)."""

    def __getitem__(self, index):
        raise NotImplementedError

    def __len__(self):
        raise NotImplementedError

    def on_epoch_end(self):
        pass

    def __iter__(self):
        while True:
            for i in range(len(self)):
                yield self[i]
            self.on_epoch_end()


def multi_gpu_model(model, gpus=None, **kwargs):
    """Imported but never called by the reference (localisation_part/training_dct_pascal_j2d_resnet.py:65).
    Multi-GPU training here is one process per GPU (see dist.py); this returns the model unchanged."""
    return model
