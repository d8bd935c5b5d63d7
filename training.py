#!/usr/bin/env python
"""Drop-in for classification_part/training.py: same flags (-r/--restart, -c/--configuration, --horovod True|False,
--use_pretrained_weights, --archi), same config-as-Python-module loading (`config_file.TrainingConfiguration`), same
output-directory layout and restart logic, running on MI355X.

    python3 training.py -c config/resnet --archi deconv --use_pretrained_weights False --horovod False

`--horovod True` = one process per GPU over RCCL (launch with `python -m torch.distributed.run --nproc-per-node N
--master-addr 127.0.0.1 training.py ... --horovod True`); the `hvd` calls the reference makes are served by
jpeg_detection_resnet_ssd_amd.horovod_compat.  Extra optional flags: --epochs / --steps_per_epoch / --batch_size to
shorten a run."""
import argparse
import csv
import random
import string
import sys
from operator import itemgetter
from os import environ, listdir, makedirs
from os.path import dirname, isfile, join
from shutil import copyfile

parser = argparse.ArgumentParser()
parser.add_argument("-r", "--restart", help="Restart the training from a previous stopped config. The argument is the "
                    "path to the experiment folder.", type=str)
parser.add_argument("-c", "--configuration", help="Path to the directory containing the config file to use. The "
                    "configuration file should be named 'config_file.py' (see the examples in the config folder of the "
                    "repository).")
parser.add_argument("--horovod")
parser.add_argument("--use_pretrained_weights", help="Whether to load pretrained weights from Keras for ResnetRGB")
parser.add_argument("--archi", default="late_concat_rfa_thinner", help="""The network architecture to use, value can be :\n
* cb5_only, deconv, up_sampling, up_sampling_rfa, y_cb4_cbcr_cb5, late_concat_rfa_thinner, late_concat_more_channels, resnet_rgb""")
parser.add_argument("--epochs", type=int, default=None)
parser.add_argument("--steps_per_epoch", type=int, default=None)
parser.add_argument("--batch_size", type=int, default=None)
args = parser.parse_args()

if args.horovod == "True":
    args.horovod = True
    from jpeg_detection_resnet_ssd_amd import horovod_compat as hvd
elif args.horovod == "False":
    args.horovod = False
else:
    raise RuntimeError("Please specify if horovod should be used.")
deconv = args.archi == "deconv"
args.use_pretrained_weights = args.use_pretrained_weights != "False"
if args.horovod:
    hvd.init()

DCT_ARCHIS = ["cb5_only", "deconv", "up_sampling", "up_sampling_rfa", "y_cb4_cbcr_cb5", "late_concat_rfa_thinner",
              "late_concat_more_channels"]
restart_epoch = None
restart_lr = None


def build_config(TrainingConfiguration):
    if args.archi in DCT_ARCHIS:
        return TrainingConfiguration(deconv=deconv, archi=args.archi, load_pretrained_weights=args.use_pretrained_weights)
    if args.archi == "resnet_rgb":
        return TrainingConfiguration(load_pretrained_weights=args.use_pretrained_weights)
    return TrainingConfiguration()


if args.restart is not None:
    sys.path.append(join(args.restart, "config"))
    from saved_config import TrainingConfiguration
    config = build_config(TrainingConfiguration)
    key = dirname(join(args.restart, "")).split("_")[-1]
    weights_path = join(args.restart, "checkpoints")
    weights_files = sorted([[f, int(f.split("_")[0].split("-")[1])] for f in listdir(weights_path)
                            if isfile(join(weights_path, f))], key=itemgetter(1))
    config.weights = join(weights_path, weights_files[-1][0])
    restart_epoch = weights_files[-1][1]
    with open(join(args.restart, "results/results.csv"), newline="") as csvfile:
        data = [row for row in csv.reader(csvfile, delimiter=",")]
        restart_lr = float(data[restart_epoch][data[0].index("lr")])
else:
    sys.path.append(args.configuration)
    from config_file import TrainingConfiguration
    config = build_config(TrainingConfiguration)
    key = "".join(random.choice(string.ascii_uppercase + string.ascii_lowercase + string.digits) for _ in range(32))

if args.batch_size:
    config._batch_size = args.batch_size * (config.batch_size_divider if args.horovod else 1)
if args.epochs:
    config._epochs = args.epochs
if args.steps_per_epoch:
    config._steps_per_epoch = args.steps_per_epoch
    config._validation_steps = min(config._validation_steps, max(1, args.steps_per_epoch // 2))

environ.setdefault("EXPERIMENTS_OUTPUT_DIRECTORY", "./experiments")
environ.setdefault("LOG_DIRECTORY", "./logs")
is_root = (args.horovod and hvd.rank() == 0) or (not args.horovod)
if is_root:
    output_dir = join(environ["EXPERIMENTS_OUTPUT_DIRECTORY"], "{}_{}_{}".format(config.workspace, config.project_name, key))
    checkpoints_output_dir = join(output_dir, "checkpoints")
    config_output_dir = join(output_dir, "config")
    results_output_dir = join(output_dir, "results")
    for d in (output_dir, checkpoints_output_dir, config_output_dir, results_output_dir, environ["LOG_DIRECTORY"]):
        makedirs(d, exist_ok=True)

if args.horovod:
    config.prepare_horovod(hvd)
    if args.steps_per_epoch:
        config._steps_per_epoch = args.steps_per_epoch
if is_root:
    config.add_csv_logger(results_output_dir)
    config.add_model_checkpoint(checkpoints_output_dir)
    # the reference only saves the config when horovod is on (training.py:145-156), which makes plain runs
    # impossible to restart; it is saved in both cases here
    src = join(args.configuration, "config_file.py") if args.restart is None else join(args.restart, "config/saved_config.py")
    copyfile(src, join(config_output_dir, "saved_config.py"))

model = config.network
if config.weights is not None and is_root:
    print("Loading weights (by name): {}".format(config.weights))
    model.load_weights(config.weights, by_name=True)
if args.restart is not None:
    config.optimizer.iterations = config.steps_per_epoch * restart_epoch
    config.optimizer.lr = restart_lr

config.prepare_training_generators()
model.compile(loss=config.loss, optimizer=config.optimizer, metrics=config.metrics)
model.fit_generator(config.train_generator, validation_data=config.validation_generator, epochs=config.epochs,
                    steps_per_epoch=config.steps_per_epoch, callbacks=config.callbacks, workers=config.workers,
                    validation_steps=config.validation_steps, use_multiprocessing=config.multiprocessing,
                    initial_epoch=restart_epoch or 0)
