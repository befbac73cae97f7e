"""YAML experiment configs and logging, in the reference's formats (fuxictr/utils.py:26-104): ``model_config.yaml`` keyed
by expid (+ optional ``Base``), then the dataset entry keyed by ``dataset_id``, everything flattened into one params dict.

Deviation that FIXES a reference defect: the shipped dataset YAMLs live in ``configs/datasets/`` where the
reference's ``load_dataset_config`` never looks; here ``<config_dir>/../datasets/*.yaml`` and
``<config_dir>/../../datasets/*.yaml`` are searched too, so the shipped command lines work unmodified."""
import glob
import json
import logging
import os
from collections import OrderedDict

import yaml


def load_dataset_config(config_dir, dataset_id):
    candidates = glob.glob(os.path.join(config_dir, "dataset_config.yaml"))
    candidates += glob.glob(os.path.join(config_dir, "dataset_config", "*.yaml"))
    candidates += glob.glob(os.path.join(config_dir, "..", "datasets", "*.yaml"))
    candidates += glob.glob(os.path.join(config_dir, "..", "..", "datasets", "*.yaml"))
    for path in candidates:
        with open(path, "r") as fh:
            blob = yaml.load(fh, Loader=yaml.FullLoader) or {}
        if dataset_id in blob:
            return blob[dataset_id]
    raise RuntimeError("dataset_id={} is not found in config.".format(dataset_id))


def load_config(config_dir, experiment_id):
    model_configs = glob.glob(os.path.join(config_dir, "model_config.yaml")) or \
        glob.glob(os.path.join(config_dir, "model_config", "*.yaml"))
    if not model_configs:
        raise RuntimeError("config_dir={} is not valid!".format(config_dir))
    found = {}
    for path in model_configs:
        with open(path, "r") as fh:
            blob = yaml.load(fh, Loader=yaml.FullLoader) or {}
        if "Base" in blob:
            found["Base"] = blob["Base"]
        if experiment_id in blob:
            found[experiment_id] = blob[experiment_id]
        if len(found) == 2:
            break
    if experiment_id not in found:
        raise ValueError("expid={} not found in config".format(experiment_id))
    params = dict(found.get("Base", {}))
    params.update(found[experiment_id])
    params["model_id"] = experiment_id
    params.update(load_dataset_config(config_dir, params["dataset_id"]))
    return params


def set_logger(params, log_file=None):
    if log_file is None:
        log_file = os.path.join(params["model_root"], params["dataset_id"], params["model_id"] + ".log")
    os.makedirs(os.path.dirname(log_file), exist_ok=True)
    for handler in logging.root.handlers[:]:
        logging.root.removeHandler(handler)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s P%(process)d %(levelname)s %(message)s",
                        handlers=[logging.FileHandler(log_file, mode="w"), logging.StreamHandler()])


def print_to_json(data, sort_keys=True):
    items = dict((k, str(v)) for k, v in data.items())
    if sort_keys:
        items = OrderedDict(sorted(items.items(), key=lambda kv: kv[0]))
    return json.dumps(items, indent=4)


def print_to_list(data):
    return " - ".join("{}: {:.6f}".format(k, v) for k, v in data.items())
