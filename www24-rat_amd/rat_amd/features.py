"""feature_map.json reader — the model constructor's contract with the (out-of-scope) preprocessing side.

Mirrors what fuxictr/features.py:36-90 (FeatureMap) exposes to a model: ``dataset_id``, ``data_dir``,
``num_fields``, ``feature_specs`` (ordered name -> {type, vocab_size, index[, padding_idx, max_len, encoder]}).
"""
import io
import json
import os
from collections import OrderedDict


class FeatureMap:
    def __init__(self, dataset_id, data_dir=".", version="pytorch"):
        self.dataset_id = dataset_id
        self.data_dir = data_dir
        self.version = version
        self.num_fields = 0
        self.num_features = 0
        self.input_length = 0
        self.feature_specs = OrderedDict()

    def load(self, json_file):
        with io.open(json_file, "r", encoding="utf-8") as fd:
            blob = json.load(fd, object_pairs_hook=OrderedDict)
        if blob["dataset_id"] != self.dataset_id:
            raise RuntimeError("dataset_id={} does not match to feature_map!".format(self.dataset_id))
        self.num_fields = blob["num_fields"]
        self.num_features = blob.get("num_features")
        self.input_length = blob.get("input_length")
        self.feature_specs = OrderedDict(blob["feature_specs"])
        return self

    def save(self, json_file):
        os.makedirs(os.path.dirname(os.path.abspath(json_file)), exist_ok=True)
        blob = OrderedDict(dataset_id=self.dataset_id, num_fields=self.num_fields, num_features=self.num_features,
                           input_length=self.input_length, feature_specs=self.feature_specs)
        with open(json_file, "w") as fd:
            json.dump(blob, fd, indent=4)

    @classmethod
    def from_specs(cls, dataset_id, feature_specs, data_dir="."):
        fm = cls(dataset_id, data_dir)
        fm.feature_specs = OrderedDict(feature_specs)
        fm.num_fields = len(fm.feature_specs)
        fm.input_length = sum(len(s["index"]) if isinstance(s["index"], (list, tuple)) else 1
                              for s in fm.feature_specs.values())
        fm.num_features = sum(s["vocab_size"] for s in fm.feature_specs.values())
        return fm


class FieldInfo:
    """Per-field record handed to the kernels (include/rat_hip.h RatField minus the table pointer)."""
    __slots__ = ("name", "kind", "col", "ncols", "vocab", "padding_idx")

    def __init__(self, name, spec):
        self.name = name
        self.kind = spec["type"]
        index = spec["index"]
        if self.kind == "categorical":
            self.col, self.ncols = int(index), 1
            self.padding_idx = spec.get("padding_idx", None)
        elif self.kind == "sequence":
            cols = list(index)
            if cols != list(range(cols[0], cols[0] + len(cols))):
                raise NotImplementedError("sequence field %s: non-contiguous columns" % name)
            enc = spec.get("encoder", None)
            if enc != "MaskedSumPooling":
                raise NotImplementedError("sequence encoder %r is outside the RAT_m2 hot path "
                                          "(only MaskedSumPooling is used by the shipped configs)" % (enc,))
            self.col, self.ncols = int(cols[0]), len(cols)
            self.padding_idx = spec["vocab_size"] - 1
        else:
            raise NotImplementedError("feature type %r is outside the RAT_m2 hot path" % self.kind)
        if "pretrained_emb" in spec or "share_embedding" in spec:
            raise NotImplementedError("pretrained / shared embeddings are outside the RAT_m2 hot path")
        self.vocab = int(spec["vocab_size"])


def field_infos(feature_map):
    return [FieldInfo(name, spec) for name, spec in feature_map.feature_specs.items()]
