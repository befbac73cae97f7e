#!/usr/bin/env python3
"""Experiment entry with the reference's command line (run_expid.py:27-115 of the reference):

    python run_expid.py --config ./configs/RAT_m2/kkbox_x1 --expid RAT_m2_kkbox_x1_10fold_retrieval --gpu 0

load config -> feature_map.json -> batch sources -> ``getattr(models, params["model"])(feature_map, **params)`` ->
fit_generator -> load best checkpoint -> evaluate on valid/test -> append one CSV result line.  The model class is this
repo's HIP-backed plugin.  Preprocessing (CSV -> ids) and BM25 retrieval are offline steps outside the hot path: the
data directory must already hold ``feature_map.json``, ``{train,valid,test}.{h5|npz}`` and
``retrieval_{K}_{split}.{h5|npz}``; with ``--synthetic N`` a structured synthetic dataset of N training rows shaped
like the feature map (or like the named bench workload when no feature map exists) is generated instead.
"""
import argparse
import datetime
import gc
import logging
import os
import sys

ROOT = os.path.dirname(os.path.realpath(__file__))
sys.path.insert(0, os.path.join(ROOT, "www24-rat_amd"))

import rat_amd  # noqa: E402
from rat_amd import data as rat_data  # noqa: E402
from rat_amd import models, synthetic  # noqa: E402
from rat_amd.base_model import seed_everything  # noqa: E402
from rat_amd.config import load_config, print_to_json, print_to_list, set_logger  # noqa: E402
from rat_amd.features import FeatureMap  # noqa: E402


def _find(data_dir, stem):
    for ext in (".npz", ".h5"):
        p = os.path.join(data_dir, stem + ext)
        if os.path.exists(p):
            return p
    return None


def distributed_env():
    """(rank, local_rank, world) from a torch.distributed launcher's environment (torchrun / `python -m torch.distributed.run`);
    (0, 0, 1) when there is none — the reference's single-device entry (run_expid.py:27-40, torch_utils.py:34-39)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world


def build_sources(params, feature_map, synthetic_rows, shard=(0, 1)):
    """train / valid / test batch sources.  Under data parallelism (`shard` = (rank, world)) only the TRAINING source is sharded —
    `batch_size` is the global batch, each rank takes its equal slice of every batch; validation and test run unsharded on
    every rank (identical metrics everywhere, rank 0's decide: BaseModel._agreed_value)."""
    topk = params["retrieval_configs"]["topK"]
    bs = params["batch_size"]
    if synthetic_rows:
        out = {}
        for split, n, seed in (("train", synthetic_rows, 1), ("valid", max(synthetic_rows // 8, bs), 2),
                               ("test", max(synthetic_rows // 8, bs), 3)):
            data, idx, val, lens = rat_data.synthetic_split(feature_map, n, topk, seed)
            out[split] = rat_data.RetrievalBatches(data, data, idx, val, lens, bs, shuffle=(split == "train") and params.get("shuffle", True),
                                                   seed=params.get("seed", 0), shard=shard if split == "train" else (0, 1))
        return out["train"], out["valid"], out["test"]
    data_dir = os.path.join(params["data_root"], params["dataset_id"])
    rcfg = params["retrieval_configs"]
    folds = "fold" in rcfg.get("split_type", "")
    out = []
    for split in ("train", "valid", "test"):
        dpath, rpath = _find(data_dir, split), _find(data_dir, "retrieval_%d_%s" % (topk, split))
        if dpath is None:
            if split == "test":
                out.append(None)
                continue
            raise RuntimeError("missing %s.{npz,h5} under %s (encoded by the reference's build_dataset, or pass --synthetic N)" % (split, data_dir))
        # pool per split as h5_generator chooses it (datasets/data_utils.py:1218-1226,1257-1262): <X>-fold -> the training split
        # retrieves from its own other folds, valid / test from the training data; otherwise the separate retrieval pool
        if folds:
            pool = None if split == "train" else _find(data_dir, "train")
        else:
            pool = _find(data_dir, "retrieval_pool")
        if shard[1] > 1:
            # every rank takes the SAME branch: rank 0 looks and tells the others (a rank that starts later could otherwise see the file
            # rank 0 is still writing, skip the wait and run into rank 0's barrier with its next collective)
            import torch
            import torch.distributed as dist
            flag = torch.tensor([1 if rpath is not None else 0], dtype=torch.int32)
            if dist.get_backend() == "nccl":
                flag = flag.cuda()
            dist.broadcast(flag, src=0)
            have = bool(int(flag.cpu()[0]))
        else:
            have = rpath is not None
        if not have:
            # stored next to the data in the data's own format: retrieval_{K}_{split}.h5 — the reference's file name and keys
            # (data_generator.py:106-113) — when the split is an HDF5 file, .npz otherwise
            rpath = os.path.join(data_dir, "retrieval_%d_%s%s" % (topk, split, ".h5" if dpath.endswith(".h5") else ".npz"))
            # the marker is this JOB's (rendezvous port in its name): a poller can never mistake what an earlier, failed run left
            # behind for rank 0's verdict on this one
            failed = rpath + ".failed.%s" % os.environ.get("MASTER_PORT", "0")
            if shard[0] == 0:                                   # one rank computes and writes the file (atomically), the others wait
                try:
                    precompute_retrieval_file(dpath, pool, rpath, rcfg, feature_map, params)
                except BaseException as exc:                    # tell the pollers before dying: they must not wait for a file that will never come
                    if shard[1] > 1:
                        with open(failed, "w") as f:
                            f.write("%s: %s\n" % (type(exc).__name__, exc))
                    raise
            if shard[1] > 1:
                _wait_for_file(rpath, leader=shard[0] == 0, failed_marker=failed)
        elif rpath is None:                                      # (this rank's listing was older than rank 0's)
            rpath = _find(data_dir, "retrieval_%d_%s" % (topk, split))
        out.append(rat_data.batches_from_files(dpath, rpath, bs, pool_path=pool, shuffle=(split == "train") and params.get("shuffle", True),
                                               seed=params.get("seed", 0), shard=shard if split == "train" else (0, 1)))
    return out


_PROCESS_START = __import__("time").time()      # a failure marker older than this process belongs to an earlier job


def _wait_for_file(path, leader, poll_s=2.0, failed_marker=None, max_wait_s=None):
    """The top-K pre-computation of a large split can take longer than a collective's timeout (10 minutes by default), so the other ranks
    do not sit in a barrier: they poll for the finished file (written under a temporary name and renamed, so existence means complete)
    and only then meet rank 0 in a barrier that returns at once.  They give up — exit non-zero — when rank 0 leaves the
    `failed_marker` behind (its pre-computation raised) or after `max_wait_s` seconds (default: RAT_RETRIEVAL_WAIT_S, else 6 hours)."""
    import time
    import torch.distributed as dist
    if not leader:
        if max_wait_s is None:
            max_wait_s = float(os.environ.get("RAT_RETRIEVAL_WAIT_S", 6 * 3600))
        t_end = time.monotonic() + max_wait_s
        while not os.path.exists(path):
            # (a marker older than this process is a previous job's that happened to use the same port)
            if failed_marker is not None and os.path.exists(failed_marker) and os.path.getmtime(failed_marker) >= _PROCESS_START - 5.0:
                with open(failed_marker) as f:
                    raise SystemExit("rank 0 could not write %s: %s" % (path, f.read().strip()))
            if time.monotonic() > t_end:
                raise SystemExit("gave up waiting for %s after %.0f s (rank 0 is computing it; RAT_RETRIEVAL_WAIT_S sets the limit)"
                                 % (path, max_wait_s))
            time.sleep(poll_s)
    dist.barrier()


def precompute_retrieval_file(data_path, pool_path, save_path, rcfg, feature_map, params):
    """DataGenerator's pre-retrieval branch (fuxictr/pytorch/data_generator.py:106-215) with the top-K search on the device
    (rat_bm25_topk); the result is stored next to the data like the reference's retrieval_{K}_{split}.h5 (keys indices /
    values / lens) — as HDF5 through rat_amd.h5io when the split itself is an .h5 file, as .npz otherwise."""
    import numpy as np
    from rat_amd import retrieval
    if params["gpu"] < 0:
        raise RuntimeError("%s does not exist and computing it needs a GPU (--gpu >= 0): the retrieval kernel has no CPU fallback" % save_path)
    data = rat_data.load_array_file(data_path, ["data"])["data"]
    pool = None if pool_path is None else rat_data.load_array_file(pool_path, ["data"])["data"]
    logging.info("retrieval file %s not found: computing top-%d on the device (%d queries, pool %s)", save_path, rcfg["topK"], len(data),
                 "self / %s" % rcfg.get("split_type") if pool is None else "%d rows" % len(pool))
    cols = retrieval.used_col_indices(feature_map, rcfg)
    indices, values, lens = retrieval.precompute_retrieval(data, rcfg, cols, pool_array=pool, device="cuda:%d" % params["gpu"])
    root, ext = os.path.splitext(save_path)
    tmp = "%s.tmp%d%s" % (root, os.getpid(), ext)                # same directory, same extension (np.savez appends .npz otherwise)
    try:
        if save_path.endswith(".h5"):
            from rat_amd import h5io
            h5io.write_arrays(tmp, {"indices": np.asarray(indices), "values": np.asarray(values), "lens": np.asarray(lens)})
        else:
            np.savez_compressed(tmp, indices=indices, values=values, lens=lens)
        os.replace(tmp, save_path)                               # atomic: a reader never sees a partial file
    finally:
        if os.path.exists(tmp):                                  # (a failed write / rename: nothing half-written stays behind)
            os.remove(tmp)


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--version", type=str, default="pytorch", help="The model version.")
    parser.add_argument("--config", type=str, default="../configs/", help="The config directory.")
    parser.add_argument("--expid", type=str, default="FM_test", help="The experiment id to run.")
    parser.add_argument("--gpu", type=int, default=-1, help="The gpu index, -1 for cpu")
    parser.add_argument("--synthetic", type=int, default=0, help="generate N synthetic training rows instead of reading data")
    parser.add_argument("--epochs", type=int, default=None, help="override the config's epochs")
    parser.add_argument("--host_batches", action="store_true",
                        help="assemble batches on the host like the reference's DataLoader (default on a GPU: the data stays "
                             "resident in HBM and every batch is one rat_batch_assemble launch)")
    args = vars(parser.parse_args(argv))
    params = load_config(args["config"], args["expid"])
    params["gpu"], params["version"] = args["gpu"], args["version"]
    if args["epochs"] is not None:
        params["epochs"] = args["epochs"]
    # data parallelism: under a torch.distributed launcher every process is one rank — one GPU each (LOCAL_RANK
    # replaces --gpu), RCCL process group over 127.0.0.1 / xGMI (gloo when --gpu -1: the CPU tests), the SAME seed on every rank
    # (identical initial replicas), the training batches sharded by rank, rank 0 alone logs, checkpoints and writes the result line
    rank, local_rank, world = distributed_env()
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args["gpu"] >= 0:
            params["gpu"] = local_rank
            torch.cuda.set_device(local_rank)
        if not dist.is_initialized():
            dist.init_process_group("nccl" if args["gpu"] >= 0 else "gloo", rank=rank, world_size=world)
    if rank == 0:
        set_logger(params)
    else:                                    # the other ranks keep their own (quiet) log next to rank 0's
        set_logger(params, log_file=os.path.join(params["model_root"], params["dataset_id"], "%s.rank%d.log" % (params["model_id"], rank)))
        logging.getLogger().setLevel(logging.WARNING)
    logging.info(print_to_json(params))
    seed_everything(seed=params["seed"])

    data_dir = os.path.join(params["data_root"], params["dataset_id"])
    feature_map = FeatureMap(params["dataset_id"], data_dir, params["version"])
    json_file = os.path.join(data_dir, "feature_map.json")
    if os.path.exists(json_file):
        feature_map.load(json_file)
    elif args["synthetic"]:
        spec = synthetic.WORKLOADS["tiny"]
        feature_map = synthetic.feature_map_for(params["dataset_id"], spec)
    else:
        raise RuntimeError("feature_map not exist!")
    train_gen, valid_gen, test_gen = build_sources(params, feature_map, args["synthetic"], shard=(rank, world))

    model_class = getattr(models, params["model"])
    model = model_class(feature_map, **params)
    model.count_parameters()
    if model.device.type == "cuda" and not args["host_batches"]:
        train_gen, valid_gen = train_gen.to_device(model.device), valid_gen.to_device(model.device)
        test_gen = test_gen.to_device(model.device) if test_gen else test_gen
    model.fit_generator(train_gen, validation_data=valid_gen, **params)

    logging.info("Load best model: {}".format(model.checkpoint))
    model.load_weights(model.checkpoint)
    logging.info("****** Validation evaluation ******")
    valid_result = model.evaluate_generator(valid_gen)
    del train_gen, valid_gen
    gc.collect()
    logging.info("******** Test evaluation ********")
    test_result = model.evaluate_generator(test_gen) if test_gen else {}

    if rank == 0:
        result_file = os.path.join(params["model_root"], params["dataset_id"], params["model_id"] + ".csv")
        with open(result_file, "a+") as fw:
            fw.write(" {},[command] python {},[exp_id] {},[dataset_id] {},[train] {},[val] {},[test] {}\n".format(
                datetime.datetime.now().strftime("%Y%m%d-%H%M%S"), " ".join(sys.argv), args["expid"], params["dataset_id"],
                "N.A.", print_to_list(valid_result), print_to_list(test_result)))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    return valid_result, test_result


if __name__ == "__main__":
    main()
