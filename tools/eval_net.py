#!/usr/bin/env python3
"""Counterpart of the reference's `tools/plain_train_net.py --eval-only` for the MI355X engine
(plain_train_net.py:521-544 setup, :573-589 build/load/test): config YAML + optional weights ->
run the detector over images and dump per-image predictions (the evaluator's dict layout,
dafne/evaluation/dafne_evaluator.py:44-58).  Without a dataset (none ships here) it runs on
synthetic uint8 BGR images with seeded random weights -- config #1 of BASELINE.json is
`--config-file configs/hrsc_r50.yaml --height 800 --width 1216 --num-images 1`.

    python tools/eval_net.py --config-file configs/hrsc_r50.yaml [--weights model.pth] [KEY VALUE ...]

Multi-GPU (plain_train_net.py:660-671 `launch(main, num_gpus, ...)`; dafne_evaluator.py:60-64): one process per GPU,
contiguous image shards, one RCCL gather of the packed detections to rank 0, rank 0 writes the outputs.  Either

    python tools/eval_net.py --num-gpus 8 --config-file ... --num-images 64 --batch 8
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/eval_net.py --config-file ...

(the first form spawns the workers itself, like detectron2's launch; the second reads RANK / LOCAL_RANK / WORLD_SIZE).
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-file", required=True)
    ap.add_argument("--weights", default="")
    ap.add_argument("--num-images", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="images per detector call (detectron2's inference_on_dataset uses 1)")
    ap.add_argument("--num-gpus", type=int, default=1, help="spawn this many worker processes, one per GPU")
    ap.add_argument("--height", type=int, default=0, help="synthetic image height (default INPUT.MIN_SIZE_TEST)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--image-dir", default="",
                    help="read the images from this directory (PNG / JPEG / BMP / TIFF) through dafne_amd.data.build_test_loader -- the "
                         "reference's build_test_loader: PIL decode on worker threads, INPUT.RESIZE_TYPE resize on the GPU -- instead "
                         "of synthetic tiles; --num-images caps the count (0: all)")
    ap.add_argument("--write-synthetic-dir", default="",
                    help="first write --num-images synthetic tiles (smooth content + noise: PNGs that compress like aerial tiles) into "
                         "this directory, then use it as --image-dir: the file-fed loop on a box without a dataset")
    ap.add_argument("--decode-workers", type=int, default=32,
                    help="host workers decoding image files ahead of the GPU (capped at the CPUs this process may use, less two)")
    ap.add_argument("--decode-backend", choices=["thread", "process"], default="thread")
    ap.add_argument("--tta", action="store_true", help="TEST.AUG multi-scale / flip inference")
    ap.add_argument("--tta-shard-views", action="store_true",
                    help="with --tta on several GPUs: shard the VIEWS of every image over the ranks (merge NMS on rank 0) instead "
                         "of sharding the images -- the latency form: one image's 27 views take 1/N of the time")
    ap.add_argument("--images-on", choices=["host", "device"], default="host",
                    help="where the synthetic tiles live when the loop starts: host (pageable CPU tensors, as a data loader yields them; "
                         "uploaded through pinned staging under the previous batch) or device (resident in HBM, as bench.py times)")
    ap.add_argument("--serial", action="store_true", help="synchronous model(inputs) per batch instead of the streamed loop")
    ap.add_argument("--warmup-batches", type=int, default=4,
                    help="batches run through the detector before the loop is timed (launch plans, packed weights, HIP graphs, pinned buffers "
                         "are built on first use); their results are discarded")
    ap.add_argument("--output", default="")
    ap.add_argument("--task1-dir", default="", help="DOTA configs: write Task1_<class>.txt files here and merge the tiles "
                                                    "(Task1_merged/) with the device NMS (dota_evaluation.py:110-184)")
    ap.add_argument("--dataset-name", default="",
                    help="score the predictions as the reference's evaluator of this dataset does (get_evaluator, tools/plain_train_net.py:"
                         "171-214: a name containing dota / hrsc / ucas): Task1 files + VOC07 AP per class against the annotations under "
                         "--dataset-root (labelTxt/ | labelXml/ | Annotations/), written to --eval-dir.  Images: --image-dir, or -- with "
                         "$DAFNE_DATA_DIR set -- the registered dataset of that name (dota_1_5_val_1024, hrsc_test, ucas_aod_test, ...)")
    ap.add_argument("--dataset-root", default="", help="the dataset's root directory (MetadataCatalog's root_dir in the reference)")
    ap.add_argument("--eval-dir", default="", help="output folder of --dataset-name (default OUTPUT_DIR/inference/<dataset name>)")
    ap.add_argument("opts", nargs=argparse.REMAINDER, help="KEY VALUE config overrides")
    return ap.parse_args(argv)


def synthetic_inputs(n, h, w, seed, lo=0, hi=None, pixels=True):
    """Image i is a pure function of (seed, i): every rank builds exactly its own shard [lo, hi); pixels=False gives the
    metadata only (rank 0 needs ids / file names of ALL images, not their pixels)."""
    out = []
    for i in range(lo, n if hi is None else hi):
        d = {"height": h, "width": w, "image_id": i, "file_name": "P%04d__1__0___%d.png" % (i // 4, 824 * (i % 4))}
        if pixels:
            g = torch.Generator().manual_seed(seed * 1000003 + i)
            d["image"] = torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8)
        out.append(d)
    return out


def write_synthetic_tiles(root, n, h, w, seed):
    """n PNG tiles named like DOTA's split tiles; smooth low-frequency content plus noise, so that the files compress (and
    decode) like photographs rather than like random bytes."""
    import numpy as np
    from PIL import Image
    os.makedirs(root, exist_ok=True)
    for i in range(n):
        g = torch.Generator().manual_seed(seed * 1000003 + i)
        low = torch.rand(1, 3, max(h // 32, 2), max(w // 32, 2), generator=g)
        img = torch.nn.functional.interpolate(low, size=(h, w), mode="bilinear", align_corners=False)[0]
        img = (img * 220 + torch.rand(3, h, w, generator=g) * 12).clamp_(0, 255).to(torch.uint8)
        Image.fromarray(np.ascontiguousarray(img.permute(1, 2, 0).numpy())).save(
            os.path.join(root, "P%04d__1__0___%d.png" % (i // 4, 824 * (i % 4))), compress_level=3)


def run(args, rank=0, world=1, local_rank=0):
    import dafne_amd.modeling  # noqa: F401  (registers the classes)
    from dafne_amd.checkpoint import load_weights
    from dafne_amd.config import load_cfg
    from dafne_amd.evaluation.driver import inference_on_images, instances_to_rows
    from dafne_amd.evaluation.gather import shard_range, to_predictions
    from dafne_amd.modeling.tta import OneStageRCNNWithTTA
    from dafne_amd.registry import build_model

    cfg = load_cfg(args.config_file, args.opts)
    if not torch.cuda.is_available():
        raise SystemExit("eval_net.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    model = build_model(cfg)
    if args.weights or cfg.MODEL.WEIGHTS:
        missing, unexpected = load_weights(model, args.weights or cfg.MODEL.WEIGHTS)
        if rank == 0:
            print("loaded weights: %d missing, %d unexpected keys" % (len(missing), len(unexpected)))
    else:
        import bench
        model.load_state_dict(bench.seeded_state_dict(model, args.seed))
    model.to(dev)
    model.invalidate()
    h = args.height or cfg.INPUT.MIN_SIZE_TEST
    w = args.width or cfg.INPUT.MIN_SIZE_TEST
    n = args.num_images
    records = None
    if args.write_synthetic_dir:
        if rank == 0 and not os.path.isdir(args.write_synthetic_dir):
            write_synthetic_tiles(args.write_synthetic_dir, n, h, w, args.seed)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        args.image_dir = args.write_synthetic_dir
    if args.dataset_name and not args.image_dir and os.environ.get("DAFNE_DATA_DIR"):
        # the reference's way (plain_train_net.py:568-570, do_test): the dataset by its registered name below $DAFNE_DATA_DIR
        from dafne_amd.data import DatasetCatalog, MetadataCatalog, register_all
        register_all(cfg)
        records = DatasetCatalog.get(args.dataset_name)
        args.image_dir = MetadataCatalog.get(args.dataset_name).image_root
        args.dataset_root = args.dataset_root or MetadataCatalog.get(args.dataset_name).root_dir
    elif args.image_dir:
        from dafne_amd.data import list_image_records
        records = list_image_records(args.image_dir)
    if records is not None:
        if args.tta_shard_views:
            raise SystemExit("--tta-shard-views runs on synthetic tiles (every rank needs every image); --tta alone takes --image-dir")
        if n > 0:
            records = records[:n]
        n = len(records)
    lo, hi = shard_range(n, rank, world)
    mine = synthetic_inputs(n, h, w, args.seed, lo, hi) if records is None else []
    k_cap = model.proposal_generator.dafne_outputs.packed_k_cap()      # the detector's own capacity rule
    tta = OneStageRCNNWithTTA(cfg, model) if args.tta else None
    if cfg.ENGINE.WEIGHT_DTYPE == "fp8_e4m3" and model.fp8_act_scales() is None and cfg.ENGINE.FP8_ACT_CALIBRATION == "explicit":
        # no scales came with the weights: calibrate on the first batch of every rank's shard; calibrate_fp8 MAX-reduces the
        # amax values over the ranks (a collective), so all ranks serve the same quantised model at any world size.  A rank
        # with an empty shard contributes the batch of image 0.
        if records is not None:
            from dafne_amd.data import DAFNeTestMapper
            mp = DAFNeTestMapper(cfg, dev)
            cal = [mp(r) for r in (records[lo:hi][:args.batch] or records[:1])]
            if cal:
                model.calibrate_fp8(model._pack_inputs(cal)[0])
        else:
            cal = mine[:args.batch] if mine else synthetic_inputs(n, h, w, args.seed, 0, min(1, n))
            if cal:
                model.calibrate_fp8(torch.stack([x["image"] for x in cal]).to(dev))

    def detect_batch(b0, b1):                # TTA: one merged Instances per image -> packed rows for the gather
        chunk = mine[b0 - lo:b1 - lo]
        insts = [o["instances"] for o in tta(chunk)]
        return instances_to_rows(insts, k_cap, dev)

    meta = synthetic_inputs(n, h, w, args.seed, pixels=False) if records is None else None
    if tta is not None and args.tta_shard_views:
        # SURVEY 8(e), configs[3]: every rank sees every image and runs ITS share of the image's views; one gather per image
        # lands the per-view detections on rank 0, which inverts, concatenates and runs the merged NMS (tta.py:173-197,264-268)
        everyone = synthetic_inputs(n, h, w, args.seed)
        merged = [tta.inference_view_sharded(x, rank=rank, world=world, device=dev) for x in everyone]
        torch.cuda.synchronize()
        if rank != 0:
            return None
        out = instances_to_rows([o["instances"] for o in merged], k_cap, dev)
    elif tta is not None and records is None:
        out = inference_on_images(detect_batch, n, k_cap, batch_size=args.batch, rank=rank, world=world, device=dev)
        torch.cuda.synchronize()
        if rank != 0:
            return None
    else:
        # the reference's loop: inference_on_dataset(model, data_loader, evaluator) (plain_train_net.py:316-336), streamed --
        # batch i runs on the benchmarked layout while batch i - 1's outputs go to the evaluator.  With --tta and image files it is
        # do_test_with_TTA's (plain_train_net.py:338-356): the same loader, the model wrapped in OneStageRCNNWithTTA (called
        # synchronously per batch; the wrapper groups and pipelines the views of a batch's images itself)
        from dafne_amd.evaluation.inference import DafneEvaluator, inference_on_dataset
        runner = tta if tta is not None else model
        if args.images_on == "device":
            for x in mine:
                x["image"] = x["image"].to(dev)
            torch.cuda.synchronize()
        b = max(args.batch, 1)
        if records is not None:
            from dafne_amd.data.loader import InferenceLoader
            loader = InferenceLoader(cfg, records, batch_size=b, device=dev, num_workers=args.decode_workers, shard=(rank, world),
                                     backend=args.decode_backend, prefetch_batches=4)
        else:
            loader = [mine[i:i + b] for i in range(0, len(mine), b)]
        ev = DafneEvaluator("synthetic", cfg, distributed=world > 1, k_cap=k_cap, device=dev, pad_to=(n + world - 1) // world)
        stats = {}
        if args.warmup_batches > 0 and len(loader):
            import itertools
            warm = (loader * args.warmup_batches)[:args.warmup_batches] if isinstance(loader, list) else \
                list(itertools.islice(itertools.cycle(list(itertools.islice(iter(loader), 2))), args.warmup_batches))
            inference_on_dataset(runner, warm, None)     # (two per plan set: eager, then graph capture)
        if args.serial:
            class _Sync:                      # the synchronous form: model(inputs) per batch
                def __init__(self, m):
                    self.m = m

                def __call__(self, inputs):
                    return self.m(inputs)
            res = inference_on_dataset(_Sync(runner), loader, ev, stats)
        else:
            res = inference_on_dataset(runner, loader, ev, stats)
        print("rank %d: inference_on_dataset %d images in %.3f s = %.1f images/s (batch %d, %s, images on the %s)"
              % (rank, stats["images"], stats["seconds"], stats["images_per_sec"], b,
                 "TTA, synchronous per batch" if tta is not None else
                 "synchronous" if args.serial else "streamed, %d sub-batch streams" % cfg.ENGINE.PIPELINE_SPLITS,
                 args.images_on if records is None else "disk (%s, %d decode %s workers)" % (args.image_dir, loader.num_workers, args.decode_backend)), flush=True)
        if rank != 0:
            return None
        preds = res["predictions"]
        if meta is not None:
            by_id = {m["image_id"]: m for m in meta}
            for p in preds:
                m = by_id[p["image_id"]]
                p.update(file_name=m["file_name"], height=m["height"], width=m["width"])
        preds.sort(key=lambda p: p["image_id"])
        out = None
    if out is not None:
        rows_all, counts_all = out
        preds = to_predictions(rows_all, counts_all, image_ids=[m["image_id"] for m in meta])
        for p, m in zip(preds, meta):
            p.update(file_name=m["file_name"], height=m["height"], width=m["width"])
    for p in preds:
        print("image %s: %d detections, best score %.4f" % (p["image_id"], len(p["scores"]),
                                                             float(p["scores"].max()) if len(p["scores"]) else 0.0))
    if args.output:
        torch.save(preds, args.output)
    if args.task1_dir:
        from dafne_amd.evaluation import dota_evaluation as de
        names = list(de.CLASSNAMES_DOTA_1_0) + (["container-crane"] if cfg.MODEL.DAFNE.NUM_CLASSES == 16 else [])
        t1 = os.path.join(args.task1_dir, "Task1")
        merged = os.path.join(args.task1_dir, "Task1_merged")
        os.makedirs(t1, exist_ok=True)
        os.makedirs(merged, exist_ok=True)
        de._generate_task_1_files(None, preds, args.task1_dir, t1, names[:cfg.MODEL.DAFNE.NUM_CLASSES], cfg)
        de.run_merge(t1, merged)
        n_in = sum(len(open(os.path.join(t1, f)).readlines()) for f in os.listdir(t1))
        n_out = sum(len(open(os.path.join(merged, f)).readlines()) for f in os.listdir(merged))
        print("Task1: %d tile detections -> %d after the tile merge (%s)" % (n_in, n_out, merged))
    if args.dataset_name:
        import types
        from collections import OrderedDict
        from dafne_amd.evaluation.inference import get_evaluator
        ev2 = get_evaluator(cfg, args.dataset_name, output_folder=args.eval_dir or None, distributed=False,
                            metadata=types.SimpleNamespace(root_dir=args.dataset_root, is_test="test" in args.dataset_name.lower()))
        os.makedirs(ev2._output_dir, exist_ok=True)
        ev2._results = OrderedDict()
        ev2._eval_predictions(preds)
        for k, v in ev2._results.get("task1", {}).items():
            print("%-18s: %.4f" % (k, v))
    return preds


def _worker(local_rank, world, port, args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(local_rank), LOCAL_RANK=str(local_rank),
                      WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    _distributed_main(args)


def _distributed_main(args):
    import torch.distributed as dist
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    out = run(args, rank, world, local_rank)        # an exception propagates: no barrier on the error path (it would hang
    dist.barrier()                                  # the healthy ranks or mask the error); the launcher tears the job down
    dist.destroy_process_group()
    return out


def main(argv=None):
    args = parse_args(argv)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:            # under torch.distributed.run
        return _distributed_main(args)
    if args.num_gpus > 1:                                      # plain_train_net.py:660-671: launch one process per GPU
        import socket
        import torch.multiprocessing as mp
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_worker, args=(args.num_gpus, port, args), nprocs=args.num_gpus, join=True)
        return None
    return run(args)


if __name__ == "__main__":
    main()
