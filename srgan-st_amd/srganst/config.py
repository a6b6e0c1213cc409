"""Config surface - attribute-compatible with reference config.py:15-139 (same tree, same
defaults, same add_g_criterion/remove_g_criterion/get_all_params), plus DIST.* / KERNEL.* knobs.

Differences, all additive:
  * criterion objects default to the HIP-path modules of srganst.loss (same call protocol
    ``criterion(sr, gt) -> 0-dim``); 'Adversarial' keeps its special (D(sr), real_label) call;
  * class-level dicts are copied per instance (the reference shares them between instances,
    config.py:19,33,45,57 - an accident, not an interface).
"""
from __future__ import annotations

import os

import copy

import torch


class dotdict(dict):
    """dot.notation access to dictionary attributes (reference config.py:3-13)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__
    __dir__ = dict.keys
    __repr__ = dict.__repr__


class Config:
    def __init__(self):
        from . import loss as L

        self.DEVICE = "cuda:0" if torch.cuda.is_available() else "cpu"

        self.EXP = dotdict()
        self.EXP.USER = "s204163"
        self.EXP.NAME = "experiment-name"
        self.EXP.START_EPOCH = 0
        self.EXP.N_EPOCHS = 40
        self.EXP.LABEL_SMOOTHING = 0.1

        self.LOG_TRAIN_PERIOD = 100
        self.LOG_VALIDATION_PERIOD = 1
        self.D_CHECKPOINT_INTERVAL = 100
        self.G_CHECKPOINT_INTERVAL = 100

        self.DATA = dotdict()
        self.DATA.TRAIN_GT_IMAGES_DIR = f"/work3/{self.EXP.USER}/data/train"
        self.DATA.TEST_SET = "Set5"
        self.DATA.TEST_GT_IMAGES_DIR = f"/work3/{self.EXP.USER}/data/{self.DATA.TEST_SET}/GTmod12"
        self.DATA.TEST_LR_IMAGES_DIR = f"/work3/{self.EXP.USER}/data/{self.DATA.TEST_SET}/LRbicx4"
        self.DATA.TEST_SR_IMAGES_DIR = "results/_test"
        self.DATA.SEED = 0
        self.DATA.UPSCALE_FACTOR = 4
        self.DATA.BATCH_SIZE = 16
        self.DATA.GT_IMAGE_SIZE = 96

        self.MODEL = dotdict()
        self.MODEL.G_CONTINUE_FROM_WARMUP = False
        self.MODEL.G_WARMUP_WEIGHTS = ""
        self.MODEL.D_CONTINUE_FROM_WARMUP = False
        self.MODEL.D_WARMUP_WEIGHTS = ""
        self.MODEL.G_IN_CHANNEL = 3
        self.MODEL.G_OUT_CHANNEL = 3
        self.MODEL.G_N_CHANNEL = 64
        self.MODEL.G_N_RCB = 16

        self.MODEL.G_LOSS = dotdict()
        self.MODEL.G_LOSS.VGG19_LAYERS = {"features.17": 1 / 8, "features.26": 1 / 4, "features.35": 1 / 2}
        self.MODEL.G_LOSS.DISC_FEATURES_LOSS_LAYERS = {"features.4": 1 / 4, "features.10": 1 / 2}
        self.MODEL.G_LOSS.CRITERIONS = {"Adversarial": L.BCEWithLogitsLoss()}
        self.MODEL.G_LOSS.CRITERION_WEIGHTS = {
            "Adversarial": 0.001, "ContentVGG": 1.0, "ContentDiscriminator": 2000.0, "Pixel": 1.0,
            "BestBuddy": 50.0, "Gram": 500.0, "PatchwiseST": 100.0, "ST": 1 / 3,
        }
        self.MODEL.G_LOSS.WARMUP_CRITERIONS = {"Pixel": L.MSELoss()}
        self.MODEL.G_LOSS.WARMUP_WEIGHTS = {"Pixel": 1.0}
        self.MODEL.D_IN_CHANNEL = 3
        self.MODEL.D_OUT_CHANNEL = 1
        self.MODEL.D_N_CHANNEL = 64

        self.SOLVER = dotdict()
        self.SOLVER.D_UPDATE_INTERVAL = 100
        self.SOLVER.D_OPTIMIZER = "Adam"
        self.SOLVER.D_BASE_LR = 1e-4
        self.SOLVER.D_BETA1 = 0.9
        self.SOLVER.D_BETA2 = 0.999
        self.SOLVER.D_WEIGHT_DECAY = 0
        self.SOLVER.D_EPS = 1e-4
        self.SOLVER.G_OPTIMIZER = "Adam"
        self.SOLVER.G_BASE_LR = 1e-4
        self.SOLVER.G_BETA1 = 0.9
        self.SOLVER.G_BETA2 = 0.999
        self.SOLVER.G_WEIGHT_DECAY = 0
        self.SOLVER.G_EPS = 1e-4

        self.SCHEDULER = dotdict()
        self.SCHEDULER.STEP_SIZE = self.EXP.N_EPOCHS // 2
        self.SCHEDULER.GAMMA = 0.5

        # --- additions (not in the reference) ---
        self.DIST = dotdict()
        self.DIST.BACKEND = "nccl"          # RCCL on ROCm; "gloo" in the CPU tests
        self.DIST.BUCKET_D = True           # D grads in two buckets (features / classifier)
        self.DIST.OVERLAP_COMM = True       # hide the gradient all-reduces behind compute (engine.TrainEngine._step_overlapped)
        # RCCL only: the mean-all-reduces are captured INSIDE the iteration's hipGraph (the process group's stream forks from the branch that
        # produced the bucket and is joined into the graph's origin stream) - the iteration stays ONE graph at N > 1.  Off / gloo: the
        # graphs are cut where a collective goes out (_step_overlapped)
        # OFF: built, bit-identical to the cut-graph schedule (tests/test_dp_gpu.py, RCCL world 1), but measured SLOWER on ROCm 7.2: as
        # soon as the graph holds the collectives' nodes the runtime runs its two compute branches one after the other (device stamps:
        # the generator's backward starts when the discriminator branch has ended) - 5.80 ms against 5.20 ms for the cut graphs and
        # 4.97 ms single-process (tools/time_dp.py, three discriminator forwards)
        self.DIST.ONE_GRAPH = os.environ.get("SST_DP_ONE_GRAPH", "0") != "0"
        # overlapped schedule, N > 1, RCCL: the generator's all-reduce on its own communicator (does not queue behind the discriminator's
        # buckets).  OFF by default: it cannot be exercised here (RCCL needs one GPU per rank, a call has one) and an untested
        # communicator must not be the first thing a multi-GPU run meets; SST_DP_G_OWN_GROUP=1 to try it on a node
        self.DIST.G_OWN_GROUP = os.environ.get("SST_DP_G_OWN_GROUP", "0") != "0"
        self.KERNEL = dotdict()
        self.KERNEL.USE_GRAPH = True        # capture the train step into a hipGraph
        self.KERNEL.SYNC_LOSS_EVERY_STEP = False  # reference does .item() per criterion per step (train.py:141)
        # discriminator step: D(gt) and D(sr) passes as two parallel branches of the graph (engine.TrainEngine._d_two_stream)
        self.KERNEL.D_TWO_STREAMS = os.environ.get("SST_D_TWO_STREAMS", "0") != "0"
        # the discriminator step beside the generator's backward, whole iteration = one graph (engine.TrainEngine._iter_gd)
        self.KERNEL.OVERLAP_GD = os.environ.get("SST_OVERLAP_GD", "1") != "0"
        # the discriminator step's D(sr.detach()) forward (train.py:158) is not run again: it repeats the generator step's D(sr)
        # (same input, same weights, deterministic kernels); its running-statistics side effects are replayed (disc_graph.replay_running_stats)
        # D(gt)'s forward starts with the iteration, beside the generator's forward (running statistics replayed in the reference's order).
        # Off: measured 5.44 vs 5.36 ms - two conv-bound passes side by side just take turns (G forward + D(sr): 1.30 -> 1.84 ms)
        self.KERNEL.EARLY_D_GT = os.environ.get("SST_EARLY_D_GT", "0") != "0"
        # merged iteration: the conv weight gradients of the discriminator's last backward pass run on the generator's stream after its
        # backward (the discriminator branch is the longer one)
        # (= how many layers, counted from the first: the ones the backward chain reaches last; 0 = none, 8 = all)
        self.KERNEL.DEFER_D_WGRAD = int(os.environ.get("SST_DEFER_D_WGRAD", "8"))
        # merged iteration: D's weight packing on the side stream beside the generator's forward
        # ... with the two passes batched (BATCH_D_STEP) the side branch is the shorter one: nothing is moved (measured: 5.03 ms with
        # 0 layers deferred, 5.05 / 5.08 / 5.09 / 5.11 with 1 / 2 / 3 / 4, 5.17 with all 8)
        self.KERNEL.DEFER_D_WGRAD_BATCHED = int(os.environ.get("SST_DEFER_D_WGRAD_B", "0"))
        # merged iteration: the discriminator's Adam in two launches - the classifier (18.9 of 23.6 M parameters) on the side stream as
        # soon as its gradient is complete and the generator's backward has read the weights, the feature stack after the join
        # OFF: measured slower (5.015 vs 4.960 ms, same box): the 75.5 MB classifier's update streams 528 MB through the Infinity Cache
        # while both branches are running and evicts what their conv kernels were re-reading; at the join nothing else runs
        self.KERNEL.SPLIT_D_ADAM = os.environ.get("SST_SPLIT_D_ADAM", "0") != "0"
        # merged iteration, the other direction: with the discriminator step batched the GENERATOR's branch is the longer one - its
        # weight gradients (leaves of its backward) run on the discriminator's stream after that branch's work.  Mask: 1 conv3 (9x9),
        # 2 the up-sampling convs, 4 the grouped trunk launch, 8 conv1 (9x9); 0 = none
        # OFF: measured slower (5.00 ms with none, 5.26 with the trunk launch moved, 5.76-5.88 with more): the generator's backward does not
        # get shorter without them (device stamps: it ends at 4.9-5.0 ms either way - its chain of short launches only gets the chip when
        # the discriminator branch's persistent conv kernels leave it), the moved launches just run after everything else.  Stream
        # priorities change nothing either (main high: 5.00 ms; side high: 8.55 ms)
        self.KERNEL.DEFER_G_WGRAD = int(os.environ.get("SST_DEFER_G_WGRAD", "0"))
        # merged iteration: where the discriminator step's branch forks off the generator's stream - 0: after the generator step's forward
        # and losses (round 2), 1: after the generator's backward has passed the discriminator's classifier, 2: after it has left the
        # discriminator altogether.  The generator's backward is a chain of short launches that only advances when the other branch's
        # persistent conv kernels let it (its first launches - a 1-workgroup BCE backward, the classifier's data-gradient - took 75 /
        # 123 us beside the discriminator step's forward, 5 / 37 us alone): what runs before the fork runs at full speed
        self.KERNEL.FORK_D_STEP_AT = int(os.environ.get("SST_FORK_D_STEP_AT", "0"))
        self.KERNEL.EARLY_D_PACK = os.environ.get("SST_EARLY_D_PACK", "1") != "0"
        self.KERNEL.REUSE_D_SR = os.environ.get("SST_REUSE_D_SR", "1") != "0"
        # the discriminator step's two passes, D(gt) and D(sr.detach()) (train.py:155-158), as ONE batch of 2B images with per-pass
        # train-mode BatchNorm statistics (disc_graph.forward on a list of inputs): one launch per layer instead of two, the classifier
        # weight streamed once per direction.  Same values per pass up to fp32 summation order (the weight gradients sum over 2B images in
        # one kernel instead of fl(dW_sr + dW_gt)); applies when the step runs both passes (i.e. not on top of REUSE_D_SR's kept pass)
        self.KERNEL.BATCH_D_STEP = os.environ.get("SST_BATCH_D_STEP", "1") != "0"
        self.KERNEL.LR_ON_DEVICE = False    # True: the LR batch is synthesised from the GT batch on the GPU (sst_bicubic, same
                                            # values as dataset.py:28 on the 1/255 grid) instead of taking the loader's copy

    def add_g_criterion(self, name: str, value, weight: float = 1.0) -> None:
        self.MODEL.G_LOSS.CRITERIONS[name] = value
        self.MODEL.G_LOSS.CRITERION_WEIGHTS[name] = weight

    def remove_g_criterion(self, name: str) -> None:
        if name in self.MODEL.G_LOSS.CRITERIONS:
            del self.MODEL.G_LOSS.CRITERIONS[name]
            del self.MODEL.G_LOSS.CRITERION_WEIGHTS[name]

    def get_all_params(self) -> str:
        params = [getattr(self, attr) for attr in dir(self)
                  if not callable(getattr(self, attr)) and not attr.startswith("__")]
        return str(params)
