from .build import build_optimizer, freeze_bn, FlatSGD, FlatAdam, TrainStep, GraphedTrainStep, param_ranges
from .train_loop import (WarmupMultiStepLR, WarmupCosineLR, build_lr_scheduler, Checkpointer, PeriodicCheckpointerOnlyOne,
                         do_train, make_train_step)
