from .build import build_optimizer, freeze_bn, FlatSGD, TrainStep, GraphedTrainStep
