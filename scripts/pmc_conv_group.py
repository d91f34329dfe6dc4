"""the launches bench_train.dominant_kernel_roofline times (grouped 3x3 256->256 convolutions over the five pyramid levels, fp32:
forward, backward-data, weight gradient), a few times each, for the rocprofv3 --pmc passes of scripts/pmc_r03.sh"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
bt = importlib.import_module("bench_train")
print(bt.dominant_kernel_roofline(torch.device("cuda:0"), "fp32", reps=4)["all_directions"])
