import importlib, os, sys, time, torch, collections, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "boxnet_bench.py")).read().split("for _ in range(70):")[0]
exec(src)
for it in range(8):
    model.inference(batch, experiment_type={"use_pred_boxes": False}, generator=gen); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for it in range(9):
    model.inference(batch, experiment_type={"use_pred_boxes": False}, generator=gen); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(6)
