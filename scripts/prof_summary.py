"""summary of gpurun_out/prof_<mode>.csv: kernel time per step by family"""
import csv, sys, collections
f = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fam = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "at::native" in n: k = "ATen:" + ("fill" if "FillFunctor" in n else "add" if "CUDAFunctor_add" in n else "copy" if "direct_copy" in n else "cat" if "CatArray" in n else "other")
    elif "wgrad" in n: k = "conv wgrad"
    elif "igemm" in n or "conv_patch" in n: k = "conv fwd/bwd-data"
    elif "k_bn_" in n: k = "BatchNorm"
    elif "splitk" in n: k = "split-K epilogue"
    elif "roi" in n: k = "RoIAlign"
    elif "sgd" in n or "nonfinite" in n: k = "optimizer"
    elif "topk" in n or "nms" in n: k = "topk/nms"
    else: k = "other own"
    fam[k][0] += int(r["Calls"]); fam[k][1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in fam.values())
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:22s} {t / steps / 1e3:9.1f} us/step  {c / steps:7.1f} launches/step")
print(f"{'total':22s} {tot / steps / 1e3:9.1f} us/step  {sum(v[0] for v in fam.values()) / steps:7.1f} launches/step")
