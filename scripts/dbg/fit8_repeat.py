import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_configs_full as T
import conftest
from bench import chore_opt
opt = chore_opt("fp16x3") if len(sys.argv) < 2 else chore_opt(sys.argv[1])
runs = [T._fit8(opt, False) for _ in range(3)] + [T._fit8(opt, True) for _ in range(3)]
names = ("pose", "betas", "trans", "obj_t", "obj_s", "R")
for i in range(1, 6):
    print("run", i, "vs 0:", {n: float(np.abs(a - b).max()) for n, a, b in zip(names, runs[0], runs[i])})
