import json, sys, torch
sys.path.insert(0, "/root/repo")
import bench
print(json.dumps(bench.eval_knn_leg(torch.device("cuda:0"))))
