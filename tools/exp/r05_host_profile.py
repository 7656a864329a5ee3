"""r05: where does the HOST spend its ~18 us per launch in the eager CIFAR step?  cProfile of 20 eager steps of tools/bench_cifar.py's trainer (bs 64)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")
step, _ = bench.build(dev, "simclr", arch="resnet18", reduce_bottom_conv=True)
batch = {"aug_1": torch.randn(64, 3, 32, 32, device=dev), "aug_2": torch.randn(64, 3, 32, 32, device=dev)}
for _ in range(5):
    step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step(batch)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
